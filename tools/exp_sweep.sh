#!/bin/bash
# CLI runs with the library's stage timing
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
os.environ["E2E_READS"] = "4000000"; os.environ["E2E_CPU_READS"] = "1000"
__file__ = os.path.join(os.getcwd(), "tools", "cli_e2e.py")
src = open("tools/cli_e2e.py").read().split('run("raw index files')[0]
exec(src)
for env in ({"SPUMONI_CACHE": "write"}, {}, {}, {"SPUMONI_HOST_FORMAT": "1"}, {}):
    e = dict(os.environ, **env)
    r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{d}/ref", "-p", f"{d}/reads.fa", "-P", "-c", "-n"], capture_output=True, env=e)
    print(env, r.stderr.decode()[-900:])
PY
python bench.py --no-cpu-baseline --no-extras 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('headline', round(d['value']/1e6,1), 'M reads/s; ms_per_step', d['ms_per_step'], 'kernel', d['roofline']['kernel_ms_avg'])"
