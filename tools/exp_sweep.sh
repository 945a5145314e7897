#!/bin/bash
# CLI runs: mapped tail against one pwrite per stream
python - <<'PY'
import os, subprocess, sys, time
sys.path.insert(0, os.getcwd())
os.environ["E2E_READS"] = "4000000"; os.environ["E2E_CPU_READS"] = "1000"
__file__ = os.path.join(os.getcwd(), "tools", "cli_e2e.py")
src = open("tools/cli_e2e.py").read().split('run("raw index files')[0]
exec(src)
for env in ({"SPUMONI_CACHE": "write"}, {}, {}, {}):
    e = dict(os.environ, **env)
    r = subprocess.run([f"{ROOT}/spumoni_amd/bin/spumoni", "run", "-r", f"{d}/ref", "-p", f"{d}/reads.fa", "-P", "-c", "-n"], capture_output=True, env=e)
    print(env, r.stderr.decode()[-700:])
PY
