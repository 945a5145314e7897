#!/bin/bash
# Round 5: where the time outside "loading the index" + "processing the patterns" goes: every stderr / stdout line of one run
# with the time since the process was started (SPX_FREE_TRACE marks the way out).
out=$GRAFT_REPO_ROOT/gpurun_out/r05_exit
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for mode in "X=1" "SPUMONI_SLOW_EXIT=1" "SPUMONI_REPORT_ONLY=1"; do
python - "$d" "$mode" <<'PY'
import subprocess, sys, time, os
d, mode = sys.argv[1], sys.argv[2]
k, v = mode.split("=")
env = dict(os.environ, SPX_FREE_TRACE="1", **{k: v})
t0 = time.time()
p = subprocess.Popen(["spumoni_amd/bin/spumoni", "run", "-r", d + "/ref", "-p", d + "/reads.fa", "-P", "-c", "-n"], stderr=subprocess.STDOUT, stdout=subprocess.PIPE, env=env, bufsize=0)
print("==", mode)
buf = b""
while True:
    c = p.stdout.read(1)
    if not c:
        break
    buf += c
    if c == b"\n":
        line = buf.decode(errors="replace").rstrip()
        buf = b""
        if line and ("timing] writer" not in line) and ("gpu worker" not in line):
            print("%7.3f  %s" % (time.time() - t0, line[:150].replace("\033[32m", "").replace("\033[0m", "")))
p.wait()
print("%7.3f  (process gone)" % (time.time() - t0))
PY
done > $out/exit.txt 2>&1
rm -rf /dev/shm/e2e
cat $out/exit.txt
