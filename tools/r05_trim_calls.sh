#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r05_trim
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for mode in "X=1" "SPUMONI_TRIM_MIN=99999999999" "SPUMONI_PIN_SHARE=1"; do
  for rep in 1 2; do
    echo "== $mode rep $rep"
    env SPUMONI_CALL_TRACE=1 $mode timeout 20 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "first super-batch|gpu worker|calls\]"
  done
done > $out/calls2.txt 2>&1
rm -rf /dev/shm/e2e
