#!/bin/bash
# Round 5: feeders per run (SPUMONI_FEEDERS) with the default three workers, values and report-only.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_feeders
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for rep in 1 2 3; do
  for f in 2 3 4; do
    for mode in "X=1" "SPUMONI_REPORT_ONLY=1"; do
      echo "== rep $rep SPUMONI_FEEDERS=$f $mode"
      env SPUMONI_FEEDERS=$f $mode timeout 20 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "first super-batch|segment" | cut -c1-250
    done
  done
done > $out/feeders.txt 2>&1
cat $out/feeders.txt
rm -rf /dev/shm/e2e
