#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r05_load
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for mode in "X=1" "SPUMONI_REPORT_ONLY=1" "SPUMONI_MAP_OUTPUT=0"; do
  echo "== $mode"
  ( time env SPX_TIMING=1 $mode timeout 60 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n ) 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -vE "text_begin|text_fetch|calls\]" | head -60
done > $out/load.txt 2>&1
rm -rf /dev/shm/e2e
