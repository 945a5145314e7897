#!/bin/bash
# Round 5: the tail of the CLI run -- giving the files' excess back (un-registering, page table entries, the cut), with the
# page table entries dropped ahead by 8 threads (default) and without (SPUMONI_SETTLE_THREADS=0); then the CLI tests.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_trim
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for rep in 1 2 3; do
  for mode in "" "SPUMONI_SETTLE_THREADS=0" "SPUMONI_SETTLE_THREADS=16" "SPUMONI_REPORT_ONLY=1"; do
    echo "== rep $rep $mode"
    env $mode timeout 60 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "first super-batch|Finished processing"
  done
done > $out/settle.txt 2>&1
cat $out/settle.txt
rm -rf /dev/shm/e2e
timeout 1200 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
