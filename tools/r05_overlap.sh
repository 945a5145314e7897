#!/bin/bash
# Round 5: the CLI's device pipeline: when every worker is inside the library (SPUMONI_CALL_TRACE), workers per device.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_overlap
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for g in ${OVERLAP_GPUS:-0 0,0 0,0,0}; do
  echo "== SPUMONI_GPUS=$g"
  env SPUMONI_CALL_TRACE=1 SPUMONI_GPUS=$g ${OVERLAP_ENV:-X=1} timeout 60 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "first super-batch|gpu worker|segment|calls\]|phases\]"
done > $out/calls.txt 2>&1
cat $out/calls.txt
rm -rf /dev/shm/e2e
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_text.py tests/test_gpu_cli.py -m gpu -x -q > $out/pytest.txt 2>&1
tail -5 $out/pytest.txt
