#!/bin/bash
# Round 5: the host harness on the box (gpurun): CLI / cache / sanitizer tests, then the end-to-end stage times.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_cli
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_gpu_cli.py tests/test_gpu_cache.py tests/test_gpu_text.py tests/test_sanitizers.py -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
timeout 900 python tools/cli_e2e.py > $out/cli_e2e.txt 2>&1
d=/dev/shm/e2e
for mode in "" "SPUMONI_REPORT_ONLY=1"; do
  echo "== SPX_TIMING=1 $mode"
  env SPX_TIMING=1 SPUMONI_GPUS=0 $mode spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "spx\]|timing\]" | head -120
done > $out/cli_timing.txt 2>&1
rm -rf /dev/shm/e2e
grep "^==" $out/cli_e2e.txt
