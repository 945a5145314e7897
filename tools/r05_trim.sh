#!/bin/bash
# Round 5: the tail of the CLI run.  The upper part of a value stream's prepared tail is not registered with the device
# (SPUMONI_PIN_SHARE, default 0.7) and its excess is cut off BESIDE the run (EarlyTrim); SPUMONI_PIN_SHARE=1 is the form
# before (everything registered, the excess cut at the end).  Every run under a short timeout: the first attempt at cutting
# beside the run -- registered pages -- hung the device.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_trim
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
cp $d/reads.fa.pseudo_lengths $d/want.pseudo_lengths; cp $d/reads.fa.report $d/want.report
for rep in 1 2 3 4; do
  for mode in "X=1" "SPUMONI_PIN_SHARE=1" "SPUMONI_GPUS=0,0" "SPUMONI_REPORT_ONLY=1"; do
    echo "== rep $rep $mode"
    env $mode timeout 20 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "first super-batch|output bytes|Finished processing"
    echo "   exit $? ; files: $(cmp $d/reads.fa.pseudo_lengths $d/want.pseudo_lengths && cmp $d/reads.fa.report $d/want.report && echo identical)"
  done
done > $out/settle.txt 2>&1
cat $out/settle.txt
rm -rf /dev/shm/e2e
timeout 1200 python -m pytest tests/test_gpu_cli.py -m gpu -x -q > $out/pytest.txt 2>&1
tail -3 $out/pytest.txt
