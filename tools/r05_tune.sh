#!/bin/bash
# Round 5: the two knobs of the split tails on one box -- the registered share of a tail (PML, 4e6 reads) and the super-batch of
# MS runs (1e6 reads); first read .. last byte of three runs each.
out=$GRAFT_REPO_ROOT/gpurun_out/r05_tune
mkdir -p $out
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 timeout 600 python tools/cli_e2e.py > $out/setup.txt 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write timeout 120 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
for share in 0.5 0.6 0.7 0.8 0.9; do
  echo "== SPUMONI_PIN_SHARE=$share: $(for rep in 1 2 3; do env SPUMONI_PIN_SHARE=$share timeout 20 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -oE "first read .. last byte [0-9.]+"| grep -oE "[0-9.]+$"; done | tr '\n' ' ')"
done > $out/tune.txt 2>&1
for sb in 16 32 64; do
  echo "== PML SPUMONI_SUPER_BATCH=$sb MB: $(for rep in 1 2 3; do env SPUMONI_SUPER_BATCH=$((sb << 20)) timeout 20 spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -oE "first read .. last byte [0-9.]+"| grep -oE "[0-9.]+$"; done | tr '\n' ' ')"
done >> $out/tune.txt 2>&1
cat $out/tune.txt
rm -rf /dev/shm/e2e
