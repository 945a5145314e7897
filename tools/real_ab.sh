#!/bin/bash
# A/B of every spumoni_amd/libspumoni_gpu*.so on a REAL digested BWT at the declared table density (tools/real_ab.py), interleaved
# on one box; then (AB_HEADLINE=1) the declared C3 headline of each.   usage (gpurun): bash tools/real_ab.sh
cd "$(dirname "$0")/.."
export TMPDIR=/tmp
python tools/real_ab.py > /tmp/real_ab_first.txt 2>/tmp/real_ab.err || { tail -5 /tmp/real_ab.err; exit 1; }   # (builds the cache)
for rep in $(seq 1 ${AB_REPS:-2}); do
  for lib in spumoni_amd/libspumoni_gpu*.so; do
    SPUMONI_GPU_LIB=$PWD/$lib REAL_AB_CHECK=$([ $rep = 1 ] && echo 4000 || echo 0) timeout 900 python tools/real_ab.py 2>/tmp/real_ab.err || tail -3 /tmp/real_ab.err
  done
done
if [ -n "$AB_HEADLINE" ]; then
  AB_REPS=${AB_HEADLINE} bash tools/ab.sh --no-extras
fi
