#!/bin/bash
lib=spumoni_amd/libspumoni_gpu.so
for old in "" 1; do
  echo "== SPX_OLD_WALK=$old"
  env ${old:+SPX_OLD_WALK=1} SPUMONI_GPU_LIB=$PWD/$lib python tools/sweep.py ms 2>&1 | grep -E "Gsteps" | cut -c1-190
done
