#!/bin/bash
for lib in spumoni_amd/libspumoni_gpu.so spumoni_amd/libspumoni_gpu_prev.so; do
  echo "== $lib"
  SPUMONI_GPU_LIB=$PWD/$lib MS_BENCH_BITS=16 python tools/ms_bench.py 2>&1 | grep -E "doc:"
  SPUMONI_GPU_LIB=$PWD/$lib python tools/sweep.py ms 2>&1 | grep -E "Gsteps" | cut -c1-190
done
