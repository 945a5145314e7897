#!/bin/bash
# other modes, old library vs new (interleaved)
for lib in spumoni_amd/libspumoni_gpu_r02.so spumoni_amd/libspumoni_gpu.so; do
  echo "== $lib"
  SPUMONI_GPU_LIB=$PWD/$lib MS_BENCH_BITS=16 python tools/ms_bench.py 2>&1 | grep -E "doc:|rebuilt"
  SPUMONI_GPU_LIB=$PWD/$lib python tools/sweep.py ms 2>&1 | grep -E "Gsteps"
  SPUMONI_GPU_LIB=$PWD/$lib python tools/sweep.py long 2>&1 | grep -E "auto|plain"
done
