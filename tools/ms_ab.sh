#!/bin/bash
# every spumoni_amd/libspumoni_gpu*.so through tools/ms_ab.py (C4 shape) and tools/ms_bench.py (E. coli, MS lengths), interleaved
for rep in $(seq 1 ${AB_REPS:-2}); do
  for lib in spumoni_amd/libspumoni_gpu*.so; do
    SPUMONI_GPU_LIB=$PWD/$lib timeout 900 python tools/ms_ab.py ${MS_AB_LG:-27} 2>&1 | grep -v "amdgpu.ids"
    [ -n "$MS_AB_ECOLI" ] && { echo "== $lib (E. coli)"; SPUMONI_GPU_LIB=$PWD/$lib MS_BENCH_BITS=16 timeout 900 python tools/ms_bench.py 2>&1 | grep "+doc"; }
  done
done
