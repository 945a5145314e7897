#!/bin/bash
for rep in 1 2; do for lib in spumoni_amd/libspumoni_gpu*.so; do
  echo -n "$lib: "; SPUMONI_GPU_LIB=$PWD/$lib MS_BENCH_BITS=16 python tools/ms_bench.py 2>&1 | grep -E "MS \+doc:"
done; done
