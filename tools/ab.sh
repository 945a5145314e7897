#!/bin/bash
# A/B two builds of libspumoni_gpu.so on the bench workload, interleaved, on ONE box (boxes differ
# by a few percent).  Usage (through gpurun):  bash tools/ab.sh  [bench args]
# Variants: every spumoni_amd/libspumoni_gpu*.so
for rep in 1 2 3; do
  for lib in spumoni_amd/libspumoni_gpu*.so; do
    SPUMONI_GPU_LIB=$PWD/$lib python bench.py --no-cpu-baseline "$@" 2>/dev/null | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,1), 'M reads/s', d['roofline']['kernel_ms_avg'], 'ms')"
  done
done
