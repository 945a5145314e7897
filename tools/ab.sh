#!/bin/bash
# A/B of builds of libspumoni_gpu.so on the bench workload, interleaved, on ONE box (boxes differ
# by a few percent).  Usage (through gpurun):  bash tools/ab.sh  [bench args]
# Variants: every spumoni_amd/libspumoni_gpu*.so   (AB_REPS repetitions, default 3)
for rep in $(seq 1 ${AB_REPS:-3}); do
  for lib in spumoni_amd/libspumoni_gpu*.so; do
    SPUMONI_GPU_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras "$@" 2>/tmp/ab.err | tail -1 | \
      python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e6,1), 'M reads/s', d['roofline']['kernel_ms_avg'], 'ms')" \
      || tail -2 /tmp/ab.err
  done
done
