#!/bin/bash
# A/B of builds of libspumoni_gpu.so on the bench workload, interleaved, on ONE box (boxes differ
# by a few percent).  Usage (through gpurun):  bash tools/ab.sh  [bench args]
# Variants: every spumoni_amd/libspumoni_gpu*.so   (AB_REPS repetitions, default 3)
# AB_LEGS: extra legs to run and print besides the headline (default positive_100,positive_0)
LEGS=${AB_LEGS:-positive_100,positive_0}
for rep in $(seq 1 ${AB_REPS:-3}); do
  for lib in spumoni_amd/libspumoni_gpu*.so; do
    SPUMONI_GPU_LIB=$PWD/$lib timeout 600 python bench.py --no-cpu-baseline --legs "$LEGS" "$@" 2>/tmp/ab.err | tail -1 | \
      python -c "
import sys,json
d=json.loads(sys.stdin.read())
r=d['roofline']
out=['$lib', f\"{d['value']/1e6:.1f} M reads/s {r['kernel_ms_avg']} ms rows/step {r['row_loads_per_step']} dir/step {r['dir_loads_per_step']}\"]
for k in '$LEGS'.split(','):
    v=d.get(k)
    if v and 'steps_per_s' in v: out.append(f\"{k}: {v['steps_per_s']/1e9:.1f} G steps/s frac {v['roofline_frac']} rows {v.get('row_loads_per_step')} dir {v.get('dir_loads_per_step')}\")
    elif v: out.append(f'{k}: {v}')
print(' | '.join(out))" \
      || tail -2 /tmp/ab.err
  done
done
