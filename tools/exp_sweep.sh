#!/bin/bash
# NEXT EXPERIMENT (not run yet: round 3's GPU minutes were spent when the balanced pieces landed).
# The balancing passes are gated on the longest run (SPX_BALANCE_MIN_RUN, 2048) so that the bench indexes keep the layout
# their traffic was measured on.  What does lowering the gate do to them?  Headline (longest run ~155: one image pass,
# a handful of cuts at most) and dna_m200 (longest run ~1240, longest image ~20-30 runs: cut at span 8) with the gate at
# 2048 (shipped) and at 64, interleaved on ONE box; flatten time, rows, reads/s, row gathers per character.
#   usage (through gpurun):  bash tools/exp_sweep.sh > gpurun_out/gate_sweep.txt 2>&1
# If neither moves by more than the box noise: ship the gate at 64 and take profiles/traffic.json again
# (tools/profile_round.sh); better still, decide from the images themselves ("Next").
for rep in 1 2; do
  for gate in 2048 64; do
    SPX_BALANCE_MIN_RUN=$gate SPX_TIMING=1 timeout 900 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --legs dna_m200,positive_100 2>/tmp/gate.err | tail -1 | \
      python -c "
import sys, json
d = json.loads(sys.stdin.read()); r = d['roofline']; x = d.get('dna_m200', {}); p = d.get('positive_100', {})
print('gate $gate | headline %.1f M reads/s rows/step %s flat_runs %s flatten %s s | positive_100 %.1f G steps/s | dna_m200 %.1f G steps/s frac %s' % (
    d['value'] / 1e6, r['row_loads_per_step'], d['config']['index_layout']['flat_runs'], d['config']['setup_s']['flatten_on_gpu'],
    p.get('steps_per_s', 0) / 1e9, x.get('steps_per_s', 0) / 1e9, x.get('roofline_frac')))" || tail -3 /tmp/gate.err
    grep "pieces, pass" /tmp/gate.err | sed 's/^/    /'
  done
done
