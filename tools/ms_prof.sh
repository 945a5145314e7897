#!/bin/bash
# per-kernel times of PML+doc / MS+doc on the 5-strain E. coli case (run through gpurun): bash tools/ms_prof.sh <tag>
tag=${1:-ms}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for bits in 16 32; do
  rm -rf /tmp/msp
  MS_BENCH_BITS=$bits timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/msp -- python tools/ms_bench.py > $out/run_$bits.txt 2>/dev/null
  f=$(find /tmp/msp -name "*kernel_stats.csv" | head -1)
  python - "$f" > $out/kernels_$bits.txt <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "k_walk_lanes" in n or "k_ms_extend" in n or "k_expand" in n or "k_text_from" in n:
        print(f'{n[:110]:110s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us  total {float(r["TotalDurationNs"])/1e6:9.3f} ms')
PY
  echo "== $bits-bit outputs"; grep "+doc\|rebuilt" $out/run_$bits.txt; cat $out/kernels_$bits.txt
done
