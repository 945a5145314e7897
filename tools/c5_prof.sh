#!/bin/bash
# per-kernel times of the chunked walk (run through gpurun): bash tools/c5_prof.sh <tag>
tag=${1:-c5}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for n in ${C5_READS:-50000 6250}; do
  rm -rf /tmp/c5p
  LONG1_READS=$n rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -- python tools/sweep.py long1 > $out/run_$n.txt 2>/dev/null
  f=$(find /tmp/c5p -name "*kernel_stats.csv" | head -1)
  python - "$f" > $out/kernels_$n.txt <<PY
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if "spx::" in n:
        print(f'{n[:110]:110s} calls {r["Calls"]:>4s} avg {float(r["AverageNs"])/1e3:10.1f} us  total {float(r["TotalDurationNs"])/1e6:9.3f} ms')
PY
  cat $out/run_$n.txt | grep "^C5"; cat $out/kernels_$n.txt
done
