#!/bin/bash
# Per-kernel times (rocprofv3 --kernel-trace --stats) of the legs whose dominant kernel is NOT the PML walk: the MS + document walk
# and k_ms_extend (c4_ms_doc, real_bwt_ms_doc), the digestion kernels (digest_inclusive, digest_long_reads, real_bwt_digest_walk),
# the chunked walk's passes (long_reads).   usage (gpurun): bash tools/other_kernels_prof.sh <out.txt>
out=${1:-gpurun_out/other_kernels.txt}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof_o && mkdir -p /tmp/prof_o
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_o/kt -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 \
   --legs c4_ms_doc,real_bwt_ms_doc,digest_inclusive,digest_long_reads,real_bwt_digest_walk,long_reads > /tmp/prof_o/bench.json 2>/tmp/prof_o/err.log
mv /tmp/prof_o/kt/*/* /tmp/prof_o/kt/ 2>/dev/null
python - "$out" <<'PY'
import csv, glob, sys, re
out = sys.argv[1]
rows = []
for f in glob.glob("/tmp/prof_o/kt/*kernel_stats.csv"):
    rows += list(csv.DictReader(open(f)))
lines = ["rocprofv3 --kernel-trace --stats -- python bench.py --no-cpu-baseline --steps 3 --warmup 1 --legs c4_ms_doc,real_bwt_ms_doc,digest_inclusive,digest_long_reads,real_bwt_digest_walk,long_reads",
         "(spx kernels only; name shortened; calls, average / min / max in microseconds, share of all GPU time of the command)"]
for r in sorted(rows, key=lambda r: -float(r["TotalDurationNs"])):
    n = r["Name"]
    if "spx" not in n:
        continue
    n = re.sub(r"void |spx::\(anonymous namespace\)::|\(spx::.*|\(unsigned.*|\(.*", "", n)
    lines.append(f"{n:70s} calls {int(r['Calls']):5d}  avg {float(r['AverageNs'])/1e3:10.1f}  min {float(r['MinNs'])/1e3:10.1f}  max {float(r['MaxNs'])/1e3:10.1f}  {float(r['Percentage']):6.2f} %")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines[:60]))
PY
