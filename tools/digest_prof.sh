#!/bin/bash
# per-kernel times of the device digestion (rocprofv3 kernel stats of tools/digest_bench.py): bash tools/digest_prof.sh [bench args]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/dp
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp -- python tools/digest_bench.py --cpu-reads 1000 "$@" 2>/dev/null | grep -v amdgpu
f=$(find /tmp/dp -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_digest" in n or "rocprim" in n or "k_zero" in n:
        print("%-100s calls %4s avg %9.1f us  min %9.1f  max %9.1f" % (n[:100], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
