#!/bin/bash
# gather_modes.bin over allocation types and load flavours, then the memory-side request counters of a few of them.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
GB=${1:-64}
mkdir -p gpurun_out
timeout 600 tools/gather_modes.bin $GB 2>&1 | tee gpurun_out/gather_modes.txt
rocprofv3 -L 2>/dev/null | grep -o "TCC_EA0_RDREQ[A-Za-z0-9_]*\|TCC_EA0_RD_UNCACHED[A-Za-z0-9_]*\|TCC_BUBBLE[A-Za-z0-9_]*\|TCC_MISS[A-Za-z0-9_]*\|TCC_REQ[A-Za-z0-9_]*" | sort -u | tr '\n' ' ' | tee -a gpurun_out/gather_modes.txt
echo | tee -a gpurun_out/gather_modes.txt
for combo in "0 0" "0 5" "2 0" "1 0"; do
  rm -rf /tmp/gm
  timeout 300 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum --kernel-trace --output-format csv -d /tmp/gm -- tools/gather_modes.bin $GB $combo > /tmp/gm.log 2>&1
  grep "alloc=" /tmp/gm.log | tee -a gpurun_out/gather_modes.txt
  f=$(find /tmp/gm -name "*counter_collection.csv" | head -1)
  python - "$f" <<PY | tee -a gpurun_out/gather_modes.txt
import csv, sys, collections
agg = collections.defaultdict(float)
try:
    for row in csv.DictReader(open(sys.argv[1])):
        if "k_chase" in row["Kernel_Name"]:
            agg[row["Counter_Name"]] += float(row["Counter_Value"])
    g = 4 * 256 * 256 * 1050
    print("   per gather:", {k: round(v / g, 3) for k, v in agg.items()})
except Exception as e:
    print("   no counters:", e)
PY
done
