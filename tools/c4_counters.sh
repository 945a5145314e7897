#!/bin/bash
# SQ counters of the MS + document walk at the declared C4 (bench leg c4_ms_doc), two passes: instruction / wait cycles, then LDS.
# Per-launch averages per kernel; the PML walk of the headline is in the same table for comparison.   usage (gpurun): bash tools/c4_counters.sh
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for pass in 1 2; do
  if [ $pass = 1 ]; then C="SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY";
  else C="SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAVES"; fi
  rm -rf /tmp/pmc$pass
  timeout 900 rocprofv3 --pmc $C --kernel-trace --output-format csv -d /tmp/pmc$pass -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --legs c4_ms_doc "$@" > /tmp/pmc$pass.log 2>&1 || tail -5 /tmp/pmc$pass.log
  f=$(find /tmp/pmc$pass -name "*counter_collection.csv" | head -1)
  python - "$f" <<PY
import csv, sys, collections, re
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = re.sub(r"void spx::\(anonymous namespace\)::", "", row["Kernel_Name"])[:48]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[(k, row["Counter_Name"])] += 1
for k, v in agg.items():
    if "walk" in k or "extend" in k:
        print(k, {a: "%.3e" % (b / cnt[(k, a)]) for a, b in sorted(v.items())}, "launches", max(cnt[(k, a)] for a in v))
PY
done
