#!/bin/bash
# SQ counters of the walk kernel on the bench workload (one pass, 8 SQ slots).  Run through gpurun.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/pmc1
timeout 900 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY --kernel-trace --output-format csv -d /tmp/pmc1 -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline "$@" > /tmp/pmc1.log 2>&1
f=$(find /tmp/pmc1 -name "*counter_collection.csv" | head -1)
python - "$f" <<PY
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter()
for row in csv.DictReader(open(sys.argv[1])):
    k = row["Kernel_Name"][:60]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    if row["Counter_Name"] == "SQ_WAVE_CYCLES": cnt[k] += 1
for k, v in agg.items():
    if "walk" in k or "digest" in k or "expand" in k:
        print(k, cnt[k], {a: "%.3e" % (b / cnt[k]) for a, b in sorted(v.items())})
PY
