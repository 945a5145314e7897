#!/bin/bash
# why is the walk slower per gather at r = 2e9?  SQ + TCC counters of the headline batch on three indexes (run through gpurun)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() {  # label, env, bench args
  echo "=== $1"
  for pass in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
              "TCC_EA0_RDREQ_sum TCC_HIT_sum TCC_MISS_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum" \
              "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"; do
    rm -rf /tmp/pmcd
    env $2 timeout 900 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcd -- python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $3 > /tmp/pmcd.log 2>&1
    f=$(find /tmp/pmcd -name "*counter_collection.csv" | head -1)
    python - "$f" <<PY
import csv, sys, collections
agg = collections.defaultdict(float); cnt = collections.Counter()
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters", e); rows = []
for row in rows:
    if "k_walk_fast" in row["Kernel_Name"]:
        agg[row["Counter_Name"]] += float(row["Counter_Value"]); cnt[row["Counter_Name"]] += 1
print({a: "%.4g" % (b / cnt[a]) for a, b in sorted(agg.items())})
PY
  done
  grep -o '"kernel_ms_avg": [0-9.]*' /tmp/pmcd.log | tail -1
}
run "r=1e9, 6.8 slots/run (204 GB)" "A=1" ""
run "r=1e9, 1.7 slots/run (102 GB)" "SPX_FAT_SLOTS_PER_RUN=1.7" ""
run "r=2e9, 1.7 slots/run (204 GB)" "A=1" "--runs 2000000000"
