#!/bin/bash
# Counters of k_ms_extend (and the MS walk beside it) on the E. coli case of tools/ms_bench.py.  Run through gpurun.
# Three passes (SQ, more SQ, cache), counters only with --kernel-trace.
#   PMC_CMD: another command than tools/ms_bench.py (e.g. "python tools/ms_ab.py 27"); PMC_MATCH: kernel-name substrings, "|"-separated
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export MS_BENCH_BITS=16
for pass in "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_RD" \
            "SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INST_CYCLES_VMEM_RD SQ_ACTIVE_INST_VALU SQ_INSTS_SMEM" \
            "TCP_TCC_READ_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_PENDING_STALL_CYCLES_sum"; do
  rm -rf /tmp/pmcx
  timeout 600 rocprofv3 --pmc $pass --kernel-trace --output-format csv -d /tmp/pmcx -- ${PMC_CMD:-python tools/ms_bench.py} > /tmp/pmcx.log 2>&1
  f=$(find /tmp/pmcx -name "*counter_collection.csv" | head -1)
  python - "$f" "${PMC_MATCH:-k_ms_extend|k_walk_fast<1}" <<PY
import csv, sys, collections
agg = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.defaultdict(collections.Counter)
try:
    rows = list(csv.DictReader(open(sys.argv[1])))
except Exception as e:
    print("no counters:", e); rows = []
for row in rows:
    k = row["Kernel_Name"][:64]
    agg[k][row["Counter_Name"]] += float(row["Counter_Value"])
    cnt[k][row["Counter_Name"]] += 1
for k, v in agg.items():
    if any(t in k for t in sys.argv[2].split("|")):
        print(k, {a: "%.4g" % (b / cnt[k][a]) for a, b in sorted(v.items())}, "dispatches", max(cnt[k].values()))
PY
done
tail -3 /tmp/pmcx.log
