#!/bin/bash
# Round 5: the two regimes of the r = 2e9 index in ONE process (tools/c5_regimes.py: 0 GB and 12 GB held during the layout),
# under rocprofv3 with the translation counters the box exposes, one pass per counter set; per k_walk_fast dispatch in launch
# order: duration and counters.   bash tools/c5_regimes_pmc.sh  -> gpurun_out/r05_c5_regimes_pmc.txt
out=$GRAFT_REPO_ROOT/gpurun_out/r05_c5_regimes_pmc.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export C5_REGIMES_HOLDS=0,12,0,12
: > $out
for set in "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE" "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum" "TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum" "TCP_TCP_LATENCY_sum TCP_TA_TCP_STATE_READ_sum"; do
  rm -rf /tmp/pmc_c5
  echo "== --pmc $set" >> $out
  timeout 900 rocprofv3 --pmc $set --kernel-trace --output-format csv -d /tmp/pmc_c5 -- python tools/c5_regimes.py 2>&1 | grep "^B:" | cut -c1-120 >> $out
  python - >> $out <<'PY'
import csv, glob, collections
ct = glob.glob("/tmp/pmc_c5/**/*counter_collection.csv", recursive=True)
kt = glob.glob("/tmp/pmc_c5/**/*kernel_trace.csv", recursive=True)
dur = {}
for f in kt:
    for row in csv.DictReader(open(f)):
        if "k_walk_fast" in row["Kernel_Name"]:
            dur[row["Dispatch_Id"]] = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e6
rows = collections.OrderedDict()
for f in ct:
    for row in csv.DictReader(open(f)):
        if "k_walk_fast" in row["Kernel_Name"]:
            rows.setdefault(int(row["Dispatch_Id"]), {})[row["Counter_Name"]] = float(row["Counter_Value"])
for d in sorted(rows):
    print(f"  dispatch {d:4d}  {dur.get(str(d), 0):7.3f} ms  " + "  ".join(f"{k}={v:.4g}" for k, v in sorted(rows[d].items())))
PY
done
tail -60 $out
