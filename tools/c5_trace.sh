# per-launch durations, in launch order, of the chunked long-read batches of the long_reads_c5 bench leg
# (gpurun_out/c5_trace.txt: start_us dur_us kernel).  RUNS / READS as for bench.py.
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/c5t; mkdir -p gpurun_out
rocprofv3 --kernel-trace --output-format csv -d /tmp/c5t -- python bench.py --no-cpu-baseline --legs long_reads_c5 ${RUNS:+--runs $RUNS} --reads ${READS:-100000} 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['long_reads_c5']; print({k:v for k,v in d.items() if k!='what'})"
f=$(find /tmp/c5t -name "*kernel_trace.csv" | head -1)
python - "$f" > gpurun_out/c5_trace.txt <<PY
import csv, sys, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
t0 = rows[0][0]
for s, e, n in rows:
    n = n.replace("(anonymous namespace)::", "").replace("void ", "")
    if not n.startswith("spx::"): continue
    n = re.sub(r"\(.*", "", n)
    print("%12.1f %9.1f %s" % ((s - t0) / 1e3, (e - s) / 1e3, n[:110]))
PY
wc -l gpurun_out/c5_trace.txt
