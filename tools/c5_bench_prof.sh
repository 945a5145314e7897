cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/c5p
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/c5p -- python bench.py --runs 1000000 --no-cpu-baseline --legs long_reads_c5 --runs 1000000000 --reads 100000 2>/dev/null | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read())['long_reads_c5']; print({k:v for k,v in d.items() if k!='what'})"
f=$(find /tmp/c5p -name "*kernel_stats.csv" | head -1)
python - "$f" <<PY
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Name"]
    if "k_walk" in n or "k_chunk" in n or "k_classify" in n:
        print("%-105s calls %4s avg %9.1f us total %9.2f ms" % (n[:105], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
PY
