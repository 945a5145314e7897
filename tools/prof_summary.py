"""Condenses rocprofv3 CSV output (kernel stats + per-pass PMC) into one small text file.
usage: prof_summary.py <dir with kt/ and pmc*/> <out.txt>"""
import csv, glob, os, sys
from collections import defaultdict

src, out = sys.argv[1], sys.argv[2]
lines = []
for f in sorted(glob.glob(os.path.join(src, "kt", "*kernel_stats.csv"))):
    lines.append("== rocprofv3 --kernel-trace --stats: " + os.path.basename(f))
    rows = list(csv.reader(open(f)))
    lines.append(",".join(rows[0]))
    for r in rows[1:]:
        if "spx" in r[0] or len(lines) < 12:
            lines.append(",".join(r))
for d in sorted(glob.glob(os.path.join(src, "pmc*"))):
    for f in glob.glob(os.path.join(d, "*counter_collection.csv")):
        rows = list(csv.reader(open(f)))
        h = rows[0]
        kn, cn, cv = h.index("Kernel_Name"), h.index("Counter_Name"), h.index("Counter_Value")
        for kernel in ("k_walk_fast", "k_walk_lanes", "k_expand_lengths"):
            agg = defaultdict(list)
            for r in rows[1:]:
                if kernel in r[kn]:
                    agg[r[cn]].append(float(r[cv]))
            if not agg:
                continue
            lines.append("== rocprofv3 --pmc (%s), %s dispatches only: per-dispatch values" % (os.path.basename(d), kernel))
            for k, v in agg.items():
                lines.append(f"{k}: n={len(v)} mean={sum(v)/len(v):.6g} min={min(v):.6g} max={max(v):.6g}")
open(out, "w").write("\n".join(lines) + "\n")
print("\n".join(lines))
