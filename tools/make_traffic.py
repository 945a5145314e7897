"""profiles/traffic.json from the PMC passes of tools/profile_round.sh.

usage: make_traffic.py <dir with pmc_fetch/ pmc_write/ [pmc_tcc/]> <bench.json of the same command> <out.json>

HBM bytes per step (the walk kernel + the kernel that writes the lengths out from its reset bits) = 2 x FETCH_SIZE + WRITE_SIZE (both in KB): on gfx950 FETCH_SIZE
tallies every 128-byte L2 line fill as 64 bytes (MI355X_MICROARCH.md; calibrated on this access pattern in
profiles/r01_gather_calibration.txt).  The figure is keyed on the library version, the index geometry and the
batch (bench.py's roofline.traffic_key): bench.py quotes it only when the key matches."""
import csv, glob, json, os, sys

src, bench_json, out = sys.argv[1:4]


def mean_counter(sub, name):
    """per step: the walk's dispatch plus the dispatch that writes the lengths out from the walk's reset bits"""
    total, n = None, 0
    for kernel in ("k_walk_fast", "k_walk_lanes", "k_expand_lengths"):
        vals = []
        for f in glob.glob(os.path.join(src, sub, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if kernel in row["Kernel_Name"] and row["Counter_Name"] == name:
                    vals.append(float(row["Counter_Value"]))
        if vals:
            total = (total or 0.0) + sum(vals) / len(vals)
            n = max(n, len(vals))
    return total, n


line = [l for l in open(bench_json) if l.startswith("{")][-1]
b = json.loads(line)
fetch, nf = mean_counter("pmc_fetch", "FETCH_SIZE")
write, nw = mean_counter("pmc_write", "WRITE_SIZE")
rd, _ = mean_counter("pmc_tcc", "TCC_EA0_RDREQ_sum")
assert fetch is not None and write is not None, "no k_walk dispatches in the PMC output"
tj = {
    "key": b["roofline"]["traffic_key"],
    "hbm_bytes_per_launch": int(2 * fetch * 1024 + write * 1024),
    "line_fills_per_launch": int(rd) if rd else None,  # TCC_EA0_RDREQ: 128-byte lines read out of HBM
    "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, tools/profile_round.sh), mean over {nf} "
              f"k_walk_fast (k_walk_lanes + k_expand_lengths where the state machine runs) dispatches of `python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras`: "
              f"FETCH_SIZE {fetch:.4g} KB (x2: gfx950 counts a 128-B line fill as 64 B), WRITE_SIZE {write:.4g} KB"
              + (f", TCC_EA0_RDREQ {rd:.4g}" if rd else ""),
}
json.dump(tj, open(out, "w"), indent=1)
print(json.dumps(tj))
