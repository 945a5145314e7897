#!/bin/bash
# Round 5: where the device's time goes inside one `spumoni run` (4e6 x 200 bp): rocprofv3 kernel + memory-copy trace of the
# CLI itself, then the busy intervals of kernels, host-to-device and device-to-host copies over the "processing" window.
#   bash tools/r05_cli_trace.sh  -> gpurun_out/r05_cli_trace.txt
out=$GRAFT_REPO_ROOT/gpurun_out/r05_cli_trace.txt
cd $GRAFT_REPO_ROOT
E2E_ONLY_SETUP=1 python tools/cli_e2e.py > /dev/null 2>&1
d=/dev/shm/e2e
SPUMONI_CACHE=write spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/cli_trace
echo "== ${TRACE_ENV:-default}" >> $out.tmp
env ${TRACE_ENV:-X=1} rocprofv3 --kernel-trace --memory-copy-trace --output-format csv -d /tmp/cli_trace -- $GRAFT_REPO_ROOT/spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "timing\]|done\." > $out
python - >> $out <<'PY'
import csv, glob
def load(pat, name_key):
    rows = []
    for f in glob.glob("/tmp/cli_trace/**/*" + pat, recursive=True):
        for r in csv.DictReader(open(f)):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get(name_key, "")))
    return rows
ker = load("kernel_trace.csv", "Kernel_Name")
cp = load("memory_copy_trace.csv", "Direction")
walk = [k for k in ker if "k_walk_fast" in k[2]]
t0 = min(k[0] for k in walk); t1 = max(k[1] for k in ker + cp if k[0] >= t0)
def busy(iv):
    iv = sorted((a, b) for a, b, _ in iv if b > t0 and a < t1)
    tot, cur_a, cur_b = 0, None, None
    for a, b in iv:
        a = max(a, t0); b = min(b, t1)
        if cur_b is None or a > cur_b:
            if cur_b is not None: tot += cur_b - cur_a
            cur_a, cur_b = a, b
        else:
            cur_b = max(cur_b, b)
    if cur_b is not None: tot += cur_b - cur_a
    return tot / 1e6
win = (t1 - t0) / 1e6
inw = [k for k in ker if k[0] >= t0]
h2d = [c for c in cp if "HOST_TO_DEVICE" in c[2].upper() and c[0] >= t0]
d2h = [c for c in cp if "DEVICE_TO_HOST" in c[2].upper() and c[0] >= t0]
print(f"window (first walk kernel .. last device activity): {win:.1f} ms")
print(f"  kernels busy (union)        {busy(inw):7.1f} ms   ({len(inw)} launches; k_walk_fast alone {busy(walk):.1f} ms in {len(walk)} launches)")
print(f"  host-to-device busy (union) {busy(h2d):7.1f} ms   ({len(h2d)} copies, {sum(1 for _ in h2d)} )")
print(f"  device-to-host busy (union) {busy(d2h):7.1f} ms   ({len(d2h)} copies)")
print(f"  anything busy (union)       {busy(inw + h2d + d2h):7.1f} ms  -> idle {win - busy(inw + h2d + d2h):.1f} ms")
print(f"  copies of both directions at once: {busy(h2d) + busy(d2h) - busy(h2d + d2h):.1f} ms; a kernel and a copy at once: {busy(inw) + busy(h2d + d2h) - busy(inw + h2d + d2h):.1f} ms")
# the timeline of ~4 super-batches from the middle of the run: every copy above 100 KB and every kernel above 20 us
mid = t0 + (t1 - t0) // 2
ev = [(a, b, "H2D " if "HOST_TO_DEVICE" in n.upper() else "D2H ", "") for a, b, n in cp if b - a > 20000] + [(a, b, "kern", n.split("(")[0][:60]) for a, b, n in ker if b - a > 20000]
ev = sorted(e for e in ev if mid <= e[0] < mid + 13_000_000)
print("  timeline from the middle of the run (ms from its start; start, end, duration):")
for a, b, kind, name in ev:
    print(f"    {(a - mid) / 1e6:7.3f} {(b - mid) / 1e6:7.3f} {(b - a) / 1e6:6.3f}  {kind} {name}")
big = sorted(d2h, key=lambda c: c[0] - c[1])[:5]
print("  longest device-to-host copies (ms):", [round((c[1] - c[0]) / 1e6, 2) for c in big])
PY
rm -rf /dev/shm/e2e
cat $out
