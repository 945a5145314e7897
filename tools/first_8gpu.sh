#!/bin/bash
# First contact with an 8-GPU node (nothing in this repository has ever run on two devices: DESIGN.md 5 / 6).
# One command, everything into profiles/ (or $1): topology and peer access, the bench at 1 / 2 / 4 / 8 ranks (each line
# carries both scalings), the CLI with one and with all devices on a 10^7-read FASTA, replica clone times, the multi-GPU
# tests that the one-GPU boxes skip.  Nothing here is needed for correctness -- the paths are covered by the gloo and the
# two-ranks-on-one-GPU tests -- it is the first minute of measurement made to count.
#   bash tools/first_8gpu.sh [outdir]
set -u
ROOT=$(cd "$(dirname "$0")/.." && pwd)
out=${1:-$ROOT/profiles/first_8gpu}
mkdir -p "$out"
cd "$ROOT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
ngpu=$(python - <<'PY'
import torch
print(torch.cuda.device_count())
PY
)
echo "devices visible: $ngpu" | tee "$out/summary.txt"
{
  rocm-smi --showtopo 2>/dev/null
  rocm-smi --showmeminfo vram 2>/dev/null | head -40
  python - <<'PY'
import torch
n = torch.cuda.device_count()
print("peer access matrix (row can reach column):")
for a in range(n):
    print(" ".join("1" if a == b or torch.cuda.can_device_access_peer(a, b) else "0" for b in range(n)))
PY
  lscpu | grep -E "Model name|^CPU\(s\)|NUMA"
  cat /sys/fs/cgroup/cpu.max 2>/dev/null
} > "$out/topology.txt" 2>&1
python -c "import __graft_entry__ as g; g.build()" > "$out/build.log" 2>&1
# ---- the bench: weak and strong scaling in one line per N (other_scaling) ----
for n in 1 2 4 8; do
  [ "$n" -le "$ngpu" ] || continue
  timeout 1800 python bench.py --gpus $n --steps 20 --warmup 3 --no-extras > "$out/bench_n$n.json" 2> "$out/bench_n$n.log"
  python - "$out/bench_n$n.json" <<'PY' | tee -a "$out/summary.txt"
import json, sys
try:
    j = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    o = j.get("other_scaling") or {}
    print(f"bench n_gpus={j['n_gpus']}: {j['scaling']} {j['value']/1e6:.0f} M reads/s ({j['ms_per_step']:.2f} ms/step); "
          f"{o.get('scaling')} {o.get('value', 0)/1e6:.0f} M reads/s; per rank ms {j.get('per_rank_ms_per_step', {}).get('min')}..{j.get('per_rank_ms_per_step', {}).get('max')}")
except Exception as e:
    print("bench line unreadable:", e)
PY
done
# ---- the CLI: one queue of parsed super-batches, three workers on one device / two per device on all, the index copied device to device ----
d=${E2E_DIR:-/dev/shm/e2e8}
E2E_DIR=$d E2E_READS=${E2E_READS:-10000000} E2E_CPU_READS=20000 E2E_ONLY_SETUP=1 timeout 1800 python tools/cli_e2e.py > "$out/cli_setup.txt" 2>&1
for gpus in 0,0,0 all; do
  [ "$gpus" = all ] && [ "$ngpu" -lt 2 ] && continue
  for mode in "" "SPUMONI_REPORT_ONLY=1"; do
    echo "== SPUMONI_GPUS=$gpus $mode"
    env SPUMONI_GPUS=$gpus $mode spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "timing\]|done\.|finished"
  done
done > "$out/cli_multi_gpu.txt" 2>&1
grep -E "^==|first read|replica|gpu worker" "$out/cli_multi_gpu.txt" | tee -a "$out/summary.txt"
rm -rf "$d"
# ---- tests that need more than one device ----
timeout 1800 python -m pytest tests -m gpu -q -k "ranks or clone or cli_two or four_workers or sharding" > "$out/pytest_multi.txt" 2>&1
tail -3 "$out/pytest_multi.txt" | tee -a "$out/summary.txt"
echo "everything under $out"
