#!/bin/bash
# Round 5, first contact with the box: what the host side of `spumoni run` has to work with.
#   bash tools/r05_probe.sh   -> gpurun_out/r05_probe/{box.txt, drain.txt, cli_e2e.txt, cli_timing.txt}
out=$GRAFT_REPO_ROOT/gpurun_out/r05_probe
mkdir -p $out
cd $GRAFT_REPO_ROOT
{
  echo "== nproc $(nproc); cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
  uname -r
  lscpu | grep -E "Model name|^CPU\(s\)|Thread|Socket|NUMA"
  free -g | head -2
  echo "== /dev/shm"; mount | grep shm; df -h /dev/shm | tail -1
  echo "== THP"; for f in enabled shmem_enabled defrag; do echo "$f: $(cat /sys/kernel/mm/transparent_hugepage/$f 2>/dev/null)"; done
  echo "== taskset"; taskset -p $$
  rocm-smi --showmemuse --showuse 2>/dev/null | head -12
} > $out/box.txt 2>&1
g++ -O2 -pthread tools/drain_bench.cpp -o /tmp/drain_bench
for nt in 4 8 16; do /tmp/drain_bench /dev/shm/x 2000 154 $nt; done > $out/drain.txt 2>&1
timeout 900 python tools/cli_e2e.py > $out/cli_e2e.txt 2>&1
d=/dev/shm/e2e
for mode in "" "SPUMONI_REPORT_ONLY=1"; do
  echo "== SPX_TIMING=1 $mode"
  env SPX_TIMING=1 $mode spumoni_amd/bin/spumoni run -r $d/ref -p $d/reads.fa -P -c -n 2>&1 | sed 's/\x1b\[[0-9;]*m//g' | grep -E "spx\]|timing\]" | head -150
done > $out/cli_timing.txt 2>&1
rm -rf /dev/shm/e2e /dev/shm/x*
tail -5 $out/cli_e2e.txt
