#!/bin/bash
out=$GRAFT_REPO_ROOT/gpurun_out/r05_drain
mkdir -p $out
cd $GRAFT_REPO_ROOT
g++ -O2 -pthread tools/drain_bench.cpp -o /tmp/drain_bench
for nt in 4 16; do /tmp/drain_bench /dev/shm/x 2000 154 $nt; done > $out/drain.txt 2>&1
timeout 300 tools/drain_hip.bin /dev/shm/x 2000 154 > $out/drain_hip.txt 2>&1
cat $out/drain_hip.txt; grep -E "pre\+|^pwrite1|repetition" $out/drain.txt
