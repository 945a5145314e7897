#!/bin/bash
# Collects the evidence kept under profiles/ for one kernel version (run through gpurun):
#   bash tools/profile_round.sh v6
# -> gpurun_out/prof_<tag>/{summary.txt, bench.json, bench_under_rocprof.json, sq.txt}
tag=${1:-vX}
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof && mkdir -p /tmp/prof
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -- $CMD > $out/bench_under_rocprof.json 2>/tmp/prof/kt.log
mv /tmp/prof/kt/*/* /tmp/prof/kt/ 2>/dev/null
CMD2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline"
rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof/pmc_fetch -- $CMD2 > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof/pmc_write -- $CMD2 > /dev/null 2>&1
rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/prof/pmc_tcc -- $CMD2 > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_tcc; do mv /tmp/prof/$d/*/* /tmp/prof/$d/ 2>/dev/null; done
python tools/prof_summary.py /tmp/prof $out/summary.txt > /dev/null
bash tools/pmc_sq.sh > $out/sq.txt 2>&1
python bench.py > $out/bench.json 2>/dev/null
tail -c 600 $out/summary.txt
