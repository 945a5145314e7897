#!/bin/bash
# Collects the evidence kept under profiles/ for one kernel version (run through gpurun):
#   bash tools/profile_round.sh r02_v1 [bench args, e.g. --stand-in]
# -> gpurun_out/prof_<tag>/{summary.txt, bench.json, bench_under_rocprof.json, sq.txt, traffic.json}
tag=${1:-vX}; shift
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
mkdir -p $out
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf /tmp/prof && mkdir -p /tmp/prof
CMD="python bench.py --steps 5 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof/kt -- $CMD > $out/bench_under_rocprof.json 2>/tmp/prof/kt.log
mv /tmp/prof/kt/*/* /tmp/prof/kt/ 2>/dev/null
CMD2="python bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-extras $*"
timeout 900 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d /tmp/prof/pmc_fetch -- $CMD2 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d /tmp/prof/pmc_write -- $CMD2 > /dev/null 2>&1
timeout 900 rocprofv3 --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_HIT_sum TCC_MISS_sum --kernel-trace --output-format csv -d /tmp/prof/pmc_tcc -- $CMD2 > /dev/null 2>&1
for d in pmc_fetch pmc_write pmc_tcc; do mv /tmp/prof/$d/*/* /tmp/prof/$d/ 2>/dev/null; done
python tools/prof_summary.py /tmp/prof $out/summary.txt > /dev/null
python tools/make_traffic.py /tmp/prof $out/bench_under_rocprof.json $out/traffic.json > /dev/null
bash tools/pmc_sq.sh --no-extras "$@" > $out/sq.txt 2>&1
tail -c 600 $out/summary.txt
