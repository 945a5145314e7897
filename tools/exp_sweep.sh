export TMPDIR=/tmp
g++ -O2 -pthread tools/prep_bench.cpp -o /tmp/prep_bench
nproc; free -g | head -2
for args in "8 0 4" "8 1 4" "8 1 8" "8 1 16" "8 1 32" "8 1 64"; do /tmp/prep_bench $args; done
E2E_READS=16000000 E2E_PREP_AB=1 E2E_CPU_READS=1000 python tools/cli_e2e.py 2>&1 | grep -v "^    \[timing\] gpu worker\|writer \|segment\|first read"
