export TMPDIR=/tmp
for n in 10000000 16000000; do
E2E_READS=$n E2E_PREP_AB=falloc,pipe E2E_PREP_REPS=3 E2E_CPU_READS=1000 python tools/cli_e2e.py 2>&1 | grep "^== \|files prepared" | cut -c1-420
done
