export TMPDIR=/tmp
# same box: round 5's library against this one on the real BWT -- equal slots, equal memory, each one's default density at r = 1e9
for rep in 1 2; do
SPUMONI_GPU_LIB=$PWD/spumoni_amd/libspumoni_gpu_r05.so SPX_FAT_SLOTS_PER_RUN=6.8 REAL_AB_CHECK=0 python tools/real_ab.py 2>/tmp/e.err || tail -3 /tmp/e.err
for s in 6.8 8.24 9.93; do SPX_FAT_SLOTS_PER_RUN=$s REAL_AB_CHECK=0 python tools/real_ab.py 2>/tmp/e.err || tail -3 /tmp/e.err; done
done
AB_REPS=2 bash tools/ab.sh
