export TMPDIR=/tmp
for s in 6.8 8.24 9.9; do SPX_FAT_SLOTS_PER_RUN=$s REAL_AB_CHECK=4000 python tools/real_ab.py 2>/tmp/e.err || tail -3 /tmp/e.err; done
for rep in 1 2; do
  python bench.py --no-cpu-baseline --legs positive_100,positive_0 2>/tmp/b.err | tail -1 | python -c "
import sys,json
d=json.loads(sys.stdin.read()); r=d['roofline']
print(d['config']['index_hbm_bytes'], d['config']['index_layout'].get('fat_slots_per_run'), f\"{d['value']/1e6:.1f} M reads/s {r['kernel_ms_avg']} ms rows {r['row_loads_per_step']} dir {r['dir_loads_per_step']}\", 'pos100', d['positive_100']['steps_per_s']/1e9, 'pos0', d['positive_0']['steps_per_s']/1e9, d['config']['setup_s'])" || tail -3 /tmp/b.err
done
