# host_path leg: growing pieces against equal pieces, with the pipeline's own timeline (SPX_PIPE_TRACE).  Through gpurun.
for mode in grow even; do
  if [ $mode = even ]; then export SPX_PIPE_EVEN=1; else unset SPX_PIPE_EVEN; fi
  SPX_PIPE_TRACE=1 python bench.py --no-cpu-baseline --legs host_path --steps 3 --warmup 1 2>/tmp/hp_$mode.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_path']
print('$mode', d['value'], h['value'], h['ms_per_batch'], h['calls_ms'])"
  grep -A12 "spx pipeline: host" /tmp/hp_$mode.err | tail -13
done
