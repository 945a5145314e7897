# host_path leg of a short bench run (little time between the index build's frees and the leg: the slow mode of
# profiles/r03_host_path_modes.txt shows), copies by SDMA engines (default) against copies by blit kernels.  Through gpurun.
for mode in sdma blit sdma blit; do
  if [ $mode = blit ]; then export HSA_ENABLE_SDMA=0; else unset HSA_ENABLE_SDMA; fi
  SPX_PIPE_TRACE=1 python bench.py --no-cpu-baseline --legs host_path --steps 3 --warmup 1 --extra-steps 8 2>/tmp/hp_$mode.err | python -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); h=d['host_path']
print('$mode', d['value'], h['value'], h['ms_per_batch'], h['calls_ms'])"
  grep -A12 "spx pipeline: host" /tmp/hp_$mode.err | tail -13 | cut -c1-60 | paste - - - - | head -4
done
