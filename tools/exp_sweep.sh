#!/bin/bash
for s in 0 3 0 3; do python bench.py --no-cpu-baseline --legs host_path --settle $s 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('settle $s:', round(d['value']/1e6,1), 'M reads/s', d['roofline']['kernel_ms_avg'], 'ms; host_path', d['host_path'].get('calls_ms'), round(d['host_path']['value']/1e6,1))"; done
