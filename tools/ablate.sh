#!/bin/bash
# developer tool: time the physics kernel with phases disabled (wrong results, timing only)
for f in 0 16777216 33554432 67108864 134217728 251658240; do
  timeout -k 5 120 python bench.py --steps 120 --no-cpu-baseline --flags $f 2>/dev/null | python3 -c "
import sys,json
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('flags', $f, 'ms/step %.3f'%d['ms_per_step'], {k:round(v,3) for k,v in d['roofline']['kernel_ms_per_step'].items()})" >> gpurun_out/ablate.log 2>&1 || exit 1
done
