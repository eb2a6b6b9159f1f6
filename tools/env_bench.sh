#!/bin/bash
# bench.py (480 steps, no CPU baseline) under several values of one environment variable: tools/env_bench.sh VAR v1 v2 ...
var=$1; shift
for v in "$@"; do
  env $var=$v timeout -k 10 200 python3 bench.py --steps 480 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$var=$v', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), d['roofline']['kernel_ms_per_step'])" || exit 1
done
