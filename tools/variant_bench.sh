#!/bin/bash
# bench.py (960 steps, no CPU baseline) on several builds of the library: tools/variant_bench.sh lib1.so lib2.so ...
# (development aid: the builds come from build.build_lib(out=..., defines=...) with experiment switches)
for lib in "$@"; do
  HS_LIB_PATH=$PWD/$lib timeout -k 10 200 python3 bench.py --steps 960 --no-cpu-baseline 2>/dev/null | python3 -c "
import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$lib', round(d['value']/1e6,2), 'M', round(d['ms_per_step'],4), d['roofline']['kernel_ms_per_step'])" || exit 1
done
