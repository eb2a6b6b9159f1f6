for f in variants/lib_t*.so; do
  HS_LIB_PATH=$PWD/$f timeout -k 5 100 python bench.py --steps 480 --no-cpu-baseline 2>&1 | grep metric | python3 -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$f', round(d['value']/1e6,3), d['roofline']['kernel_ms_per_step'])" || exit 1
done
