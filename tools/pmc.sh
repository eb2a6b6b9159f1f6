#!/bin/bash
# HBM traffic of the step kernels from rocprofv3 PMC counters (separate passes, kernel-trace only),
# plus a calibration copy with the same access pattern (one coalesced dword per lane).
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
cat > /tmp/pmc_calib.py <<'PY'
import ctypes, os
L = ctypes.CDLL(os.path.join(os.environ["GRAFT_REPO_ROOT"], "marl-hideandseek_amd", "lib", "libhideseek.so"))
L.hs_debug_calibrate.argtypes = [ctypes.c_int64]
assert L.hs_debug_calibrate(1 << 29) == 0          # 512 MiB read + 512 MiB written
PY
for c in FETCH_SIZE WRITE_SIZE; do
  t=$(echo $c | tr A-Z a-z | cut -d_ -f1)
  rm -rf gpurun_out/pmc_$t gpurun_out/pmc_calib_$t
  timeout -k 5 250 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_$t -- python3 bench.py --steps 40 --no-cpu-baseline > gpurun_out/pmc_$t.log 2>&1 || exit 1
  timeout -k 5 100 rocprofv3 --kernel-trace --pmc $c --output-format csv -d gpurun_out/pmc_calib_$t -- python3 /tmp/pmc_calib.py > gpurun_out/pmc_calib_$t.log 2>&1 || exit 1
done
ls gpurun_out/pmc_fetch/*/ gpurun_out/pmc_calib_fetch/*/
