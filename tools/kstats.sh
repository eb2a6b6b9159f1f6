#!/bin/bash
# per-kernel durations of the benchmark step: rocprofv3 --kernel-trace --stats on bench.py (arguments are passed to bench.py)
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats
timeout -k 5 400 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats -- python3 bench.py --no-cpu-baseline "$@" > gpurun_out/kstats.log 2>&1 || exit 1
f=$(ls gpurun_out/kstats/*/*kernel_stats.csv | head -1)
cp $f gpurun_out/kernel_stats.csv
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open('gpurun_out/kernel_stats.csv')))
for r in rows[:16]:
    print(f"{r['Name'][:70]:70s} calls={r['Calls']:>6s} avg_us={float(r['AverageNs'])/1e3:8.1f} pct={r['Percentage']}")
PY
grep metric gpurun_out/kstats.log | tail -1 > gpurun_out/kstats_bench.json
