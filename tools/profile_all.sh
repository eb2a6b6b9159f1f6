#!/bin/bash
# All the profile artifacts of a round in one go (run on the GPU box): $1 = tag, e.g. r2
tag=${1:-rX}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/prof_$tag
python3 marl-hideandseek_amd/build.py --timing > /dev/null || exit 1
O=gpurun_out/prof_$tag
# the PMC passes first: bench.py reports roofline.traffic only from a profiles/<tag>_traffic.json measured on THESE kernel sources
bash tools/pmc.sh > $O/pmc.log 2>&1 || exit 1
python3 tools/pmc_summary.py gpurun_out $O/traffic.json 16000 > $O/traffic_summary.txt || exit 1
cp $O/traffic.json profiles/${tag}_traffic.json
{ python3 -c "import bench; print('csrc_sha', bench.csrc_fingerprint())"; bash tools/sqpmc.sh; } > $O/sq_counters.txt 2>&1 || exit 1
cp $O/sq_counters.txt profiles/${tag}_sq_counters.txt
timeout -k 10 400 python3 bench.py > $O/bench.json 2> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --worlds-per-gpu 65536 --flags 65536 --steps 480 > $O/bench_physics_only_65536.json 2>> $O/bench.err || exit 1
timeout -k 10 200 python3 bench.py --no-cpu-baseline --worlds-per-gpu 16384 --steps 960 > $O/bench_shard_16384.json 2>> $O/bench.err || exit 1
bash tools/kstats.sh > $O/kstats.txt 2>&1 || exit 1
cp gpurun_out/kernel_stats.csv $O/kernel_stats.csv; cp gpurun_out/kstats_bench.json $O/kstats_bench.json
HS_LIB_PATH=$PWD/marl-hideandseek_amd/lib/libhideseek_timing.so timeout -k 10 120 python3 tools/phase_timing.py 16000 2>&1 | grep -v amdgpu.ids > $O/phase_times.txt
timeout -k 10 200 python3 tools/step_hist.py 2>&1 | grep -v amdgpu.ids > $O/step_hist.txt
HS_LIB_PATH=$PWD/marl-hideandseek_amd/lib/libhideseek_timing.so timeout -k 10 200 python3 tools/phase_tail.py 16000 240 2>&1 | grep -v amdgpu.ids > $O/phase_tail.txt
timeout -k 10 300 python3 tools/train_config_bench.py > $O/train_config_bench.json 2>> $O/bench.err || echo "train_config_bench failed"
tail -3 $O/kstats.txt; cat $O/traffic_summary.txt; tail -3 $O/sq_counters.txt; cat $O/step_hist.txt
# agent-view renderer (opt-in): time per render and rocprofv3 kernel stats of the same command
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/kstats_render
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/kstats_render -- python3 tools/render_bench.py 16000 64 64 10 > $O/render_bench.txt 2>&1 || exit 1
grep -v amdgpu.ids $O/render_bench.txt | tail -2
cp $(ls gpurun_out/kstats_render/*/*kernel_stats.csv | head -1) $O/render_kernel_stats.csv
