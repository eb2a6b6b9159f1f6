"""bench.py's rank plumbing: `python bench.py --gpus N` with WORLD_SIZE unset starts N ranks by
itself — the launcher process touches no GPU —, relays rank 0's line, and fails loudly when a rank cannot run.  On the CPU
box the ranks run bench.py's stub (HS_BENCH_STUB=1: the same process-group / barrier / max-reduce / gather code around a
sleep); the -m gpu tests run the real thing on the one GPU of the test box."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _env(**kw):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT", "MASTER_ADDR")}
    env.update(kw)
    return env


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def test_launcher_starts_n_ranks_and_relays_rank0_line():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "3", "--steps", "4", "--worlds-per-gpu", "16384"],
                       env=_env(HS_BENCH_STUB="1"), capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, r.stderr
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1                                   # ONE line, rank 0's
    out = json.loads(lines[0])
    assert out["stub"] and out["n_gpus"] == 3 and out["world_size"] == 3
    assert out["env"]["RANK"] == "0" and out["env"]["WORLD_SIZE"] == "3" and out["env"]["MASTER_ADDR"] == "127.0.0.1"
    assert out["world_offsets"] == [0, 16384, 32768]         # contiguous global world ranges
    assert len(out["ms_per_step_per_rank"]) == 3
    # max over ranks: rank r sleeps 10 (r + 1) ms
    assert abs(out["ms_per_step"] - max(out["ms_per_step_per_rank"])) < 1e-9
    assert out["ms_per_step_per_rank"][2] > out["ms_per_step_per_rank"][0]


def test_torchrun_path_still_works():
    """The driver's way for N > 1: python -m torch.distributed.run ... bench.py --gpus N."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                        "--master-addr", "127.0.0.1", "--master-port", str(_free_port()), BENCH, "--gpus", "2", "--steps", "2"],
                       env=_env(HS_BENCH_STUB="1"), capture_output=True, text=True, timeout=180)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["world_size"] == 2


def test_gpus_flag_must_match_the_launchers_world_size():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "2"],
                       env=_env(HS_BENCH_STUB="1", RANK="0", LOCAL_RANK="0", WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                                MASTER_PORT=str(_free_port())), capture_output=True, text=True, timeout=60)
    assert r.returncode != 0 and "WORLD_SIZE=2" in r.stderr


def test_missing_devices_fail_loudly_not_silently():
    """No stub: on a box with fewer GPUs than ranks the job must fail, never print a line with a smaller n_gpus."""
    import torch
    ndev = torch.cuda.device_count()
    n = ndev + 2
    r = subprocess.run([sys.executable, BENCH, "--gpus", str(n), "--steps", "2", "--no-cpu-baseline"], env=_env(),
                       capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert "failed" in r.stderr


@pytest.mark.gpu
def test_two_ranks_on_a_one_gpu_box_fail_with_device_not_visible():
    import torch
    if torch.cuda.device_count() != 1:
        pytest.skip("needs exactly one visible GPU")
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "8", "--no-cpu-baseline"], env=_env(),
                       capture_output=True, text=True, timeout=600)
    assert r.returncode != 0
    assert "device 1 not visible" in r.stderr
    assert not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


@pytest.mark.gpu
def test_one_rank_through_the_launcher_path():
    r = subprocess.run([sys.executable, BENCH, "--gpus", "1", "--steps", "16", "--warmup", "2", "--worlds-per-gpu", "2048",
                        "--no-cpu-baseline"], env=_env(HS_BENCH_FORCE_LAUNCHER="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 1 and out["config"]["total_worlds"] == 2048 and out["value"] > 0
    assert out["device_status"]["dropped_candidate_pairs"] == 0
    assert len(out["ms_per_step_per_rank"]) == 1 and "p50" in out["ms_per_step_percentiles"]
    assert "traffic_source" in out["roofline"]


@pytest.mark.gpu
def test_two_real_ranks_rehearsed_on_one_gpu():
    """The whole N > 1 path with the product in it: HS_BENCH_REHEARSE=1 maps both ranks to GPU 0 and uses gloo for the
    barrier / reductions (RCCL refuses two ranks on one device).  Two simulators with world offsets 0 and 1024 step side
    by side; the line counts both ranks and both shards' worlds.  Not a scaling number — plumbing with real simulators."""
    r = subprocess.run([sys.executable, BENCH, "--gpus", "2", "--steps", "12", "--warmup", "2", "--worlds-per-gpu", "1024",
                        "--no-cpu-baseline"], env=_env(HS_BENCH_REHEARSE="1"), capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    out = json.loads([ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1])
    assert out["n_gpus"] == 2 and out["config"]["total_worlds"] == 2048 and out["scaling"] == "weak"
    assert len(out["ms_per_step_per_rank"]) == 2 and out["cpu_baseline"] is None if "cpu_baseline" in out else True
    assert abs(out["value"] - 2048 * 12 / (out["ms_per_step"] * 12 * 1e-3)) < 1e-3 * out["value"]
