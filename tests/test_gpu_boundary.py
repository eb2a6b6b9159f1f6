"""Drop-in boundary on the GPU: the tensor contract of src/mgr.cpp:1062-1331 and the call sequence
of scripts/benchmark.py (restated here — the reference tree does not exist on the GPU box)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sim(n=32, **kw):
    import gpu_hideseek
    args = dict(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=n, sim_flags=0, rand_seed=0,
                min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    args.update(kw)
    return gpu_hideseek.HideAndSeekSimulator(**args)


def test_tensor_shapes_dtypes_and_device():
    import torch
    N, A = 32, 4
    sim = _sim(N, enable_batch_renderer=True, batch_render_width=64, batch_render_height=64)
    sim.init()
    R = N * A
    expect = {
        "reset": ((N, 1), torch.int32), "done": ((R, 1), torch.int32), "prep_counter": ((R, 1), torch.int32),
        "action": ((R, 5), torch.int32), "reward": ((R, 1), torch.float32), "self_data": ((R, 13), torch.float32),
        "self_type": ((R, 1), torch.int32), "self_mask": ((R, 1), torch.float32),
        "agent_data": ((R, 5, 14), torch.float32), "box_data": ((R, 9, 17), torch.float32),
        "ramp_data": ((R, 2, 14), torch.float32), "visible_agents_mask": ((R, 5, 1), torch.float32),
        "visible_boxes_mask": ((R, 9, 1), torch.float32), "visible_ramps_mask": ((R, 2, 1), torch.float32),
        "lidar": ((R, 30), torch.float32), "seed": ((R, 2), torch.int32),
        "global_positions": ((N, 17, 2), torch.float32), "rgb": ((R, 64, 64, 4), torch.uint8),
        "depth": ((R, 64, 64, 1), torch.float32), "episode_result": ((N, 2), torch.float32),
        "policy_assignments": ((R, 1), torch.int32), "agent_mask": ((R, 1), torch.float32),
    }
    for name, (shape, dt) in expect.items():
        t = getattr(sim, name + "_tensor")().to_torch()
        assert tuple(t.shape) == shape and t.dtype == dt and t.is_cuda and t.is_contiguous(), name


def test_tensors_alias_simulator_memory():
    """The scripts mutate action/reset in place (benchmark.py:64-65,82-84): views must be zero-copy
    and persistent."""
    import torch
    sim = _sim(16, sim_flags=8)
    sim.init()
    a1 = sim.action_tensor().to_torch()
    a2 = sim.action_tensor().to_torch()
    assert a1.data_ptr() == a2.data_ptr()
    p0 = sim.self_data_tensor().to_torch()[:, :2].clone()
    a1[:, 0] = 4                      # +800 N in x for every agent (ZeroAgentVelocity buckets)
    for _ in range(5):
        sim.step()
        a1[:, 0] = 4
    p1 = sim.self_data_tensor().to_torch()[:, :2]
    hiders = sim.self_type_tensor().to_torch()[:, 0] == 1
    assert (torch.linalg.norm(p1 - p0, dim=1)[hiders] > 0.05).float().mean() > 0.8
    sim.set_action(0, 2, 2, 2, 0, 0)
    assert a1[0].tolist() == [2, 2, 2, 0, 0]
    sim.trigger_reset(3, 1)
    assert sim.reset_tensor().to_torch()[3, 0].item() == 1
    sim.step()
    assert sim.reset_tensor().to_torch().sum().item() == 0      # consumed (sim.cpp:185)


def test_benchmark_script_call_sequence():
    """Same calls as scripts/benchmark.py:21-92 at a small world count."""
    import torch
    import gpu_hideseek
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=256, sim_flags=0, rand_seed=0,
        min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1,
        enable_batch_renderer=True, batch_render_width=64, batch_render_height=64)
    sim.init()
    actions = sim.action_tensor().to_torch()
    resets = sim.reset_tensor().to_torch()
    assert actions.shape == (1024, 5) and resets.shape == (256, 1)
    rgb = sim.rgb_tensor().to_torch()
    assert rgb.shape == (1024, 64, 64, 4)
    move = actions[..., 0:2]
    move.copy_(torch.zeros_like(move))
    for _ in range(5):
        sim.step()
    for _ in range(20):
        sim.step()
        torch.randint(-5, 5, move.shape, out=move, dtype=torch.int32, device=torch.device("cuda"))
    assert torch.isfinite(sim.self_data_tensor().to_torch()).all()
    del sim


def test_step_async_on_a_caller_stream():
    """Manager::gpuJAXStep analogue (mgr.cpp:1006-1022): enqueue on a caller stream, no sync inside."""
    import torch
    a, b = _sim(64, rand_seed=4), _sim(64, rand_seed=4)
    a.init(); b.init()
    strm = torch.cuda.Stream()
    for _ in range(6):
        a.step()
        b.step_async(strm.cuda_stream)
    strm.synchronize()
    assert np.array_equal(a.debug_bodies()[0].view(np.int32), b.debug_bodies()[0].view(np.int32))


def test_skip_observations_extension_leaves_physics_unchanged(oracle=None):
    a, b = _sim(64, rand_seed=6), _sim(64, rand_seed=6, sim_flags=1 << 16)
    a.init(); b.init()
    for _ in range(20):          # inside the prep phase the observation side effect cannot matter
        a.step(); b.step()
    assert np.array_equal(a.debug_bodies()[0].view(np.int32), b.debug_bodies()[0].view(np.int32))
    assert b.lidar_tensor().to_torch().abs().sum().item() == 0


def _run_variant(env, steps=30, n=300):
    """Runs a fresh interpreter so that libhideseek picks the environment switches up at hs_create."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys, hashlib, numpy as np
sys.path.insert(0, {os.path.join(root, 'marl-hideandseek_amd')!r})
import torch, gpu_hideseek
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=1, gpu_id=0, num_worlds={n}, sim_flags=0, rand_seed=3, min_hiders=2,
      max_hiders=3, min_seekers=1, max_seekers=3, num_pbt_policies=1)
sim.init()
act = sim.action_tensor().to_torch()
for s in range({steps}):
    g = torch.arange(act.shape[0], device=act.device)
    act[:, 0] = ((g * 7 + s) % 11).int(); act[:, 1] = ((g * 3 + 2 * s) % 11).int(); act[:, 2] = ((g + s) % 11).int()
    act[:, 3] = ((g + s) % 13 == 0).int(); act[:, 4] = ((g * 2 + s) % 17 == 0).int()
    sim.step()
b, m = sim.debug_bodies()
h = hashlib.sha256(b.tobytes() + m.tobytes() + sim.lidar_tensor().to_torch().cpu().numpy().tobytes()
                   + sim.reward_tensor().to_torch().cpu().numpy().tobytes()).hexdigest()
print("DIGEST", h)
"""
    out = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    return [l for l in out.stdout.splitlines() if l.startswith("DIGEST")][0]


def test_graph_replay_and_eager_launch_agree():
    """The step replayed as HIP graphs and launched eagerly must give bit-identical state (the graph only
    changes how the kernels are submitted); so must a world count that is not a multiple of the physics
    kernel's worlds-per-workgroup."""
    ref = _run_variant({"HS_GRAPH": "1"})
    assert _run_variant({"HS_GRAPH": "0"}) == ref


def test_world_count_with_a_partial_octet_matches_the_oracle(oracle):
    """301 worlds = 37 octets + 5 worlds: the padding slots of the last octet must stay inert.  Same action stream as
    _run_variant (moves, turns, grabs and locks), 30 steps, every body and the exported tensors against the oracle."""
    import torch
    n = 301
    sim = _sim(n, rand_seed=3, min_hiders=2, max_hiders=3, min_seekers=1, max_seekers=3)
    ref = oracle.RefSim(n, rand_seed=3, min_hiders=2, max_hiders=3, min_seekers=1, max_seekers=3, threads=8)
    sim.init(); ref.init()
    act = sim.action_tensor().to_torch()
    g = np.arange(act.shape[0], dtype=np.int64)
    for s in range(30):
        a = np.stack([(g * 7 + s) % 11, (g * 3 + 2 * s) % 11, (g + s) % 11, ((g + s) % 13 == 0), ((g * 2 + s) % 17 == 0)],
                     axis=1).astype(np.int32)
        ref.tensor("action")[:] = a
        act.copy_(torch.from_numpy(a).to(act.device))
        sim.step(); ref.step()
    gb, gm = sim.debug_bodies()
    rb, rm = ref.bodies()
    assert np.array_equal(gm, rm) and np.array_equal(gb.view(np.int32), rb.view(np.int32))
    for k in ("lidar", "reward", "self_data", "box_data", "visible_agents_mask", "global_positions", "action"):
        got = getattr(sim, k + "_tensor")().to_torch().cpu().numpy().reshape(ref.tensor(k).shape)
        assert np.array_equal(np.ascontiguousarray(got).view(np.int32), ref.tensor(k).view(np.int32)), k


def test_headless_cpp_driver_runs():
    """marl-hideandseek_amd/tools/headless.cpp: the C ABI driven from C++ (reference src/headless.cpp)."""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "marl-hideandseek_amd", "lib", "headless")
    if not os.path.exists(exe):
        import build as hs_build
        hs_build.build_lib(); hs_build.build_headless()
    out = subprocess.run([exe, "CUDA", "512", "20", "--rand-actions"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    assert out.stdout.startswith("FPS ") and float(out.stdout.split()[1]) > 0
    bad = subprocess.run([exe, "CPU", "16", "1"], capture_output=True, text=True, timeout=60)
    assert bad.returncode != 0 and "CPU" in bad.stderr


def test_physics_only_65536_worlds():
    """BASELINE.json configs[2] (65 536 worlds, observations skipped): runs, stays finite, and a sampled world
    range matches a small simulator with the same global world ids bit for bit."""
    import torch
    N, steps, lo, n = 65536, 6, 40000, 32
    big = _sim(N, sim_flags=1 << 16)
    small = _sim(n, sim_flags=1 << 16, world_offset=lo)
    big.init(); small.init()
    ab, asm = big.action_tensor().to_torch(), small.action_tensor().to_torch()
    for s in range(steps):
        g = torch.arange(N * 4, device=ab.device)
        ab[:, 0] = ((g * 7 + s) % 10 - 5).int(); ab[:, 1] = ((g * 3 + 2 * s) % 10 - 5).int()
        asm[:, 0:2] = ab[lo * 4:(lo + n) * 4, 0:2]
        big.step(); small.step()
    bb, bm = big.debug_bodies()
    sb, sm_ = small.debug_bodies()
    assert np.isfinite(bb).all()
    assert np.array_equal(bb[lo:lo + n].view(np.int32), sb.view(np.int32)) and np.array_equal(bm[lo:lo + n], sm_)


def test_importing_the_package_before_torch_keeps_torch_working():
    """scripts/benchmark.py:1-2 imports gpu_hideseek first and torch second.  PyTorch-ROCm bundles its own HIP runtime;
    the package must not pull the system copy in ahead of it (torch would then report "No HIP GPUs")."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = f"""
import sys
sys.path.insert(0, {os.path.join(root, 'marl-hideandseek_amd')!r})
import gpu_hideseek
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=8, sim_flags=0,
      rand_seed=0, min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1, enable_batch_renderer=True)
import torch
sim.init()
actions = sim.action_tensor().to_torch()
rgb = sim.rgb_tensor().to_torch()
sim.step()
assert torch.cuda.is_available() and actions.is_cuda and tuple(rgb.shape) == (32, 64, 64, 4)
print("ORDER-OK")
"""
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "ORDER-OK" in out.stdout, out.stderr[-2000:]


def test_build_then_smoke_in_one_process():
    """The driver calls __graft_entry__.build() and smoke() in separate processes; in ONE process build() must not bring
    /opt/rocm's HIP runtime in ahead of torch's (its symbol and import checks run in child processes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.build(); g.smoke()"], cwd=root,
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0 and "smoke OK" in out.stdout, out.stderr[-2000:]
