"""The Python face as the reference's five scripts use it.  The import and constructor statements below are the
scripts' own (restated here: the reference tree does not travel), executed literally against this package.
Without a GPU the constructor must get as far as probing the device and then fail loudly (no fallback) —
that proves every keyword bound; with a GPU (`-m gpu`, test_gpu_boundary.py) the same statements run for real.
"""
import types

import pytest

# (script, statements up to and including the constructor; argparse results restated as a namespace)
SCRIPTS = {
    # scripts/benchmark.py:1-2,21-35 (argv: 16000 1920 0 0 1)
    "benchmark.py": """
import gpu_hideseek
import torch
num_worlds, num_steps, entities_per_world, reset_chance = 16, 4, 0, 0.0
sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode = gpu_hideseek.madrona.ExecMode.CUDA,
        gpu_id = 0,
        num_worlds = num_worlds,
        sim_flags = 0,
        rand_seed = 0,
        min_hiders = 2,
        max_hiders = 2,
        min_seekers = 2,
        max_seekers = 2,
        num_pbt_policies = 1,
        enable_batch_renderer = True,
        batch_render_width = 64,
        batch_render_height = 64,
)
""",
    # scripts/cpu_benchmark.py:1-2,21-35
    "cpu_benchmark.py": """
import gpu_hideseek
import torch
num_worlds = 16
sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode = gpu_hideseek.madrona.ExecMode.CPU,
        gpu_id = 0,
        num_worlds = num_worlds,
        sim_flags = 0,
        rand_seed = 0,
        min_hiders = 2,
        max_hiders = 2,
        min_seekers = 2,
        max_seekers = 2,
        num_pbt_policies = 1,
        enable_batch_renderer = False,
        batch_render_width = 64,
        batch_render_height = 64,
)
""",
    # scripts/jax_train.py:14-16,69-81
    "jax_train.py": """
import gpu_hideseek
from gpu_hideseek import SimFlags
from gpu_hideseek.madrona import ExecMode
sim = gpu_hideseek.HideAndSeekSimulator(
    exec_mode = ExecMode.CUDA if args.gpu_sim else ExecMode.CPU,
    gpu_id = args.gpu_id,
    num_worlds = args.num_worlds,
    sim_flags = SimFlags.RandomFlipTeams | SimFlags.UseFixedWorld | SimFlags.ZeroAgentVelocity,
    min_hiders = args.num_hiders,
    max_hiders = args.num_hiders,
    min_seekers = args.num_seekers,
    max_seekers = args.num_seekers,
    num_pbt_policies = args.pbt_ensemble_size,
    rand_seed = 5,
)
""",
    # scripts/jax_infer.py:11-13,66-79
    "jax_infer.py": """
import gpu_hideseek
from gpu_hideseek import SimFlags
from gpu_hideseek.madrona import ExecMode
num_policies = 1
sim = gpu_hideseek.HideAndSeekSimulator(
    exec_mode = ExecMode.CUDA if args.gpu_sim else ExecMode.CPU,
    gpu_id = args.gpu_id,
    num_worlds = args.num_worlds,
    num_pbt_policies = num_policies if num_policies > 1 else 1,
    rand_seed = 5,
    sim_flags = SimFlags.UseFixedWorld | SimFlags.ZeroAgentVelocity,
    min_hiders = args.num_hiders,
    max_hiders = args.num_hiders,
    min_seekers = args.num_seekers,
    max_seekers = args.num_seekers,
)
""",
    # scripts/jax_policy.py and scripts/common.py never touch the simulator module (they import jax / flax /
    # madrona_learn only); their part of the face is the observation names checked below.
}

ARGS = types.SimpleNamespace(gpu_sim=True, gpu_id=0, num_worlds=16, num_hiders=3, num_seekers=3, pbt_ensemble_size=0)

# Manager::trainInterface, src/mgr.cpp:1338-1375 — the names scripts/jax_policy.py reads its observations by
OBS_NAMES = ["prep_counter", "self_data", "self_type", "self_mask", "self_lidar", "agent_data", "box_data", "ramp_data",
             "vis_agents_mask", "vis_boxes_mask", "vis_ramps_mask"]


@pytest.mark.parametrize("script", sorted(SCRIPTS))
def test_script_import_and_constructor_lines(script, hideseek_lib):
    import torch
    ns = {"args": ARGS}
    if script == "cpu_benchmark.py":
        # exec_mode CPU: refused, never emulated (there is no CPU execution path in the product)
        with pytest.raises(NotImplementedError):
            exec(SCRIPTS[script], ns)
        return
    if torch.cuda.is_available():
        exec(SCRIPTS[script], ns)
        assert ns["sim"].agents_per_world in (4, 6)
        ns["sim"].close()
    else:
        with pytest.raises(RuntimeError, match="no HIP device"):
            exec(SCRIPTS[script], ns)


def test_madrona_is_a_real_submodule():
    import importlib
    import sys
    m = importlib.import_module("gpu_hideseek.madrona")
    assert isinstance(m, types.ModuleType) and sys.modules["gpu_hideseek.madrona"] is m
    import gpu_hideseek
    assert gpu_hideseek.madrona is m and m.ExecMode.CUDA == 1 and m.ExecMode.CPU == 0
    assert m.Tensor is gpu_hideseek.Tensor


def test_train_interface_names_roles_and_order(hideseek_lib):
    import gpu_hideseek
    tab = gpu_hideseek.train_interface()
    names = [n for n, _, _ in tab]
    assert names == ["actions", "resets", "sim_ctrl", "policy_assignments"] + OBS_NAMES + [
        "rewards", "dones", "episode_results", "checkpoint_data"]
    by = {n: (r, g) for n, r, g in tab}
    assert [n for n, r, _ in tab if r == "observations"] == OBS_NAMES
    assert by["self_lidar"] == ("observations", "lidar_tensor")
    assert by["vis_agents_mask"] == ("observations", "visible_agents_mask_tensor")
    assert by["vis_boxes_mask"][1] == "visible_boxes_mask_tensor" and by["vis_ramps_mask"][1] == "visible_ramps_mask_tensor"
    assert by["actions"] == ("actions", "action_tensor") and by["resets"] == ("resets", "reset_tensor")
    assert by["sim_ctrl"] == ("sim_ctrl", None)
    assert by["policy_assignments"] == ("pbt_inputs", "policy_assignments_tensor")
    assert by["rewards"] == ("rewards", "reward_tensor") and by["dones"] == ("dones", "done_tensor")
    assert by["episode_results"] == ("pbt_outputs", "episode_result_tensor")
    assert by["checkpoint_data"] == ("checkpoint_data", "ckpt_tensor")
    for _, _, g in tab:
        assert g is None or callable(getattr(gpu_hideseek.HideAndSeekSimulator, g))
    # the observation block is also the buffer order of the stream entry points (mgr.cpp:183-197, 351-362)
    from test_gpu_configs import OBS
    assert [by[n][1][:-len("_tensor")] for n in OBS_NAMES] == OBS


def test_xla_targets_are_exported_in_xlas_abi(hideseek_lib):
    """bindings.cpp:97-118: the four custom-call targets exist as C symbols with XLA's signature and the Python face
    names them as the reference's jax() dict does ('save_ckpts' is read at scripts/jax_infer.py:137)."""
    import ctypes as C
    from gpu_hideseek import _native
    assert set(_native.XLA_TARGETS) == {"init", "step", "save_ckpts", "load_ckpts"}
    L = C.CDLL(hideseek_lib)
    for sym in _native.XLA_TARGETS.values():
        assert hasattr(L, sym)
    L.hs_xla_last_status.restype = C.c_int32
    # a call without a descriptor touches no GPU: it is recorded as an invalid-argument failure
    L.hs_xla_step.restype = None
    L.hs_xla_step.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p, C.c_size_t]
    L.hs_xla_step(None, None, None, 0)
    assert L.hs_xla_last_status(1) == 1 and L.hs_xla_last_status(0) == 0


def test_shard_index_arithmetic():
    from gpu_hideseek.sharded import locate, shard_ranges
    r = shard_ranges(131072, 8)
    assert r == [(g * 16384, 16384) for g in range(8)]
    r = shard_ranges(16000, 3)
    assert r == [(0, 5334), (5334, 5333), (10667, 5333)] and sum(n for _, n in r) == 16000
    assert locate(r, 0) == (0, 0) and locate(r, 5333) == (0, 5333) and locate(r, 5334) == (1, 0)
    assert locate(r, 15999) == (2, 5332)
    with pytest.raises(ValueError):
        locate(r, 16000)
    with pytest.raises(ValueError):
        shard_ranges(3, 4)
