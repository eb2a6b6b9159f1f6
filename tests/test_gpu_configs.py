"""BASELINE.json configs[3] and configs[4] on ONE GPU, against the oracle.

configs[3] = 131 072 worlds sharded over 8 GPUs: a rank is a 16 384-world handle with a `world_offset`; the last
rank (offset 114 688) is run here for 245 steps, i.e. across the lock-step level regeneration of step 240.
configs[4] = 16 000 worlds as scripts/jax_train.py drives them: flags RandomFlipTeams|UseFixedWorld|
ZeroAgentVelocity, seed 5, 3 hiders + 3 seekers, actions from the real bucket ranges [0,5)^3 x [0,2)^2
(scripts/jax_train.py:69-81,146-148), through the stream entry point with caller-owned buffers
(Manager::gpuJAXStep, src/mgr.cpp:379-398, 1006-1022).

Full-size runs are checked through size-independent properties (finite state, the done / prep-counter /
reward timeline, seed rows = {episode, GLOBAL world id}) AND against the oracle on EVERY world, bit for bit: the oracle
steps the whole batch beside the GPU on all host threads (no sampled ranges).  configs[1] — the
benchmark batch, 16 000 worlds x 245 steps across the lock-step regeneration — is here as well.
"""
import os

import numpy as np
import pytest


def _threads():
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    return max(1, min(n, 32))

pytestmark = pytest.mark.gpu

OBS = ["prep_counter", "self_data", "self_type", "self_mask", "lidar", "agent_data", "box_data", "ramp_data",
       "visible_agents_mask", "visible_boxes_mask", "visible_ramps_mask"]


def bits(a):
    a = np.ascontiguousarray(a)
    return a.view(np.int32) if a.dtype == np.float32 else a


def _hash(xp, g, s, c, mod):
    """Same integer hash in torch (int64 on the GPU) and numpy: action column c of global agent g at step s."""
    h = (g * 2654435761 + (s + 1) * 40503 * (c + 1)) & 0x7FFFFFFF
    return (h >> 8) % mod


def test_config3_last_shard_of_131072_worlds(oracle):
    import torch
    import gpu_hideseek
    N, A, OFF, STEPS = 16384, 4, 114688, 245
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
        min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1, world_offset=OFF)
    sim.init()
    act = sim.action_tensor().to_torch()
    # the oracle runs the WHOLE shard beside the GPU (every world, not sampled ranges)
    ranges = [(0, N)]
    refs = []
    for lo, n in ranges:
        r = oracle.RefSim(n, rand_seed=0, world_offset=OFF + lo, threads=_threads())
        r.init()
        refs.append(r)
    gdev = torch.arange(OFF * A, (OFF + N) * A, device=act.device, dtype=torch.int64)
    done = sim.done_tensor().to_torch()
    prep = sim.prep_counter_tensor().to_torch()
    reward = sim.reward_tensor().to_torch()
    for s in range(STEPS):
        act[:, 0] = (_hash(torch, gdev, s, 0, 10) - 5).to(torch.int32)
        act[:, 1] = (_hash(torch, gdev, s, 1, 10) - 5).to(torch.int32)
        for (lo, n), r in zip(ranges, refs):
            g = np.arange((OFF + lo) * A, (OFF + lo + n) * A, dtype=np.int64)
            r.tensor("action")[:, 0] = _hash(np, g, s, 0, 10) - 5
            r.tensor("action")[:, 1] = _hash(np, g, s, 1, 10) - 5
            r.step()
        sim.step()
        if s % 7 == 6:                 # every 7th step: body state and the observations that feed a policy, every world
            gb, gm = sim.debug_bodies()
            rb, rm = refs[0].bodies()
            assert np.array_equal(gm, rm) and np.array_equal(bits(gb), bits(rb)), s
            for k in ("self_data", "lidar", "box_data", "reward"):
                got = getattr(sim, k + "_tensor")().to_torch().cpu().numpy().reshape(refs[0].tensor(k).shape)
                assert np.array_equal(bits(got), bits(refs[0].tensor(k))), (s, k)
        # timeline (sim.cpp:806-841, 448-464): every world is in lock-step
        if s in (0, 50, 94, 95, 200, 238, 239, 240, 244):
            ep_step = s % 240          # curEpisodeStep the step ran with
            assert (done == (1 if ep_step == 239 else 0)).all(), s
            after = (ep_step + 1) % 240
            assert (prep == max(96 - after, 0)).all(), s
            if ep_step < 95:
                assert (reward == 0).all(), s
            else:
                assert (reward != 0).all(), s
    body, meta = sim.debug_bodies()
    assert np.isfinite(body).all()
    lidar = sim.lidar_tensor().to_torch()
    assert torch.isfinite(lidar).all() and (lidar >= 0).all() and (lidar <= 200).all()
    seed = sim.seed_tensor().to_torch().cpu().numpy().reshape(N, A, 2)
    assert (seed[:, :, 0] == 1).all(), "second episode after the regeneration at step 240"
    assert (seed[:, :, 1] == (OFF + np.arange(N))[:, None]).all(), "seed rows carry GLOBAL world ids (sim.cpp:107-111)"
    walls, info = sim.debug_walls()
    assert (info[:, 6] == STEPS % 240).all()
    for (lo, n), r in zip(ranges, refs):
        rb, rm = r.bodies()
        assert np.array_equal(bits(body[lo:lo + n]), bits(rb)) and np.array_equal(meta[lo:lo + n], rm), lo
        rw, ri = r.walls()
        assert np.array_equal(bits(walls[lo:lo + n]), bits(rw)) and np.array_equal(info[lo:lo + n], ri), lo
        for k in OBS + ["reward", "done", "global_positions", "episode_result", "seed"]:
            t = getattr(sim, k + "_tensor")().to_torch()
            rows = A if t.shape[0] == N * A else 1
            got = t[lo * rows:(lo + n) * rows].cpu().numpy().reshape(r.tensor(k).shape)
            assert np.array_equal(bits(got), bits(r.tensor(k))), (lo, k)
    assert sim.device_status()["dropped_candidate_pairs"] == 0


def test_config4_jax_train_workload_through_stream_step(oracle):
    import torch
    import gpu_hideseek
    N, A, STEPS, FLAGS, SEED = 16000, 6, 245, 13, 5
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=FLAGS, rand_seed=SEED,
        min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3, num_pbt_policies=1)
    assert sim.agents_per_world == A
    dev = torch.device("cuda", 0)
    R = N * A
    own = {k: getattr(sim, k + "_tensor")().to_torch() for k in OBS + ["reward", "done", "episode_result"]}
    obs = [torch.zeros_like(own[k]) for k in OBS]
    rew, done, epres = (torch.zeros_like(own[k]) for k in ("reward", "done", "episode_result"))
    act = torch.zeros(R, 5, dtype=torch.int32, device=dev)
    resets = torch.zeros(N, 1, dtype=torch.int32, device=dev)
    pol = torch.zeros(R, 1, dtype=torch.int32, device=dev)
    strm = torch.cuda.Stream()
    sim.stream_init(strm.cuda_stream, obs)
    ranges = [(0, N)]                            # the oracle runs every world beside the GPU
    refs = []
    for lo, n in ranges:
        r = oracle.RefSim(n, sim_flags=FLAGS, rand_seed=SEED, min_hiders=3, max_hiders=3, min_seekers=3,
                          max_seekers=3, world_offset=lo, threads=_threads())
        r.init()
        refs.append(r)
    gdev = torch.arange(R, device=dev, dtype=torch.int64)
    mods = (5, 5, 5, 2, 2)                       # actions_num_buckets (scripts/jax_train.py:146-148)
    grabbed = 0
    for s in range(STEPS):
        for c, m in enumerate(mods):
            act[:, c] = _hash(torch, gdev, s, c, m).to(torch.int32)
        for (lo, n), r in zip(ranges, refs):
            g = np.arange(lo * A, (lo + n) * A, dtype=np.int64)
            for c, m in enumerate(mods):
                r.tensor("action")[:, c] = _hash(np, g, s, c, m)
            r.step()
        strm.wait_stream(torch.cuda.current_stream())
        sim.stream_step(strm.cuda_stream, [act, resets, pol] + obs + [rew, done, epres])
        strm.synchronize()
        if s in (0, 94, 95, 150, 239, 240, 244):
            ep_step = s % 240
            assert (done == (1 if ep_step == 239 else 0)).all(), s
            assert (obs[0] == max(96 - (ep_step + 1) % 240, 0)).all(), s
            assert torch.isfinite(obs[1]).all() and torch.isfinite(obs[4]).all(), s
            assert ((rew == 0).all() if ep_step < 95 else (rew != 0).all()), s
            grabbed += int(obs[1][:, 12].sum().item())
            for (lo, n), r in zip(ranges, refs):
                for k, o in zip(OBS + ["reward", "done", "episode_result"], obs + [rew, done, epres]):
                    rows = A if o.shape[0] == R else 1
                    got = o[lo * rows:(lo + n) * rows].cpu().numpy().reshape(r.tensor(k).shape)
                    assert np.array_equal(bits(got), bits(r.tensor(k))), (s, lo, k)
    assert grabbed > 0, "grab joints were live at some checked step"
    body, meta = sim.debug_bodies()
    assert np.isfinite(body).all()
    # quaternions stay normalised in ZeroAgentVelocity mode (240 N m torques): |q|^2 within 1e-3 of 1
    q2 = (body[:, :, 3:7] ** 2).sum(axis=2)
    live = meta[:, :, 0] >= 0
    assert np.abs(q2[live] - 1).max() < 1e-3
    # UseFixedWorld: every world holds the same wall layout
    walls, info = sim.debug_walls()
    assert (walls == walls[0]).all() and (info[:, 0] == info[0, 0]).all()
    for (lo, n), r in zip(ranges, refs):
        rb, rm = r.bodies()
        assert np.array_equal(bits(body[lo:lo + n]), bits(rb)) and np.array_equal(meta[lo:lo + n], rm), lo
    assert sim.device_status()["dropped_candidate_pairs"] == 0


def test_config1_benchmark_batch_every_world_against_the_oracle(oracle):
    """BASELINE.json configs[1]: 16 000 worlds, 2 hiders + 2 seekers, flags 0, seed 0, move actions in [-5, 4] on columns
    0-1 (scripts/benchmark.py:21-35, 82-84), 245 steps — across the regeneration of every level at step 240.  Every exported
    tensor, body and wall of every world against the oracle at every 7th step and at the end."""
    import torch
    import gpu_hideseek
    N, A, STEPS = 16000, 4, 245
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
        min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    ref = oracle.RefSim(N, rand_seed=0, threads=_threads())
    sim.init(); ref.init()
    act = sim.action_tensor().to_torch()
    gdev = torch.arange(N * A, device=act.device, dtype=torch.int64)
    gnp = np.arange(N * A, dtype=np.int64)
    names = OBS + ["reward", "done", "global_positions", "episode_result", "seed", "action"]
    checked = 0
    for s in range(STEPS):
        for c in (0, 1):
            act[:, c] = (_hash(torch, gdev, s, c, 10) - 5).to(torch.int32)
            ref.tensor("action")[:, c] = _hash(np, gnp, s, c, 10) - 5
        sim.step(); ref.step()
        if s % 7 == 6 or s == STEPS - 1:
            gb, gm = sim.debug_bodies()
            rb, rm = ref.bodies()
            assert np.array_equal(gm, rm), s
            bad = np.argwhere(bits(gb) != bits(rb))
            assert bad.size == 0, (s, bad[:4].tolist())
            gw, gi = sim.debug_walls()
            rw, ri = ref.walls()
            assert np.array_equal(gi, ri) and np.array_equal(bits(gw), bits(rw)), s
            for k in names:
                got = getattr(sim, k + "_tensor")().to_torch().cpu().numpy().reshape(ref.tensor(k).shape)
                assert np.array_equal(bits(got), bits(ref.tensor(k))), (s, k)
            checked += 1
    assert checked == 35
    st = sim.device_status()
    assert st["dropped_candidate_pairs"] == 0
