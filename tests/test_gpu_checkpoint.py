"""Checkpoint save / load (Checkpoint src/sim.hpp:283-313, systems src/sim.cpp:956-1137,
Manager::saveCheckpoint/loadCheckpoint/loadCheckpoints src/mgr.cpp:905-985) and the stream entry points
behind sim.jax() (src/mgr.cpp:362-436): HIP path through the C ABI vs the CPU oracle, bit for bit."""
import numpy as np
import pytest

from test_gpu_parity import NAMES, assert_equal_state, bits, drive, make_pair

pytestmark = pytest.mark.gpu


def _ckpt_views(sim):
    ctrl = sim.ckpt_ctrl_tensor().to_torch()
    ck = sim.ckpt_tensor().to_torch()
    return ctrl, ck


def test_checkpoint_tensor_contract(oracle):
    import torch
    sim, ref, gt = make_pair(oracle, 8)
    ctrl, ck = _ckpt_views(sim)
    assert ctrl.dtype == torch.uint8 and tuple(ctrl.shape) == (8, 4)          # mgr.cpp:1209-1217
    assert ck.dtype == torch.uint8 and tuple(ck.shape) == (8, 1392)           # mgr.cpp:1219-1227
    assert int(ctrl.sum()) == 0                                               # Sim::Sim sim.cpp:1391-1393


@pytest.mark.parametrize("cfg", [dict(n=40, flags=13, seed=5, hiders=(3, 3), seekers=(3, 3)),
                                 dict(n=48, flags=0, seed=11, hiders=(1, 3), seekers=(1, 2))])
def test_save_load_parity_with_oracle(oracle, cfg):
    """Grab joints and locks live (full action buckets): saved records, restored state and the
    trajectory after a restore all match the oracle."""
    import torch
    n = cfg["n"]
    sim, ref, gt = make_pair(oracle, n, flags=cfg["flags"], seed=cfg["seed"], hiders=cfg["hiders"], seekers=cfg["seekers"])
    ctrl, ck = _ckpt_views(sim)
    drive(sim, ref, gt, 60, "full", check_every=20, seed=1)
    # --- save every world
    ctrl.view(torch.int32)[:] = 1
    ref.tensor("ckpt_ctrl")[:] = 1
    sim.save_checkpoints(); ref.save_checkpoints()
    assert int(ctrl.view(torch.int32).abs().sum()) == 0 and not ref.tensor("ckpt_ctrl").any()
    g_ck = ck.cpu().numpy()
    assert np.array_equal(g_ck, ref.tensor("ckpt")), "checkpoint records differ"
    rec = g_ck.view(np.int32).reshape(n, 348)
    assert (rec[:, 4] == 60).all()                                  # episodeStep
    grab_idx = rec[:, 5 + 13::29][:, :6]                            # agents[i].grabIdx
    assert (grab_idx >= 0).any(), "no live grab joint in the fixture: weak test"
    # --- move on, then restore half of the worlds
    drive(sim, ref, gt, 15, "full", check_every=15, seed=2)
    trig = (np.arange(n) % 2 == 0).astype(np.int32)
    ctrl.view(torch.int32)[:, 0] = torch.from_numpy(trig).to(ctrl.device)
    ref.tensor("ckpt_ctrl")[:, 0] = trig
    sim.load_checkpoints(); ref.load_checkpoints()
    assert np.array_equal(ctrl.view(torch.int32).cpu().numpy().ravel(), trig)     # trigger stays 1 (sim.cpp:963)
    assert_equal_state(sim, ref, gt, "after load")
    _, info = sim.debug_walls()
    assert (info[trig == 1, 6] == 60).all() and (info[trig == 0, 6] == 75).all()
    # --- the restored worlds keep simulating identically (joints, locks, RNG state restored)
    ctrl.zero_(); ref.tensor("ckpt_ctrl")[:] = 0
    drive(sim, ref, gt, 40, "full", check_every=10, seed=3)


def test_restore_replays_the_same_trajectory(oracle):
    """Size-independent property: save, run k steps, load, re-run the same actions -> identical outputs."""
    import torch
    sim, ref, gt = make_pair(oracle, 512, seed=21)
    ref.close()
    ctrl, ck = _ckpt_views(sim)
    rng = np.random.default_rng(5)
    acts = torch.from_numpy(rng.integers(0, 11, size=(30, 512 * 4, 3)).astype(np.int32)).cuda()
    for t in range(10):
        gt["action"][:, :3] = acts[t]; sim.step()
    ctrl.view(torch.int32)[:] = 1
    sim.save_checkpoints()
    at_save = {k: gt[k].clone() for k in NAMES if k != "action"}

    def run():
        out = []
        for t in range(10, 30):
            gt["action"][:, :3] = acts[t]; sim.step()
            out.append(torch.cat([gt[k].reshape(-1).view(torch.int32) for k in ("self_data", "lidar", "reward", "box_data")]).clone())
        return torch.stack(out)
    first = run()
    bodies_end = sim.debug_bodies()[0].copy()
    ctrl.view(torch.int32)[:] = 1
    sim.load_checkpoints()
    for k in ("self_data", "lidar", "agent_data", "box_data", "ramp_data", "prep_counter", "global_positions"):
        assert torch.equal(gt[k].view(torch.int32), at_save[k].view(torch.int32)), f"{k} not restored"
    ctrl.zero_()
    second = run()
    assert torch.equal(first, second)
    assert np.array_equal(bits(bodies_end), bits(sim.debug_bodies()[0]))


def test_single_world_save_and_load(oracle):
    import torch
    sim, ref, gt = make_pair(oracle, 16, seed=2)
    ctrl, ck = _ckpt_views(sim)
    drive(sim, ref, gt, 5, "bench", check_every=5)
    sim.save_checkpoint(3)
    rec = ck.cpu().numpy()
    assert rec[3].any() and not rec[np.arange(16) != 3].any()
    ref.tensor("ckpt_ctrl")[3] = 1; ref.save_checkpoints()
    assert np.array_equal(rec, ref.tensor("ckpt"))
    drive(sim, ref, gt, 5, "bench", check_every=5)
    sim.load_checkpoint(3)
    ref.tensor("ckpt_ctrl")[3] = 1; ref.load_checkpoints()
    assert_equal_state(sim, ref, gt, "single-world load")
    with pytest.raises(ValueError):
        sim.save_checkpoint(16)


def test_stream_entry_points_match_blocking_api(oracle):
    """hs_jax_init/step/save/load with caller-owned buffers on a caller-owned stream == init/step/... with the
    aliased tensors (buffer order of mgr.cpp:168-201, 379-398)."""
    import torch
    import gpu_hideseek
    kw = dict(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=64, sim_flags=0, rand_seed=4,
              min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    a = gpu_hideseek.HideAndSeekSimulator(**kw)
    b = gpu_hideseek.HideAndSeekSimulator(**kw)
    obs_names = ["prep_counter", "self_data", "self_type", "self_mask", "lidar", "agent_data", "box_data", "ramp_data",
                 "visible_agents_mask", "visible_boxes_mask", "visible_ramps_mask"]
    at = {k: getattr(a, k + "_tensor")().to_torch() for k in NAMES}
    bt = {k: getattr(b, k + "_tensor")().to_torch() for k in NAMES}
    obs = [torch.zeros_like(bt[k]) for k in obs_names]
    rew, done, epres = torch.zeros_like(bt["reward"]), torch.zeros_like(bt["done"]), torch.zeros_like(bt["episode_result"])
    strm = torch.cuda.Stream()
    a.init()
    b.stream_init(strm.cuda_stream, obs)
    for k, o in zip(obs_names, obs):
        assert torch.equal(o.view(torch.int32), at[k].view(torch.int32)), k
    rng = np.random.default_rng(0)
    for t in range(6):
        act = torch.from_numpy(rng.integers(0, 11, size=(256, 5)).astype(np.int32)).cuda()
        act[:, 3:] = 0
        resets = torch.zeros(64, 1, dtype=torch.int32, device="cuda")
        if t == 3:
            resets[5] = 1
        pol = torch.zeros(256, 1, dtype=torch.int32, device="cuda")
        at["action"].copy_(act); at["reset"].copy_(resets)
        a.step()
        strm.wait_stream(torch.cuda.current_stream())
        b.stream_step(strm.cuda_stream, [act, resets, pol] + obs + [rew, done, epres])
        strm.synchronize()
        for k, o in zip(obs_names + ["reward", "done", "episode_result"], obs + [rew, done, epres]):
            assert torch.equal(o.view(torch.int32), at[k].view(torch.int32)), (t, k)
    # checkpoints through the stream functions
    ctrl = torch.ones(64, 1, dtype=torch.int32, device="cuda")
    ckpts = torch.zeros(64, 1392, dtype=torch.uint8, device="cuda")
    b.stream_save_checkpoints(strm.cuda_stream, [ctrl, ckpts]); strm.synchronize()
    a.ckpt_ctrl_tensor().to_torch().view(torch.int32)[:] = 1
    a.save_checkpoints()
    assert torch.equal(ckpts, a.ckpt_tensor().to_torch())
    for s in (a, b):
        s.step()
    b.stream_load_checkpoints(strm.cuda_stream, [ctrl, ckpts] + obs); strm.synchronize()
    a.ckpt_ctrl_tensor().to_torch().view(torch.int32)[:] = 1
    a.load_checkpoints()
    for k, o in zip(obs_names, obs):
        assert torch.equal(o.view(torch.int32), at[k].view(torch.int32)), k
    assert np.array_equal(bits(a.debug_bodies()[0]), bits(b.debug_bodies()[0]))
    with pytest.raises(ValueError):
        b.stream_step(strm.cuda_stream, obs)


def test_xla_custom_call_targets_match_blocking_api(oracle):
    """The four hs_xla_* targets, called the way XLA calls a GPU custom call — target(stream, buffers, opaque,
    opaque_len) with the address taken out of the "xla._CUSTOM_CALL_TARGET" capsule `sim.jax()` would register and
    the handle as the opaque descriptor — give the blocking API's results; a call without a valid descriptor is
    recorded (the ABI cannot return it) and a failure on a handle surfaces at its next blocking call."""
    import ctypes as C
    import torch
    import gpu_hideseek
    kw = dict(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=40, sim_flags=13, rand_seed=5,
              min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3, num_pbt_policies=1)
    a = gpu_hideseek.HideAndSeekSimulator(**kw)
    b = gpu_hideseek.HideAndSeekSimulator(**kw)
    with pytest.raises(NotImplementedError) as ei:          # jaxlib is not importable: the bundle travels on the error
        b.jax(True)
    xla = ei.value.xla
    assert set(xla["targets"]) == {"init", "step", "save_ckpts", "load_ckpts"} and len(xla["opaque"]) == 8
    get_ptr = C.pythonapi.PyCapsule_GetPointer
    get_ptr.restype = C.c_void_p
    get_ptr.argtypes = [C.py_object, C.c_char_p]
    proto = C.CFUNCTYPE(None, C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t)
    fn = {k: proto(get_ptr(cap, b"xla._CUSTOM_CALL_TARGET")) for k, cap in xla["targets"].items()}

    def call(name, stream, bufs, opaque=None):
        arr = (C.c_void_p * len(bufs))(*[int(t.data_ptr()) for t in bufs])
        op = xla["opaque"] if opaque is None else opaque
        fn[name](C.c_void_p(stream.cuda_stream), arr, op, len(op))

    sig = xla["signatures"]
    dt = {"int32": torch.int32, "float32": torch.float32, "uint8": torch.uint8}
    mk = lambda lst: [torch.zeros(shape, dtype=dt[d], device="cuda") for _n, shape, d in lst]
    obs = mk(sig["init"]["results"])
    obs_names = [n for n, _s, _d in sig["init"]["results"]]
    iface_a = a.train_interface()["observations"]
    strm = torch.cuda.Stream()
    a.init()
    call("init", strm, obs)
    for n, o in zip(obs_names, obs):
        assert torch.equal(o.view(torch.int32), iface_a[n].to_torch().view(torch.int32)), n
    ins = mk(sig["step"]["operands"]); tail = mk(sig["step"]["results"][len(obs):])
    rng = np.random.default_rng(2)
    for t in range(5):
        act = np.concatenate([rng.integers(0, 5, size=(240, 3)), rng.integers(0, 2, size=(240, 2))], axis=1).astype(np.int32)
        ins[0].copy_(torch.from_numpy(act).cuda())
        a.action_tensor().to_torch().copy_(ins[0])
        a.step()
        strm.wait_stream(torch.cuda.current_stream())
        call("step", strm, ins + obs + tail)
        strm.synchronize()
        for n, o in zip(obs_names, obs):
            assert torch.equal(o.view(torch.int32), iface_a[n].to_torch().view(torch.int32)), (t, n)
        assert torch.equal(tail[0].view(torch.int32), a.reward_tensor().to_torch().view(torch.int32))
        assert torch.equal(tail[1], a.done_tensor().to_torch())
    ck = mk(sig["save_ckpts"]["operands"]) + mk(sig["save_ckpts"]["results"])
    ck[0].fill_(1)
    call("save_ckpts", strm, ck); strm.synchronize()
    a.ckpt_ctrl_tensor().to_torch().view(torch.int32)[:] = 1
    a.save_checkpoints()
    assert torch.equal(ck[1], a.ckpt_tensor().to_torch())
    for s in (a, b):
        s.step()
    call("load_ckpts", strm, ck + obs); strm.synchronize()
    a.ckpt_ctrl_tensor().to_torch().view(torch.int32)[:] = 1
    a.load_checkpoints()
    for n, o in zip(obs_names, obs):
        assert torch.equal(o.view(torch.int32), iface_a[n].to_torch().view(torch.int32)), n
    assert np.array_equal(bits(a.debug_bodies()[0]), bits(b.debug_bodies()[0]))
    # no status channel in the ABI: failures are kept
    L = b._L
    assert L.hs_xla_last_status(1) == 0
    call("step", strm, ins + obs + tail, opaque=b"\0" * 4)           # malformed descriptor
    assert L.hs_xla_last_status(1) == 1 and L.hs_xla_last_status(0) == 0


def test_replay_log_round_trip(oracle, tmp_path):
    """The replay-log format of scripts/jax_infer.py:125 / src/viewer.cpp:13-26,185-215: record a run, play it back
    in a fresh simulator, and get the recorded body state and observations at every step."""
    import torch
    import gpu_hideseek
    from gpu_hideseek import replay
    n = 24
    sim, ref, gt = make_pair(oracle, n, seed=6)
    ref.close()
    path = tmp_path / "run.log"
    rng = np.random.default_rng(3)
    seen = []
    with open(path, "wb") as f:
        for t in range(12):
            gt["action"][:, :3] = torch.from_numpy(rng.integers(0, 11, size=(n * 4, 3)).astype(np.int32)).cuda()
            sim.step()
            replay.record_step(sim, f)
            seen.append((sim.debug_bodies()[0].copy(), gt["self_data"].clone(), gt["lidar"].clone(), gt["global_positions"].clone()))
    log = replay.read_log(str(path), n)
    assert log.shape == (12, n, 1392)
    player = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=n, sim_flags=0, rand_seed=6, min_hiders=2,
        max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    player.init()
    for t in (0, 5, 11, 3):
        replay.replay_step(player, log, t)
        bodies, self_data, lidar, gpos = seen[t]
        assert np.array_equal(bits(player.debug_bodies()[0]), bits(bodies)), t
        assert torch.equal(player.self_data_tensor().to_torch().view(torch.int32), self_data.view(torch.int32)), t
        assert torch.equal(player.lidar_tensor().to_torch().view(torch.int32), lidar.view(torch.int32)), t
        assert torch.equal(player.global_positions_tensor().to_torch().view(torch.int32), gpos.view(torch.int32)), t


def test_unbounded_episode_step_does_not_corrupt_the_prep_counter(oracle):
    """under IgnoreEpisodeLength the step counter grows without bound (src/sim.cpp:196).  k_observe reads it from
    a 16-bit field of the slot header; a step of 40 000 (planted through a checkpoint) must leave prep_counter alone, as
    the reference does after step 96 (src/sim.cpp:461-463), instead of wrapping into the preparation phase again."""
    import torch
    n = 16
    sim, ref, gt = make_pair(oracle, n, flags=2, seed=4)
    ctrl, ck = _ckpt_views(sim)
    drive(sim, ref, gt, 3, "bench", check_every=3)
    ctrl.view(torch.int32)[:] = 1
    ref.tensor("ckpt_ctrl")[:] = 1
    sim.save_checkpoints(); ref.save_checkpoints()
    for step in (32767, 40000, 70000):
        rec = ck.view(torch.int32).view(n, 348)
        rec[:, 4] = step                                             # Checkpoint::episodeStep
        ref.tensor("ckpt").view(np.int32).reshape(n, 348)[:, 4] = step
        ctrl.view(torch.int32)[:] = 1
        ref.tensor("ckpt_ctrl")[:] = 1
        sim.load_checkpoints(); ref.load_checkpoints()
        ctrl.zero_(); ref.tensor("ckpt_ctrl")[:] = 0
        assert_equal_state(sim, ref, gt, f"loaded at step {step}")
        drive(sim, ref, gt, 2, "bench", check_every=1)
        _, info = sim.debug_walls()
        assert (info[:, 6] == step + 2).all()
