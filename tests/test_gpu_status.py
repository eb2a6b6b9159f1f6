"""Device-side conditions are surfaced (include/hideseek.h hs_device_status): broadphase candidate pairs beyond the LDS
capacities take the spill path (counted, never dropped, results identical), HS_GRAPH=1 can be seen to be in use; and the
multi-handle front-end (ShardedSimulator)."""
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "marl-hideandseek_amd")


def _child(code, env=None, timeout=300):
    out = subprocess.run([sys.executable, "-c", f"import sys\nsys.path.insert(0, {PKG!r})\n" + code],
                         env={**os.environ, **(env or {})}, capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, out.stderr[-3000:]
    return out.stdout


STEP_LOOP = """
import torch, gpu_hideseek
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=1, gpu_id=0, num_worlds=256, sim_flags=0, rand_seed=3, min_hiders=2,
      max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
sim.init()
for s in range(12):
    sim.step()
st = sim.device_status()
print("STATUS", st["spilled_dd_pairs"], st["spilled_static_pairs"], int(st["graphs_in_use"]), st["dropped_candidate_pairs"])
"""

# Oracle parity of a whole run in a child process (so that HS_LIB_PATH can select the library): every exported tensor
# the physics feeds and the body state, bit for bit, at several steps.  hiders / seekers 3 + 3 with grab / lock actions
# exercises joints, locked (static) boxes and the third round of bodies together with the spill path.
PARITY_LOOP = """
import sys, hashlib, numpy as np, torch, gpu_hideseek
sys.path.insert(0, %r)
import hs_ref
N, H, K, FLAGS, STEPS = %d, %d, %d, %d, %d
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=1, gpu_id=0, num_worlds=N, sim_flags=FLAGS, rand_seed=21, min_hiders=1,
      max_hiders=H, min_seekers=1, max_seekers=K, num_pbt_policies=1)
ref = hs_ref.RefSim(N, sim_flags=FLAGS, rand_seed=21, min_hiders=1, max_hiders=H, min_seekers=1, max_seekers=K, threads=8)
sim.init(); ref.init()
act = sim.action_tensor().to_torch()
rng = np.random.default_rng(4)
h = hashlib.sha256()
for s in range(STEPS):
    a = ref.tensor("action")
    lo, hi = (0, 5) if FLAGS & 8 else (0, 11)
    a[:, 0:3] = rng.integers(lo, hi, size=(a.shape[0], 3)); a[:, 3:5] = rng.integers(0, 2, size=(a.shape[0], 2)) * (rng.random((a.shape[0], 2)) < 0.2)
    act.copy_(torch.from_numpy(a.copy()).to(act.device))
    sim.step(); ref.step()
    if s %% 6 == 5 or s == STEPS - 1:
        gb, gm = sim.debug_bodies(); rb, rm = ref.bodies()
        assert np.array_equal(gm, rm), ("meta", s)
        bad = np.argwhere(gb.view(np.int32) != rb.view(np.int32))
        assert bad.size == 0, ("bodies", s, bad[:5].tolist())
        for n in ("self_data", "agent_data", "box_data", "ramp_data", "lidar", "reward", "done", "visible_boxes_mask", "global_positions"):
            g = getattr(sim, n + "_tensor")().to_torch().cpu().numpy().reshape(ref.tensor(n).shape)
            assert np.array_equal(g.view(np.int32), ref.tensor(n).view(np.int32)), (n, s)
        h.update(gb.tobytes())
st = sim.device_status()
print("PARITY", h.hexdigest(), st["spilled_dd_pairs"], st["spilled_static_pairs"], st["dropped_candidate_pairs"])
"""


def _parity(lib, worlds, hiders, seekers, flags, steps, env=None):
    code = PARITY_LOOP % (os.path.join(ROOT, "oracle"), worlds, hiders, seekers, flags, steps)
    out = _child(code, {**({"HS_LIB_PATH": lib} if lib else {}), **(env or {})}, timeout=900)
    return [l for l in out.splitlines() if l.startswith("PARITY")][0].split()


@pytest.mark.parametrize("hiders,seekers,flags", [(2, 2, 0), (3, 3, 13)])
def test_pairs_beyond_the_lds_capacities_spill_and_nothing_is_dropped(hiders, seekers, flags):
    """a candidate pair beyond the LDS capacities (16 body-body, 24 body-static per world and
    substep) must be solved, not dropped (src/sim.cpp:1356-1361 has no such cap).  A build whose capacities are ONE pair of
    each kind (libhideseek_smallcap.so) sends nearly every pair through the spill path — global lists, sequential solve —
    and must still agree with the unbounded oracle bit for bit over 120 steps, report what it spilled, drop nothing, and
    give the very same trajectory as the normal build (whose spill counters stay at zero on this small batch)."""
    import build as hs_build
    small = _parity(hs_build.build_smallcap(), 96, hiders, seekers, flags, 120)
    # (two agents of a team pushing things around meet more than one body-body pair per world soon enough; four agents
    # under the benchmark's forces mostly meet walls)
    assert int(small[3]) > 0 and (int(small[2]) > 0 or hiders < 3) and small[4] == "0", small
    normal = _parity(None, 96, hiders, seekers, flags, 120)
    assert normal[1] == small[1], "same trajectory whichever path a pair takes"
    assert normal[4] == "0"


def test_four_worlds_per_wave_gives_the_same_trajectory():
    """HS_TILE=4: the physics kernel with a wave per HALF octet (four waves per SIMD, 10 KiB of LDS, 128 registers) instead
    of a wave per octet — the same templated code with 16 lanes per world.  Oracle parity over 120 steps with joints and
    locks live, and the digest of the default tiling."""
    code = PARITY_LOOP % (os.path.join(ROOT, "oracle"), 100, 3, 3, 13, 120)
    a = [l for l in _child(code, {"HS_TILE": "4"}, timeout=900).splitlines() if l.startswith("PARITY")][0].split()
    b = [l for l in _child(code, {"HS_TILE": "8"}, timeout=900).splitlines() if l.startswith("PARITY")][0].split()
    assert a[1] == b[1] and a[4] == "0"
    code = PARITY_LOOP % (os.path.join(ROOT, "oracle"), 77, 2, 2, 0, 60)          # a world count with partial tiles, 4 agents: one round of bodies
    a = [l for l in _child(code, {"HS_TILE": "4"}, timeout=900).splitlines() if l.startswith("PARITY")][0].split()
    assert a[4] == "0"
    # ... and its spill path: capacities of one pair of each kind
    import build as hs_build
    small = _parity(hs_build.build_smallcap(), 64, 3, 3, 13, 90, env={"HS_TILE": "4"})
    assert int(small[3]) > 0 and small[4] == "0", small


def test_spill_counters_and_graph_flag_are_reported():
    import build as hs_build
    out = _child(STEP_LOOP, {"HS_LIB_PATH": hs_build.build_smallcap()})
    st = [l for l in out.splitlines() if l.startswith("STATUS")][0].split()
    assert int(st[1]) + int(st[2]) > 0 and st[4] == "0", out
    out = _child(STEP_LOOP)
    st = [l for l in out.splitlines() if l.startswith("STATUS")][0].split()
    assert st[1:3] == ["0", "0"] and st[4] == "0", out


def test_a_closed_simulator_refuses_new_views():
    """a Tensor knows its simulator only weakly (`del sim` frees the HBM at once, scripts/benchmark.py:94) and
    must not hand out views of freed memory."""
    import gpu_hideseek
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=8, sim_flags=0, rand_seed=1,
        min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    sim.init()
    t = sim.action_tensor()
    assert t.to_torch().shape == (32, 5)
    sim.close()
    with pytest.raises(RuntimeError, match="closed or deleted"):
        t.to_torch()
    sim2 = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=8, sim_flags=0, rand_seed=1,
        min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    t2 = sim2.reward_tensor()
    del sim2
    with pytest.raises(RuntimeError, match="closed or deleted"):
        t2.to_torch()


def test_graph_mode_is_visible():
    out = _child(STEP_LOOP, {"HS_GRAPH": "1"})
    assert [l for l in out.splitlines() if l.startswith("STATUS")][0].split()[3] == "1", out
    out = _child(STEP_LOOP, {"HS_GRAPH": "0"})
    assert [l for l in out.splitlines() if l.startswith("STATUS")][0].split()[3] == "0", out


def test_sharded_front_end_equals_the_monolithic_run():
    """ShardedSimulator with two handles on the one GPU of this box (each with its own stream, started together)
    reproduces a single 600-world handle bit for bit: bodies, gathered observations, scattered actions."""
    import torch
    import gpu_hideseek
    N, A = 600, 4
    kw = dict(sim_flags=0, rand_seed=11, min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
    mono = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, **kw)
    shd = gpu_hideseek.ShardedSimulator([0, 0, 0], N, **kw)
    assert [s.num_worlds for s in shd.shards] == [200, 200, 200] and [s.world_offset for s in shd.shards] == [0, 200, 400]
    mono.init(); shd.init()
    act_m = mono.action_tensor().to_torch()
    act_s = shd.action_tensor()
    assert act_s.shape == (N * A, 5)
    rng = np.random.default_rng(5)
    for step in range(20):
        a = torch.from_numpy(rng.integers(-5, 5, size=(N * A, 5)).astype(np.int32))
        a[:, 2:] = 0
        act_m.copy_(a)
        act_s.scatter(a)
        mono.step(); shd.step()
    for name in ("self_data", "lidar", "box_data", "reward", "global_positions", "seed"):
        m = getattr(mono, name + "_tensor")().to_torch().cpu()
        g = getattr(shd, name + "_tensor")().gather()
        assert g.is_pinned() and torch.equal(g.view(torch.int32), m.view(torch.int32)), name
    mb = mono.debug_bodies()[0]
    sb = np.concatenate([s.debug_bodies()[0] for s in shd.shards])
    assert np.array_equal(mb.view(np.int32), sb.view(np.int32))
    shd.trigger_reset(399, 1)
    assert shd.shards[1].reset_tensor().to_torch()[199, 0].item() == 1
    shd.set_action(401 * A + 2, 1, 2, 3, 0, 0)
    assert shd.shards[2].action_tensor().to_torch()[1 * A + 2].tolist() == [1, 2, 3, 0, 0]
    assert shd.device_status()["dropped_candidate_pairs"] == 0
    shd.close()
