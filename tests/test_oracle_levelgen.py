"""Level generation (src/level_gen.cpp, src/geo_gen.cpp): invariants derivable from first-party
source and the committed golden layouts."""
import os

import numpy as np

from conftest import GOLDEN


def _sim(oracle, n=64, **kw):
    s = oracle.RefSim(n, **kw)
    s.init()
    return s


def test_training_level_invariants(oracle):
    s = _sim(oracle, 256, rand_seed=1, min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3)
    walls, info = s.walls()
    b, m = s.bodies()
    nwalls, nplanes, nboxes, nramps, nh, ns = (info[:, i] for i in range(6))
    assert (nwalls >= 4).all() and (nwalls <= 34).all()          # geo_gen.cpp:429-462
    assert (nplanes == 1).all()
    assert (nboxes >= 3).all() and (nboxes <= 9).all()           # level_gen.cpp:84
    assert (nramps == 2).all()
    assert (nh >= 1).all() and (nh <= 3).all() and (ns >= 1).all() and (ns <= 3).all()
    for w in range(256):
        k = nwalls[w]
        hx, hy = walls[w, :k, 2], walls[w, :k, 3]
        thin_x, thin_y = np.isclose(hx, 0.2), np.isclose(hy, 0.2)
        assert (thin_x | thin_y).all()                           # axis-aligned, thickness 0.2 (geo_gen.cpp:489-497)
        assert (np.abs(walls[w, :k, 0]) <= 18.0001).all() and (np.abs(walls[w, :k, 1]) <= 18.0001).all()
        types = m[w, :, 0]
        present = types >= 0
        assert (np.abs(b[w, present, 0]) <= 18).all() and (np.abs(b[w, present, 1]) <= 18).all()
        assert (b[w, present, 2] == 1.0).all()                   # spawn z == 1
        box_types = types[:9][:nboxes[w]]
        assert set(box_types.tolist()) <= {2, 7}                 # Cube / Box
        n_el = (box_types == 7).sum()
        assert n_el >= 3 or nboxes[w] == 3                       # >= 3 elongated (level_gen.cpp:87-88)
        assert (box_types[:n_el] == 7).all()                     # elongated first, then cubes
        assert (types[9:11] == 6).all()                          # two ramps
        na = nh[w] + ns[w]
        assert (types[11:11 + na] >= 4).all() and (types[11 + na:] == -1).all()
        # box/ramp rotations are pure yaw
        assert np.allclose(b[w, present, 4:6], 0)
    mask = s.tensor("self_mask").reshape(256, s.A)
    assert ((mask.sum(1)) == nh + ns).all()
    stype = s.tensor("self_type").reshape(256, s.A)
    for w in range(256):                                         # hiders first without RandomFlipTeams
        assert (stype[w, :nh[w]] == 1).all() and (stype[w, nh[w]:nh[w] + ns[w]] == 0).all()
    act = s.tensor("action").reshape(256, s.A, 5)
    for w in range(256):
        assert (act[w, :nh[w] + ns[w]] == [2, 2, 2, 0, 0]).all()  # makeAgent level_gen.cpp:26-32


def test_seed_tensor_and_world_keys(oracle):
    s = _sim(oracle, 16, rand_seed=9, world_offset=100)
    seed = s.tensor("seed").reshape(16, s.A, 2)
    assert (seed[:, :, 0] == 0).all()                            # episode index 0
    assert (seed[:, 0, 1] == np.arange(100, 116)).all()          # global world id (sim.cpp:107-110)
    walls, info = s.walls()
    assert len({walls[w].tobytes() for w in range(16)}) > 1      # different worlds, different layouts
    s.tensor("reset")[:] = 1
    s.step()
    assert (s.tensor("seed").reshape(16, s.A, 2)[:, 0, 0] == 1).all()


def test_use_fixed_world_gives_identical_layouts(oracle):
    s = _sim(oracle, 8, sim_flags=1, rand_seed=4)
    walls, info = s.walls()
    b, m = s.bodies()
    for w in range(1, 8):
        assert np.array_equal(walls[w], walls[0]) and np.array_equal(info[w, :4], info[0, :4])
        assert np.array_equal(b[w, :11, :7], b[0, :11, :7])
    s.tensor("reset")[:] = 1
    s.step()
    walls2, _ = s.walls()
    assert np.array_equal(walls2, walls)                         # and in every episode


def test_world_offset_shards_replicate_global_worlds(oracle):
    full = _sim(oracle, 12, rand_seed=2)
    hi = _sim(oracle, 4, rand_seed=2, world_offset=8)
    wf, inf_f = full.walls()
    wh, inf_h = hi.walls()
    assert np.array_equal(wf[8:], wh) and np.array_equal(inf_f[8:], inf_h)
    assert np.array_equal(full.bodies()[0][8:], hi.bodies()[0])


def test_random_flip_teams_flag(oracle):
    s = _sim(oracle, 64, sim_flags=4, rand_seed=3)
    _, info = s.walls()
    assert 0 < info[:, 7].sum() < 64                             # some worlds start with seekers first
    stype = s.tensor("self_type").reshape(64, s.A)
    for w in range(64):
        assert stype[w, 0] == (0 if info[w, 7] else 1)


def _compare_layout(oracle, fname, seed, flags):
    g = np.load(os.path.join(GOLDEN, fname))
    s = oracle.RefSim(8, sim_flags=flags, rand_seed=seed, min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3)
    s.init()
    for ep in range(2):
        w, info = s.walls()
        b, m = s.bodies()
        assert np.array_equal(info, g[f"info_ep{ep}"])
        assert np.array_equal(w.view(np.int32), g[f"walls_ep{ep}"].view(np.int32))
        assert np.array_equal(b[:, :, :7].copy().view(np.int32), g[f"bodies_ep{ep}"].view(np.int32))
        assert np.array_equal(m, g[f"meta_ep{ep}"])
        assert np.array_equal(s.tensor("seed"), g[f"seed_ep{ep}"])
        assert np.array_equal(s.tensor("self_type"), g[f"self_type_ep{ep}"])
        s.tensor("reset")[:] = 1
        s.step()


def test_golden_layouts_seed0(oracle):
    _compare_layout(oracle, "levelgen_seed0.npz", 0, 0)


def test_golden_layouts_seed5_fixed_flip(oracle):
    _compare_layout(oracle, "levelgen_seed5_flip_fixed.npz", 5, 1 | 4)
