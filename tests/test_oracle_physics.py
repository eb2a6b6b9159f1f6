"""Physics known answers from the debug levels and unit checks of the narrowphase
and the ray caster, on the CPU oracle."""
import ctypes as C

import numpy as np


def _level(oracle, lvl, steps):
    s = oracle.RefSim(1, sim_flags=2, min_hiders=1, max_hiders=1, min_seekers=1, max_seekers=1)
    s.tensor("reset")[:] = lvl
    s.init()
    traj = []
    for _ in range(steps):
        s.step()
        b, m = s.bodies()
        traj.append(b[0].copy())
    return np.array(traj), m[0]


def test_free_fall_matches_semi_implicit_euler(oracle):
    """Level 3 (level_gen.cpp:358-361): cube dropped from z=5; h = 1/120, g = -9.8."""
    traj, _ = _level(oracle, 3, 20)
    h, g = np.float32(1 / 30) / np.float32(4), np.float32(-9.8)
    z, v = np.float32(5), np.float32(0)
    for step in range(20):
        for _ in range(4):
            v = np.float32(v + g * h)
            z = np.float32(z + v * h)
        assert abs(traj[step, 0, 2] - z) < 1e-3   # v is re-derived from positions each substep
        assert abs(traj[step, 0, 9] - v) < 1e-2


def test_cube_settles_on_plane(oracle):
    traj, _ = _level(oracle, 3, 150)
    assert abs(traj[-1, 0, 2] - 1.0) < 2e-3
    assert np.abs(traj[-1, 0, 7:]).max() < 1e-2
    assert traj[:, 0, 2].min() > 0.99


def test_stack_and_corner_drop_stay_above_plane(oracle):
    for lvl in (2, 7):
        traj, m = _level(oracle, lvl, 200)
        present = m[:, 0] >= 0
        assert traj[:, present, 2].min() > 0.95
        assert np.abs(traj[-1, present, 7:10]).max() < 0.2
        assert np.isfinite(traj).all()


def test_fast_ramp_does_not_tunnel(oracle):
    """Level 8 (level_gen.cpp:464-499): ramp hits the ground at -30 m/s."""
    traj, m = _level(oracle, 8, 120)
    assert traj[:, 9, 2].min() > 0.2
    assert np.allclose(traj[-1, 10, :3], [-0.5, -0.5, 1.0])      # the Static ramp never moves
    assert np.isfinite(traj).all()


def test_agent_cannot_pass_wall(oracle):
    """Level 6 (level_gen.cpp:407-432): wall x in [-10,10], y in [-0.2,0.2]; the hider starts at
    (-15,-15) with yaw -45 deg, so a body-frame +y force drives it along the world diagonal into it."""
    s = oracle.RefSim(1, sim_flags=2 | 8, min_hiders=1, max_hiders=1, min_seekers=1, max_seekers=1)
    s.tensor("reset")[:] = 6
    s.init()
    pts = []
    for _ in range(200):
        s.tensor("action")[0] = [2, 4, 2, 0, 0]
        s.step()
        pts.append(s.bodies()[0][0, 11, :2].copy())
    pts = np.array(pts)
    assert pts[:, 1].max() > -3                      # it reached the wall
    beside = np.abs(pts[:, 0]) < 8.5                 # while alongside the wall ...
    assert beside.sum() > 20
    assert pts[beside, 1].max() < -1.15              # ... the cube (half extent 1) never enters it


def test_ray_body_known_answers(oracle):
    L = oracle.lib()

    def ray(obj, pos, rot, o, d):
        f = lambda v: (C.c_float * len(v))(*v)
        return L.hsref_ray_body(obj, f(pos), f(rot), f(o), f(d))
    ident = [1, 0, 0, 0]
    assert ray(2, [0, 0, 0], ident, [-5, 0, 0], [1, 0, 0]) == 4.0           # cube face at x=-1
    assert ray(2, [0, 0, 0], ident, [0, 0, 0], [1, 0, 0]) == -1.0           # origin inside: no hit
    assert ray(2, [0, 0, 0], ident, [-5, 0, 0], [-1, 0, 0]) == -1.0         # pointing away
    assert ray(7, [0, 0, 0], ident, [0, -5, 0], [0, 1, 0]) == 4.25          # elongated box half width .75
    assert abs(ray(2, [0, 0, 0], [np.sqrt(.5), 0, 0, np.sqrt(.5)], [-5, 0.5, 0], [1, 0, 0]) - 4.0) < 1e-5
    # wedge: vertical ray down onto the slanted face at y=-0.5: z on the slope = -1 + (y+2)*2/3 = 0
    assert abs(ray(6, [0, 0, 0], ident, [0, -0.5, 5], [0, 0, -1]) - 5.0) < 1e-5
    assert ray(6, [0, 0, 0], ident, [0, 0.99, 5], [0, 0, -1]) < 4.02        # near the tall end (z ~ 1)
    # t is in units of |d| (segment tests pass un-normalised d with t_max 1, sim.cpp:602)
    assert abs(ray(2, [0, 0, 0], ident, [-5, 0, 0], [8, 0, 0]) - 0.5) < 1e-6


def test_collide_face_and_separation(oracle):
    L = oracle.lib()
    n = (C.c_float * 3)(); pA = (C.c_float * 12)(); pB = (C.c_float * 12)()
    f = lambda v: (C.c_float * len(v))(*v)
    ident = f([1, 0, 0, 0])
    # two unit-half-extent cubes overlapping by 0.1 along x: face contact, 4 points, normal +x (A -> B)
    c = L.hsref_collide(2, f([0, 0, 0]), ident, 2, f([1.9, 0, 0]), ident, n, pA, pB)
    assert c == 4
    assert np.allclose(n[:], [1, 0, 0])
    a, b = np.array(pA[:]).reshape(4, 3), np.array(pB[:]).reshape(4, 3)
    assert np.allclose(((a - b) @ np.array(n[:])), 0.1, atol=1e-6)          # penetration depth
    assert L.hsref_collide(2, f([0, 0, 0]), ident, 2, f([2.1, 0, 0]), ident, n, pA, pB) == 0
    # cube resting on the ramp's slope produces contacts with a normal along the slope normal
    c = L.hsref_collide(6, f([0, 0, 0]), ident, 2, f([0, -0.5, 1.1]), f([0.9659258, 0.2588190, 0, 0]), n, pA, pB)
    assert c >= 1
    assert np.isfinite(np.array(n[:])).all() and abs(np.linalg.norm(n[:]) - 1) < 1e-5


def test_jax_config_with_grabs_stays_finite_and_normalised(oracle):
    """scripts/jax_train.py configuration (3+3 agents, RandomFlipTeams|UseFixedWorld|ZeroAgentVelocity) with grab / lock
    actions: joints snap misaligned bodies round by large angles, which the Newton-step quaternion normalisation must
    not be used for (it once produced NaN rotations here)."""
    ref = oracle.RefSim(40, sim_flags=13, rand_seed=5, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3, threads=4)
    ref.init()
    rng = np.random.default_rng(1)
    rows = ref.N * ref.A
    worst = 0.0
    for t in range(130):
        ref.tensor("action")[:] = np.stack([rng.integers(0, 11, rows), rng.integers(0, 11, rows), rng.integers(0, 11, rows),
                                            rng.integers(0, 2, rows), rng.integers(0, 2, rows)], axis=1)
        ref.step()
        b, m = ref.bodies()
        live = m[:, :, 0] >= 0
        assert np.isfinite(b[live]).all(), f"non-finite body state at step {t}"
        worst = max(worst, float(np.abs(np.linalg.norm(b[live][:, 3:7], axis=1) - 1.0).max()))
    assert worst < 1e-3, worst
    for k in ("self_data", "agent_data", "box_data", "lidar", "reward"):
        assert np.isfinite(ref.tensor(k)).all(), k
    ref.close()
