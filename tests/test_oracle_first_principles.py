"""First-principles anchors for the CPU restatement's engine parts — checks that do NOT come from the author's own
solver: conservation laws, Coulomb friction with the reference's material table (src/mgr.cpp:476-559), rigidity of
the grab joint (src/sim.cpp:343-356), resting contact of the debug stack (src/level_gen.cpp:434-462), and brute-force
geometry (numpy / scipy half-space arithmetic) against `trace_ray`'s hull tests and the convex narrowphase.

Scenes are built through the Checkpoint record (src/sim.hpp:283-313): save, edit the body states, load.
The solver's arithmetic is frozen against these (DESIGN.md "Engine decisions"): a later change of
oracle/hs_ref_phys.hpp has to be justified by one of these tests, not by kernel time.
"""
import math
import os

import numpy as np
import pytest

from conftest import GOLDEN

BODY = [("pos", "<f4", 3), ("rot", "<f4", 4), ("lin", "<f4", 3), ("ang", "<f4", 3)]
OBJ = np.dtype(BODY + [("team", "<u4"), ("locked", "u1"), ("pad", "u1", 3)])
AGENT = np.dtype(BODY + [("grab_idx", "<i4"), ("r1", "<f4", 3), ("r2", "<f4", 3), ("att1", "<f4", 4), ("att2", "<f4", 4),
                         ("sep", "<f4")])
CKPT = np.dtype([("key", "<u4", 2), ("scores", "<i4", 2), ("step", "<i4"), ("agents", AGENT, 6), ("boxes", OBJ, 9),
                 ("ramps", OBJ, 2), ("nh", "<i4"), ("ns", "<i4"), ("nb", "<i4"), ("nr", "<i4")])
assert CKPT.itemsize == 1392

CUBE, RAMP, BOX = 2, 6, 7            # SimObject (src/sim.hpp:78-88)
FIXED_NO_EPISODE_END = 1 | 2         # UseFixedWorld | IgnoreEpisodeLength


def make(oracle, **kw):
    args = dict(sim_flags=FIXED_NO_EPISODE_END, rand_seed=1, min_hiders=1, max_hiders=1, min_seekers=1, max_seekers=1)
    args.update(kw)
    ref = oracle.RefSim(1, **args)
    ref.init()
    return ref


def edit_scene(ref, edit):
    ref.tensor("ckpt_ctrl")[:] = 1
    ref.save_checkpoints()
    rec = ref.tensor("ckpt").view(CKPT).reshape(-1)
    edit(rec[0])
    ref.tensor("ckpt_ctrl")[:] = 1
    ref.load_checkpoints()
    ref.tensor("ckpt_ctrl")[:] = 0


def put(rec, pos, rot=(1, 0, 0, 0), lin=(0, 0, 0), ang=(0, 0, 0)):
    rec["pos"] = pos; rec["rot"] = rot; rec["lin"] = lin; rec["ang"] = ang


def slots_of(ref, obj):
    _, m = ref.bodies()
    return [i for i in range(17) if m[0, i, 0] == obj]


def free_spot(ref, margin=3.0):
    """A place on the floor at least `margin` from every wall and 4.2 + margin from every body centre."""
    w, info = ref.walls()
    b, m = ref.bodies()
    for x in np.arange(-15, 15.1, 0.5):
        for y in np.arange(-15, 15.1, 0.5):
            ok = all(math.hypot(max(abs(x - w[0, k, 0]) - w[0, k, 2], 0), max(abs(y - w[0, k, 1]) - w[0, k, 3], 0)) >= margin
                     for k in range(info[0, 0]))
            ok = ok and all(m[0, i, 0] < 0 or math.hypot(b[0, i, 0] - x, b[0, i, 1] - y) >= margin + 4.2 for i in range(17))
            if ok:
                return float(x), float(y)
    raise AssertionError("no free spot in the fixed level")


def qmul(a, b):
    return np.array([a[0] * b[0] - a[1] * b[1] - a[2] * b[2] - a[3] * b[3], a[0] * b[1] + a[1] * b[0] + a[2] * b[3] - a[3] * b[2],
                     a[0] * b[2] - a[1] * b[3] + a[2] * b[0] + a[3] * b[1], a[0] * b[3] + a[1] * b[2] - a[2] * b[1] + a[3] * b[0]])


def qinv(q):
    return np.array([q[0], -q[1], -q[2], -q[3]])


def qrot(q, v):
    return qmul(qmul(q, np.array([0.0, *v])), qinv(q))[1:]


def rotm(q):
    return np.stack([qrot(q, e) for e in np.eye(3)], axis=1)


# ------------------------------------------------------------------------------------------------ conservation laws
def test_two_cube_collision_conserves_momentum(oracle):
    """Two free cubes (m = 2 each: invMass 0.5, src/mgr.cpp:476-559) collide off-centre in mid-air.  No horizontal
    external force acts, and gravity exerts no torque about the pair's centre of mass: the horizontal linear momentum is
    conserved to rounding, the vertical one follows -g M t, the angular momentum about the centre of mass to 2 % (an
    XPBD contact applies equal and opposite corrections at two points that differ by the penetration)."""
    ref = make(oracle)
    c0, c1 = slots_of(ref, CUBE)[:2]

    def scene(r):
        put(r["boxes"][c0], [-1.5, 0.45, 14], lin=[3, 0, 0])
        put(r["boxes"][c1], [1.5, -0.45, 14], lin=[-3, 0, 0])
    edit_scene(ref, scene)

    def momenta():
        b, _ = ref.bodies()
        com = (b[0, c0, :3] + b[0, c1, :3]) / 2
        P = sum(2.0 * b[0, i, 7:10] for i in (c0, c1))
        L = sum(np.cross(b[0, i, :3] - com, 2.0 * b[0, i, 7:10]) + (1 / 0.75) * b[0, i, 10:13] for i in (c0, c1))
        return P.astype(np.float64), L.astype(np.float64)
    P0, L0 = momenta()
    assert np.allclose(P0, 0) and np.allclose(L0, [0, 0, -5.4], atol=1e-5)
    hit = False
    for s in range(24):
        ref.step()
        P, L = momenta()
        b, _ = ref.bodies()
        hit = hit or abs(b[0, c0, 7] - 3.0) > 0.5
        assert abs(P[0]) < 1e-4 and abs(P[1]) < 1e-4, (s, P)
        assert abs(P[2] + 9.8 * 4.0 * (s + 1) / 30.0) < 2e-2, (s, P)
        assert np.linalg.norm(L - L0) < 0.02 * np.linalg.norm(L0) + 1e-3, (s, L)
    assert hit, "the cubes did collide"
    b, _ = ref.bodies()
    assert abs(b[0, c0, 7] - b[0, c1, 7]) < 1.0, "restitution 0: the approach speed is gone"


def test_free_flight_conserves_spin_and_follows_the_parabola(oracle):
    ref = make(oracle)
    c0 = slots_of(ref, CUBE)[0]
    edit_scene(ref, lambda r: put(r["boxes"][c0], [0, 0, 15], lin=[1.0, -2.0, 3.0], ang=[0.7, -0.4, 1.1]))
    for _ in range(20):
        ref.step()
    b, _ = ref.bodies()
    t = 20 / 30.0
    assert np.allclose(b[0, c0, 7:9], [1.0, -2.0], atol=1e-5) and abs(b[0, c0, 9] - (3.0 - 9.8 * t)) < 1e-2   # (velocities are pose differences in f32 at z = 15)
    # semi-implicit Euler at h = 1/120: z = z0 + v0 t - g h^2 n(n+1)/2 with n = 80 substeps
    assert abs(b[0, c0, 2] - (15 + 3.0 * t - 9.8 * (1 / 120) ** 2 * 80 * 81 / 2)) < 5e-3
    assert np.allclose(b[0, c0, 10:13], [0.7, -0.4, 1.1], atol=2e-3), "a cube's inertia is isotropic: omega is constant"
    assert abs(np.linalg.norm(b[0, c0, 3:7]) - 1) < 1e-5


# ------------------------------------------------------------------------------------------------ Coulomb friction
def test_cube_on_the_locked_ramp_slips_because_tan_theta_exceeds_mu_s(oracle):
    """The ramp wedge rises 2 over 3 (data/ramp_collision.obj): theta = 33.69 deg, tan = 0.667.  Cube and ramp both have
    mu_s = 0.5 (src/mgr.cpp:476-559) < tan theta, so the cube cannot stay put; but the table's KINETIC coefficient is
    (2 + 1) / 2 = 1.5 > tan theta, so once it slips it is braked harder than gravity pulls: stick-slip, i.e. a slow steady
    creep down the slope — far from the frictionless 0.5 g sin(theta) t^2 = 10.9 m in 2 s, and not sideways.  On the
    level floor the same cube does not move at all."""
    ref = make(oracle)
    c0 = slots_of(ref, CUBE)[0]
    x, y = free_spot(ref, 4.0)
    th = math.atan2(2, 3)
    n = np.array([0, -2, 3]) / math.sqrt(13)
    down = np.array([0, -3, -2]) / math.sqrt(13)

    def scene(r):
        put(r["ramps"][0], [x, y, 1])
        r["ramps"][0]["locked"] = 1; r["ramps"][0]["team"] = 1
        put(r["boxes"][c0], np.array([x, y - 0.5, 1.0]) + n * 1.001, rot=[math.cos(th / 2), math.sin(th / 2), 0, 0])
    edit_scene(ref, scene)
    b, _ = ref.bodies()
    p0 = b[0, c0, :3].copy()
    prev = 0.0
    for s in range(60):
        ref.step()
        b, _ = ref.bodies()
        along = float((b[0, c0, :3] - p0) @ down)
        assert along > prev - 1e-4, "never uphill"
        prev = along
    d = b[0, c0, :3] - p0
    assert 0.05 < d @ down < 1.0, "slips, slowly"
    assert abs(d[0]) < 0.1, "not sideways"
    assert np.abs(b[0, c0, 7:10]).max() < 0.2
    assert np.array_equal(b[0, 9, :3], np.float32([x, y, 1])), "the locked ramp did not move"
    # the same cube on the level floor: no creep
    ref = make(oracle)
    edit_scene(ref, lambda r: put(r["boxes"][c0], [x, y, 1]))
    for _ in range(60):
        ref.step()
    b, _ = ref.bodies()
    assert np.abs(b[0, c0, :3] - np.float32([x, y, 1])).max() < 1e-3 and np.abs(b[0, c0, 7:13]).max() < 1e-2


def test_floor_friction_holds_a_resting_agent_and_yields_to_a_large_force(oracle):
    """Agent on the floor: mu_s = (0.5 + 2) / 2 = 1.25, m = 1 (src/mgr.cpp:476-559) -> Coulomb threshold 12.25 N.
    No force: stays exactly where it is.  36 N forward (action 8 -> 12 * 3 N, src/sim.cpp:221-223) exceeds the static
    threshold, so it moves forward — far slower than the frictionless 36 m/s^2: the table's KINETIC coefficient for an
    agent on the floor is (16 + 2) / 2 = 9, i.e. 88 N of sliding friction against a 36 N push, so once it slips it is braked
    again within the substep (stick-slip creep, as for the cube on the ramp below): the displacement grows step by step
    and the velocity left at the end of a step is small but forward.  And it creeps STRAIGHT: no sideways drift beyond a
    fraction of the forward motion (with the whole floor reaction on one corner it used to veer off and spin)."""
    for a, moves in ((5, False), (8, True)):
        ref = make(oracle)
        x, y = free_spot(ref)
        edit_scene(ref, lambda r: put(r["agents"][0], [x, y, 1]))
        for _ in range(3):
            ref.tensor("action")[0] = [5, a, 5, 0, 0]
            ref.step()
        b, _ = ref.bodies()
        d = b[0, 11, :3] - np.float32([x, y, 1])
        if not moves:
            assert np.array_equal(d, np.zeros(3, np.float32)) and not b[0, 11, 7:13].any()
        else:
            assert 0.002 < d[1] < 0.5 * 36 * 0.1 ** 2 and b[0, 11, 8] > 0.02, d
            assert abs(d[0]) < 0.25 * d[1] and abs(b[0, 11, 12]) < 0.1, (d, b[0, 11, 12])


def test_a_straight_push_does_not_spin_the_agent(oracle):
    ref = make(oracle)
    x, y = free_spot(ref)
    edit_scene(ref, lambda r: put(r["agents"][0], [x, y, 1]))
    for _ in range(10):
        ref.tensor("action")[0] = [5, 7, 5, 0, 0]
        ref.step()
    b, _ = ref.bodies()
    assert abs(b[0, 11, 12]) < 0.05


# ------------------------------------------------------------------------------------------------ joints and stacks
def test_grabbed_box_keeps_its_pose_relative_to_the_agent(oracle):
    """makeFixedJoint (src/sim.cpp:343-356): after a grab the box is rigidly attached; its pose in the agent's frame
    stays within 1 cm / 1e-4 (quaternion) while the agent is driven around for 100 steps."""
    ref = make(oracle)
    c0 = slots_of(ref, CUBE)[0]
    x, y = free_spot(ref, 4.0)

    def scene(r):
        put(r["agents"][0], [x, y, 1])
        put(r["boxes"][c0], [x, y + 2.6, 1])
    edit_scene(ref, scene)
    ref.tensor("action")[0] = [5, 5, 5, 1, 0]
    ref.step()
    assert ref.tensor("self_data")[0, 12] == 1.0, "the grab ray (2.5 m along +y) found the cube"

    def rel():
        b, _ = ref.bodies()
        qa = b[0, 11, 3:7].astype(np.float64)
        return qrot(qinv(qa), b[0, c0, :3] - b[0, 11, :3]), qmul(qinv(qa), b[0, c0, 3:7].astype(np.float64))
    r0, q0 = rel()
    rng = np.random.default_rng(0)
    for _ in range(100):
        ref.tensor("action")[0] = [rng.integers(3, 8), rng.integers(3, 8), rng.integers(3, 8), 0, 0]
        ref.step()
        r, q = rel()
        assert np.abs(r - r0).max() < 0.01 and 1 - abs(q @ q0) < 1e-4
    assert ref.tensor("self_data")[0, 12] == 1.0


def test_debug_drop_pile_settles_with_bounded_penetration(oracle):
    """Level 7 (src/level_gen.cpp:434-462): two tilted cubes dropped from 5 m and 10 m, the second onto the first.
    During the pile-up nothing may sink into the floor by more than 5 cm; after 200 steps both rest flat on the floor
    (centre height 1 within 2 mm), and they do not overlap."""
    ref = oracle.RefSim(1, sim_flags=2, rand_seed=0, min_hiders=1, max_hiders=1, min_seekers=1, max_seekers=1)
    ref.tensor("reset")[:] = 7
    ref.init()
    _, m = ref.bodies()
    live = [i for i in range(11) if m[0, i, 0] >= 0]
    assert [m[0, i, 0] for i in live] == [CUBE, CUBE]
    for s in range(200):
        ref.step()
        b, _ = ref.bodies()
        for i in live:
            R = rotm(b[0, i, 3:7].astype(np.float64))
            assert b[0, i, 2] - np.abs(R[2]) @ np.ones(3) > -0.05, (s, i)
    assert np.abs(b[0, live, 7:13]).max() < 1e-3, "at rest"
    assert np.abs(b[0, live, 2] - 1.0).max() < 2e-3, "flat on the floor"
    NA, DA, _ = half_spaces(CUBE, b[0, live[0], :3], b[0, live[0], 3:7])
    NB, DB, _ = half_spaces(CUBE, b[0, live[1], :3], b[0, live[1], 3:7])
    assert not polytopes_intersect(NA, DA, NB, DB, shrink=0.01)


def test_quaternions_stay_normalised_with_the_zero_velocity_torques(oracle):
    """ZeroAgentVelocity mode applies 240 N m (src/sim.cpp:248-250); one Newton step of normalisation per update
     must keep |q|^2 within 1e-3 of 1 over two episodes, for every body."""
    ref = oracle.RefSim(24, sim_flags=13, rand_seed=5, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3, threads=4)
    ref.init()
    rng = np.random.default_rng(3)
    worst = 0.0
    for s in range(300):
        a = ref.tensor("action")
        a[:, :3] = rng.integers(0, 5, size=(a.shape[0], 3)); a[:, 3:] = rng.integers(0, 2, size=(a.shape[0], 2))
        ref.step()
        b, m = ref.bodies()
        q2 = (b[:, :, 3:7].astype(np.float64) ** 2).sum(axis=2)
        worst = max(worst, np.abs(q2[m[:, :, 0] >= 0] - 1).max())
        assert np.isfinite(b).all()
    assert worst < 1e-3, worst


# ------------------------------------------------------------------------------------------------ brute-force geometry
def hull_vertices(obj):
    if obj == RAMP:
        return np.array([[1, 1, 1], [1, 1, -1], [1, -2, -1], [-1, 1, 1], [-1, 1, -1], [-1, -2, -1]], np.float64)
    e = np.array([4, 0.75, 1.0]) if obj == BOX else np.ones(3)
    return np.array([[sx * e[0], sy * e[1], sz * e[2]] for sx in (-1, 1) for sy in (-1, 1) for sz in (-1, 1)], np.float64)


def half_spaces(obj, pos, rot):
    """World-space half-spaces n.x <= d of a placed hull, from its vertex set (scipy's Qhull — not the oracle's tables)."""
    from scipy.spatial import ConvexHull
    v = hull_vertices(obj) @ rotm(np.asarray(rot, np.float64)).T + np.asarray(pos, np.float64)
    eq = ConvexHull(v).equations
    return eq[:, :3], -eq[:, 3], v


def random_pose(rng, spread):
    q = rng.normal(size=4); q /= np.linalg.norm(q)
    return rng.uniform(-spread, spread, 3), q


def test_ray_hull_entry_against_half_space_clipping(oracle):
    """`ray_box_local` / `ray_wedge_local` (the hull tests inside trace_ray) against a double-precision
    half-space clip of the same ray on random poses; the cases are stored under tests/golden/ray_hull_cases.npz."""
    L = oracle.lib()
    rng = np.random.default_rng(11)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data
    cases = []
    for _ in range(3000):
        obj = int(rng.choice([CUBE, RAMP, BOX]))
        pos, q = random_pose(rng, 3.0)
        o = rng.uniform(-9, 9, 3)
        d = (pos + rng.uniform(-2.5, 2.5, 3)) - o if rng.random() < 0.8 else rng.normal(size=3)
        pos, q, o, d = (np.float32(x) for x in (pos, q, o, d))
        t = L.hsref_ray_body(obj, f(pos), f(q), f(o), f(d))
        N, D, _ = half_spaces(obj, pos, q)
        den, num = N @ d.astype(np.float64), D - N @ o.astype(np.float64)
        with np.errstate(divide="ignore", invalid="ignore"):
            tt = num / den
        tn = np.max(np.where(den < -1e-12, tt, -np.inf))
        tf = np.min(np.where(den > 1e-12, tt, np.inf))
        inside_slabs = np.all(num[np.abs(den) <= 1e-12] >= 0)
        expect = tn if (inside_slabs and tn <= tf and tn >= 0) else -1.0
        margin = min(abs(tf - tn), abs(tn)) if np.isfinite(tn) and np.isfinite(tf) else 1.0
        if margin < 1e-3:          # grazing / starting on the surface: the two roundings may disagree on hit vs miss
            continue
        cases.append((obj, *pos, *q, *o, *d, t, expect))
        assert (t < 0) == (expect < 0), (obj, pos, q, o, d, t, expect)
        if expect >= 0:
            assert abs(t - expect) < 1e-4 * max(1.0, expect), (t, expect)
    assert len(cases) > 2500
    path = os.path.join(GOLDEN, "ray_hull_cases.npz")
    if os.environ.get("HS_WRITE_GOLDEN") == "1":
        np.savez_compressed(path, cases=np.array(cases, np.float64))
    g = np.load(path)["cases"]
    assert g.shape == (len(cases), 16) and np.allclose(g, np.array(cases, np.float64), atol=0), "golden ray cases reproduce"


def polytopes_intersect(NA, DA, NB, DB, shrink=0.0):
    """LP feasibility: a point with N.x <= D - shrink for both hulls."""
    from scipy.optimize import linprog
    A = np.vstack([NA, NB]); bb = np.concatenate([DA, DB]) - shrink
    r = linprog(np.zeros(3), A_ub=A, b_ub=bb, bounds=[(None, None)] * 3, method="highs")
    return r.status == 0


def test_convex_narrowphase_against_linear_programming(oracle):
    """`collide_hulls` on random pairs against an LP over the hulls' half-spaces (scipy): 'separated' must mean the
    polytopes do not share an interior point; a manifold's normal must be a unit separating direction whose depth
    (max over contacts) un-penetrates the pair when B is moved along it, and every contact must lie on both hulls'
    surfaces or inside.  The cases are stored under tests/golden/narrowphase_cases.npz."""
    L = oracle.lib()
    rng = np.random.default_rng(5)
    f = lambda a: np.ascontiguousarray(a, np.float32).ctypes.data
    rows, hits = [], 0
    for _ in range(1200):
        oa, ob = (int(x) for x in rng.choice([CUBE, RAMP, BOX], 2))
        pa, qa = random_pose(rng, 0.5)
        pb, qb = random_pose(rng, 0.5)
        pb = pa + rng.normal(size=3) * rng.uniform(0.8, 3.5)
        if rng.random() < 0.5:          # the common case in the simulator: flat on the floor, yaw only
            ya, yb = rng.uniform(0, math.pi, 2)
            qa, qb = np.array([math.cos(ya / 2), 0, 0, math.sin(ya / 2)]), np.array([math.cos(yb / 2), 0, 0, math.sin(yb / 2)])
            pa[2] = pb[2] = 1.0
        pa, qa, pb, qb = (np.float32(x) for x in (pa, qa, pb, qb))
        n = np.zeros(3, np.float32); A = np.zeros((4, 3), np.float32); B = np.zeros((4, 3), np.float32)
        c = L.hsref_collide(oa, f(pa), f(qa), ob, f(pb), f(qb), n.ctypes.data, A.ctypes.data, B.ctypes.data)
        NA, DA, _ = half_spaces(oa, pa, qa)
        NB, DB, _ = half_spaces(ob, pb, qb)
        rows.append((oa, ob, *pa, *qa, *pb, *qb, c, *n))
        if c == 0:
            assert not polytopes_intersect(NA, DA, NB, DB, shrink=2e-3), "reported separate, but they share an interior point"
            continue
        hits += 1
        n64 = n.astype(np.float64)
        assert abs(np.linalg.norm(n64) - 1) < 1e-4
        depth = max(float((A[i].astype(np.float64) - B[i].astype(np.float64)) @ n64) for i in range(c))
        assert depth > -1e-4, depth
        # the manifold normal points from A to B: moving B by the depth (plus slack) along it separates the pair
        assert not polytopes_intersect(NA, DA, NB, DB + NB @ (n64 * (depth + 5e-3)), shrink=1e-4), (oa, ob, depth)
        # ... and it is a minimum-translation direction up to the face/edge preference: half of it does not separate
        if depth > 0.02:
            assert polytopes_intersect(NA, DA, NB, DB + NB @ (n64 * (0.45 * depth)), shrink=0.0)
        for i in range(c):
            assert np.all(NA @ A[i].astype(np.float64) <= DA + 2e-3), "contact on A lies outside A"
            assert np.all(NB @ B[i].astype(np.float64) <= DB + 2e-3), "contact on B lies outside B"
    assert hits > 150 and len(rows) - hits > 150
    path = os.path.join(GOLDEN, "narrowphase_cases.npz")
    if os.environ.get("HS_WRITE_GOLDEN") == "1":
        np.savez_compressed(path, cases=np.array(rows, np.float64))
    g = np.load(path)["cases"]
    assert np.array_equal(g, np.array(rows, np.float64)), "golden narrowphase cases reproduce"
