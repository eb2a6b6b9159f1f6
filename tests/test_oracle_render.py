"""Agent-view depth / RGB of the CPU restatement (oracle/hs_ref_render.hpp) against first principles.

The reference hands these tensors to Madrona's batch renderer, which is absent from the snapshot (PARITY UNPINNED);
first-party source fixes the camera (src/sim.cpp:1400-1403: 100 degrees vertical field of view, 0.5 above the
agent's origin), the base colours (src/mgr.cpp:621-647) and the light (src/mgr.cpp:657-659).  What is checked here
follows from those alone by pinhole geometry and Lambert's law, not from the ray caster's code."""
import math

import numpy as np

from test_oracle_first_principles import edit_scene, free_spot, make, put

TAN50 = math.tan(math.radians(50.0))
TO_LIGHT = np.array([-1.0, -1.0, 2.0]) / math.sqrt(6.0)
FLOOR, WALL = (0.5, 0.3, 0.3), (0.4, 0.4, 0.4)


def shade(base, n):
    k = 0.3 + 0.7 * max(0.0, float(np.dot(n, TO_LIGHT)))
    return [int(min(c * k, 1.0) * 255 + 0.5) for c in base]


def yaw(a):
    return (math.cos(a / 2), 0, 0, math.sin(a / 2))


def test_floor_depth_follows_the_pinhole_model(oracle):
    """A floor pixel in row py lies at view depth h / ((2 (py + 0.5) / H - 1) tan 50) for camera height h = z + 0.5,
    whatever its column; its colour is the floor material under the light."""
    ref = make(oracle)
    H, W = 64, 64
    depth, rgb = ref.render(W, H)
    b, _ = ref.bodies()
    assert depth.shape == (2, H, W, 1) and rgb.shape == (2, H, W, 4) and rgb.dtype == np.uint8
    floor_rgb = shade(FLOOR, (0, 0, 1))
    seen = 0
    for a in range(2):
        h = float(b[0, 11 + a, 2]) + 0.5
        for py in range(H // 2 + 1, H):
            cols = np.where((rgb[a, py, :, :3] == floor_rgb).all(axis=1))[0]
            want = h / ((2 * (py + 0.5) / H - 1) * TAN50)
            assert np.allclose(depth[a, py, cols, 0], want, rtol=2e-5), (a, py)
            seen += len(cols)
    assert seen > 1000
    assert (rgb[..., 3] == 255).all()


def test_wall_ahead_and_sky(oracle):
    """An agent at (x, y) that looks along +y at the outer wall (face at y = 17.8, 2.5 high): every wall pixel has view
    depth 17.8 - y and the wall's grey under the light; the rows whose rays pass above the wall's top edge are sky
    (depth 0, black, opaque)."""
    ref = make(oracle)
    x, y = free_spot(ref)

    def scene(r):
        put(r["agents"][0], [x, 15.0, 1.0], yaw(0.0))
    edit_scene(ref, scene)
    H, W = 64, 64
    depth, rgb = ref.render(W, H)
    w, info = ref.walls()
    top = [k for k in range(info[0, 0]) if abs(w[0, k, 1] - 18.0) < 1e-6 and w[0, k, 3] < 0.5]
    assert top, "outer wall at y = 18"
    wall_rgb = shade(WALL, (0, -1, 0))
    mask = (rgb[0, :, :, :3] == wall_rgb).all(axis=2)
    # nothing else stands within 2.8 in front of the camera in this column range: the centre columns see the wall
    centre = mask[:, W // 2 - 4:W // 2 + 4]
    assert centre.sum() > 100
    d = depth[0, :, W // 2 - 4:W // 2 + 4, 0][centre]
    assert np.allclose(d, 17.8 - 15.0, rtol=2e-5)
    # the top edge of the wall is 1.0 above the camera at depth 2.8: rows with v > 1 / 2.8 look over it
    for py in range(H // 2):
        v = (1 - 2 * (py + 0.5) / H) * TAN50
        if v > 1.0 / 2.8 + 0.02:
            assert (depth[0, py, W // 2 - 4:W // 2 + 4, 0] == 0).all()
            assert (rgb[0, py, W // 2 - 4:W // 2 + 4] == [0, 0, 0, 255]).all()


def test_agents_see_each_other_in_team_colours(oracle):
    """A hider and a seeker face each other 6 apart: each sees the other's near face (a 2 x 2 x 2 cube centred 0.5
    below the camera) at view depth 5, white for the hider and red-tinted for the seeker (the stand-in for the
    seeker's red face texture, src/mgr.cpp:641-645)."""
    ref = make(oracle)
    x, y = free_spot(ref, margin=4.0)
    types = ref.tensor("self_type").reshape(-1)          # AgentType: Seeker = 0, Hider = 1 (src/sim.hpp:138-141)

    def scene(r):
        put(r["agents"][0], [x, y - 3.0, 1.0], yaw(0.0))                 # looks along +y
        put(r["agents"][1], [x, y + 3.0, 1.0], yaw(math.pi))             # looks along -y
    edit_scene(ref, scene)
    depth, rgb = ref.render(64, 64)
    for me, other, n in ((0, 1, (0, -1, 0)), (1, 0, (0, 1, 0))):
        base = (1.0, 1.0, 1.0) if types[other] == 1 else (1.0, 0.3, 0.3)
        want = shade(base, n)
        blk = rgb[me, 30:38, 28:36, :3].reshape(-1, 3)
        assert (blk == want).all(), (me, blk[0], want)
        assert np.allclose(depth[me, 30:38, 28:36, 0], 5.0, rtol=2e-5)


def test_view_shape_and_inactive_agents(oracle):
    """Non-square views keep square pixels (u scales with W / H); views of agent slots without an agent are zero."""
    ref = oracle.RefSim(3, rand_seed=11, min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3)
    ref.init()
    depth, rgb = ref.render(32, 16)
    assert depth.shape == (18, 16, 32, 1)
    mask = ref.tensor("self_mask").reshape(-1)
    assert (mask == 0).any() and (mask == 1).any()
    for v in range(18):
        if mask[v] == 0:
            assert not depth[v].any() and not rgb[v].any()
        else:
            assert depth[v].any() and (rgb[v, :, :, 3] == 255).all()
    # the bottom row of a 32 x 16 view and of a 64 x 32 view of the same state show the floor at the depth its row implies
    d2, _ = ref.render(64, 32)
    v = int(np.argmax(mask))
    b, _ = ref.bodies()
    slot = 11 + v % 6
    h = float(b[v // 6, slot, 2]) + 0.5
    for dd, H in ((depth, 16), (d2, 32)):
        row = dd[v, H - 1, :, 0]
        want = h / ((2 * (H - 0.5) / H - 1) * TAN50)
        assert np.isclose(row, want, rtol=2e-5).sum() >= row.size // 2
