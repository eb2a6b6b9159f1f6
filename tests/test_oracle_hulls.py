"""The one reference-held pin of the hot path: the collision hulls the reference loads
(data/*_collision.obj via loadPhysicsObjects, src/mgr.cpp:441-588), committed as tests/golden/hulls.npz by
tests/golden/gen_hull_fixture.py.  The oracle's hull tables (oracle/hs_ref_phys.hpp: kWedgeV, box vertices from
obj_half_extents, face loops, edges) and the object-space AABBs its level generator uses (object_aabb) must describe
exactly those meshes."""
import numpy as np
import pytest

from hull_check import check_hull, check_object_params, GOLDEN


@pytest.mark.parametrize("obj", [2, 3, 4, 5, 6, 7])
def test_oracle_hull_tables_match_the_reference_meshes(oracle, obj):
    t = oracle.hull_tables(obj)
    lo, hi = check_hull(obj, t["verts"], t["faces"], t["normals"], t["edges"])
    assert np.array_equal(t["aabb"][0], lo) and np.array_equal(t["aabb"][1], hi), "object_aabb = vertex extrema of the mesh"
    assert np.array_equal(t["verts"].min(axis=0), lo) and np.array_equal(t["verts"].max(axis=0), hi)


def test_fixture_is_what_the_reference_ships():
    """Shape of the fixture itself: five hulls, boxes with 8 vertices / 6 quads, the ramp a wedge of 6 vertices / 3 quads +
    2 triangles (SURVEY Appendix A's extrema)."""
    g = np.load(GOLDEN)
    for name in ("cube", "wall", "agent", "elongated"):
        assert g[f"{name}_v"].shape == (8, 3) and g[f"{name}_f"].shape == (6, 4) and (g[f"{name}_f"] >= 0).all()
    assert g["ramp_v"].shape == (6, 3) and sorted((row >= 0).sum() for row in g["ramp_f"]) == [3, 3, 4, 4, 4]
    assert g["wall_lo"].tolist() == [-1, -1, 0] and g["wall_hi"].tolist() == [1, 1, 2.5]
    assert g["ramp_lo"].tolist() == [-1, -2, -1] and g["elongated_hi"].tolist() == [4, 0.75, 1]


@pytest.mark.parametrize("obj", range(8))
def test_oracle_object_table_matches_the_reference(oracle, obj):
    """Inverse masses, friction coefficients and the yaw-only "HACK" of loadPhysicsObjects (src/mgr.cpp:476-584), read out of
    the reference's source into tests/golden/object_table.json by tests/golden/gen_object_table_fixture.py."""
    check_object_params(obj, oracle.object_params(obj))
