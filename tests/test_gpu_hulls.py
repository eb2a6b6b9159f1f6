"""The device's hull tables — bit-packed topology and closed-form vertices of csrc/hs_collide.h, dumped through
hs_debug_dump_hull — against the reference's collision meshes (tests/golden/hulls.npz, made from
/root/reference/data/*_collision.obj by tests/golden/gen_hull_fixture.py), and against the oracle's tables index by index
(the contact manifolds carry vertex / face / edge indices across the two implementations)."""
import ctypes as C

import numpy as np
import pytest

from hull_check import check_hull, check_object_params

pytestmark = pytest.mark.gpu


def device_hull(obj):
    import gpu_hideseek
    L = gpu_hideseek._native.load()
    L.hs_debug_dump_hull.argtypes = [C.c_int32] + [C.c_void_p] * 6
    L.hs_debug_dump_hull.restype = C.c_int32
    v = np.zeros((8, 3), np.float32); f = np.full((6, 4), -1, np.int32); c = np.zeros(8, np.int32)
    n = np.zeros((6, 3), np.float32); e = np.zeros((12, 3), np.int32); l = np.zeros((8, 3), np.float32)
    assert L.hs_debug_dump_hull(obj, v.ctypes.data, f.ctypes.data, c.ctypes.data, n.ctypes.data, e.ctypes.data, l.ctypes.data) == 0
    nv, nf, ne, ned = c[:4]
    return {"verts": v[:nv], "faces": [[int(i) for i in row if i >= 0] for row in f[:nf]], "normals": n[:nf], "edges": e[:ne],
            "local": l[:nv], "ned": int(ned)}


@pytest.mark.parametrize("obj", [2, 3, 4, 5, 6, 7])
def test_device_hull_tables_match_the_reference_meshes_and_the_oracle(oracle, obj):
    t = device_hull(obj)
    lo, hi = check_hull(obj, t["verts"], t["faces"], t["normals"], t["edges"])
    assert np.array_equal(t["verts"].min(axis=0), lo) and np.array_equal(t["verts"].max(axis=0), hi)
    if obj != 3:
        assert np.array_equal(t["local"], t["verts"]), "hull_local_vertex = the hull's vertices at the identity pose"
    o = oracle.hull_tables(obj)
    assert np.array_equal(t["verts"], o["verts"]) and t["faces"] == o["faces"] and np.array_equal(t["edges"], o["edges"])
    assert np.array_equal(t["normals"].view(np.int32) & 0x7fffffff, o["normals"].view(np.int32) & 0x7fffffff)   # (sign of zero aside)
    assert np.array_equal(np.sign(t["normals"]), np.sign(o["normals"]))


@pytest.mark.parametrize("obj", range(8))
def test_device_object_table_matches_the_reference_and_the_oracle(oracle, obj):
    import gpu_hideseek
    L = gpu_hideseek._native.load()
    L.hs_debug_object_params.argtypes = [C.c_int32, C.c_void_p]
    L.hs_debug_object_params.restype = C.c_int32
    out = np.zeros(6, np.float32)
    assert L.hs_debug_object_params(obj, out.ctypes.data) == 0
    check_object_params(obj, out)
    assert np.array_equal(out.view(np.int32), oracle.object_params(obj).view(np.int32))
