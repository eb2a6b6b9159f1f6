"""Shared by tests/test_oracle_hulls.py and tests/test_gpu_hulls.py: compare a hull table (vertices, face loops, face
normals, edges with direction ids) with the reference's collision mesh in tests/golden/hulls.npz.  Vertex NUMBERING is
the implementation's own business; what must agree is the geometry: the vertex set, every face as a loop over the
same points with the same (outward) orientation, the edge set, and the vertex extrema."""
import os

import numpy as np

import json

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "hulls.npz")
OBJECT_TABLE = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "object_table.json")
# SimObject id -> collision mesh, as the reference's loadPhysicsObjects assigns them (tests/golden/object_table.json)
NAMES = {o["id"]: o["mesh"].replace("_collision.obj", "") for o in json.load(open(OBJECT_TABLE))["objects"].values() if o["mesh"]}
assert NAMES == {2: "cube", 3: "wall", 4: "agent", 5: "agent", 6: "ramp", 7: "elongated"}


def check_object_params(obj, params):
    """params = (inv_mass, mu_s, mu_d, inv_inertia xyz) of SimObject id `obj` against the reference's table."""
    tab = {o["id"]: o for o in json.load(open(OBJECT_TABLE))["objects"].values()}[obj]
    assert float(params[0]) == tab["inv_mass"] and float(params[1]) == tab["mu_s"] and float(params[2]) == tab["mu_d"], (obj, params, tab)
    for k, ax in enumerate("xyz"):
        if ax in tab["inv_inertia_zeroed"]:
            assert float(params[3 + k]) == 0.0, (obj, ax)
    if tab["inv_inertia_zeroed"] == ["x", "y"]:
        assert float(params[5]) > 0.0, "a yaw-only body still turns about z"
    if tab["inv_mass"] > 0 and tab["mesh"] and not tab["inv_inertia_zeroed"]:
        assert (params[3:6] > 0).all(), "a movable hull has a full inverse inertia"


def newell(points):
    p = np.asarray(points, np.float64)
    n = np.zeros(3)
    for i in range(len(p)):
        a, b = p[i], p[(i + 1) % len(p)]
        n += np.cross(a, b)
    return n / np.linalg.norm(n)


def canon_loop(points):
    """A face loop as a tuple of points starting at its smallest point, orientation kept."""
    pts = [tuple(float(x) for x in q) for q in points]
    k = pts.index(min(pts))
    return tuple(pts[k:] + pts[:k])


def check_hull(obj, verts, faces, normals, edges):
    g = np.load(GOLDEN)
    name = NAMES[obj]
    gv, gf = g[f"{name}_v"], g[f"{name}_f"]
    verts = np.asarray(verts, np.float32)
    assert {tuple(v) for v in verts.tolist()} == {tuple(v) for v in gv.tolist()}, f"{name}: vertex set"
    assert len(verts) == len(gv)
    gold_loops = {canon_loop(gv[[i for i in loop if i >= 0]]) for loop in gf.tolist()}
    have_loops = {canon_loop(verts[loop]) for loop in faces}
    assert have_loops == gold_loops, f"{name}: face loops (points and orientation)"
    for loop, n in zip(faces, np.asarray(normals, np.float64)):
        assert abs(np.linalg.norm(n) - 1) < 1e-6
        assert float(newell(verts[loop]) @ n) > 1 - 1e-6, f"{name}: stored normal vs loop orientation"
        c = verts.astype(np.float64).mean(axis=0)
        assert float((verts[loop[0]] - c) @ n) > 0, f"{name}: normal points outward"
    gold_edges = set()
    for loop in gf.tolist():
        loop = [i for i in loop if i >= 0]
        for i in range(len(loop)):
            a, b = tuple(gv[loop[i]].tolist()), tuple(gv[loop[(i + 1) % len(loop)]].tolist())
            gold_edges.add(frozenset((a, b)))
    have_edges = {frozenset((tuple(verts[a].tolist()), tuple(verts[b].tolist()))) for a, b, _ in np.asarray(edges).tolist()}
    assert have_edges == gold_edges and len(edges) == len(gold_edges), f"{name}: edge set"
    dirs = {}
    for a, b, d in np.asarray(edges).tolist():
        u = verts[b].astype(np.float64) - verts[a]
        u /= np.linalg.norm(u)
        if d in dirs:
            assert abs(abs(float(u @ dirs[d])) - 1) < 1e-6, f"{name}: edges of one direction id are parallel"
        else:
            dirs[d] = u
    return g[f"{name}_lo"], g[f"{name}_hi"]
