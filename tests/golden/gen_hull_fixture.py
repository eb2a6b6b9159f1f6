"""Turns the collision hulls the REFERENCE ships — /root/reference/data/{cube,wall,agent,ramp,elongated}_collision.obj,
loaded by loadPhysicsObjects (src/mgr.cpp:441-588) — into tests/golden/hulls.npz.  Run in the build container:

    python tests/golden/gen_hull_fixture.py

Only the .npz travels (data: vertex coordinates, face loops, vertex extrema; nothing of the reference's source).  It is the
one reference-held pin of the hot path: tests/test_oracle_hulls.py checks the oracle's hull tables and object AABBs
against it, tests/test_gpu_hulls.py the device's packed topology (hs_debug_dump_hull)."""
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
DATA = "/root/reference/data"
# SimObject id (src/sim.hpp:78-88) -> collision mesh (src/mgr.cpp:476-559)
HULLS = {"cube": (2, "cube_collision.obj"), "wall": (3, "wall_collision.obj"), "agent": (4, "agent_collision.obj"),
         "ramp": (6, "ramp_collision.obj"), "elongated": (7, "elongated_collision.obj")}


def parse_obj(path):
    v, f = [], []
    for line in open(path):
        t = line.split()
        if not t:
            continue
        if t[0] == "v":
            v.append([float(x) for x in t[1:4]])
        elif t[0] == "f":
            f.append([int(x.split("/")[0]) - 1 for x in t[1:]])
    return np.asarray(v, np.float32), f


def main():
    out = {}
    for name, (obj, fname) in HULLS.items():
        v, faces = parse_obj(os.path.join(DATA, fname))
        fa = np.full((len(faces), 4), -1, np.int32)
        for i, loop in enumerate(faces):
            assert len(loop) <= 4
            fa[i, :len(loop)] = loop
        out[f"{name}_obj"] = np.int32(obj)
        out[f"{name}_v"] = v
        out[f"{name}_f"] = fa
        out[f"{name}_lo"] = v.min(axis=0)
        out[f"{name}_hi"] = v.max(axis=0)
        print(name, obj, v.shape, fa.shape, v.min(axis=0), v.max(axis=0))
    np.savez_compressed(os.path.join(HERE, "hulls.npz"), **out)


if __name__ == "__main__":
    main()
