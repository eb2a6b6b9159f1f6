"""Turns the object table the REFERENCE states in source — loadPhysicsObjects, /root/reference/src/mgr.cpp:441-588, and the
SimObject enumeration, src/sim.hpp:78-88 — into tests/golden/object_table.json: per SimObject the inverse mass, the static /
dynamic friction coefficients, the collision mesh it is built from and whether the "HACK" lines zero its inverse inertia
about x and y (yaw-only bodies).  Run in the build container:

    python tests/golden/gen_object_table_fixture.py

Only the .json travels: a dozen numbers read out of the reference, not its text.  tests/test_oracle_hulls.py checks the
oracle's tables against it, tests/test_gpu_hulls.py the device's."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def fnum(s):
    return float(s.rstrip("f"))


def main():
    sim = open(os.path.join(REF, "sim.hpp")).read()
    enum = re.search(r"enum class SimObject\s*:\s*uint32_t\s*\{(.*?)\}", sim, re.S).group(1)
    names = [n.strip() for n in enum.split(",") if n.strip() and n.strip() != "NumObjects"]
    ids = {n: i for i, n in enumerate(names)}
    mgr = open(os.path.join(REF, "mgr.cpp")).read()
    body = mgr[mgr.index("static void loadPhysicsObjects"):mgr.index("static void loadRenderObjects")]
    meshes = re.findall(r'/\s*"(\w+_collision\.obj)"', body)
    table = {}
    for m in re.finditer(r"src_objs\[\(uint32_t\)SimObject::(\w+)\]\s*=\s*\{(.*?)\};", body, re.S):      # primitives (sphere, plane)
        blk = m.group(2)
        table[m.group(1)] = {"id": ids[m.group(1)], "mesh": None,
                             "inv_mass": fnum(re.search(r"\.invMass\s*=\s*([\d.]+f?)", blk).group(1)),
                             "mu_s": fnum(re.search(r"\.muS\s*=\s*([\d.]+f?)", blk).group(1)),
                             "mu_d": fnum(re.search(r"\.muD\s*=\s*([\d.]+f?)", blk).group(1))}
    for m in re.finditer(r"src_objs\[\(uint32_t\)SimObject::(\w+)\]\s*=\s*setupHull\((\d+),\s*([\d.]+f?),\s*\{\s*\.muS\s*=\s*([\d.]+f?),\s*\.muD\s*=\s*([\d.]+f?)", body):
        table[m.group(1)] = {"id": ids[m.group(1)], "mesh": meshes[int(m.group(2))], "inv_mass": fnum(m.group(3)),
                             "mu_s": fnum(m.group(4)), "mu_d": fnum(m.group(5))}
    for name in table:
        hack = {ax for ax in "xyz" if re.search(r"SimObject::%s\]\.mass\.invInertiaTensor\.%s\s*=\s*0\.f" % (name, ax), body)}
        table[name]["inv_inertia_zeroed"] = sorted(hack)
    assert sorted(table) == sorted(names), (sorted(table), names)
    json.dump({"source": "src/mgr.cpp:441-588 loadPhysicsObjects, src/sim.hpp:78-88 SimObject", "objects": table},
              open(os.path.join(HERE, "object_table.json"), "w"), indent=1, sort_keys=True)
    for n in names:
        print(n, table[n])


if __name__ == "__main__":
    main()
