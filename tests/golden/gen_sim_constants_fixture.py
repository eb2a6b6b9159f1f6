"""Reads the simulation constants the REFERENCE states in first-party source into tests/golden/sim_constants.json:
src/sim.cpp:14-17 (deltaT, substeps, preparation steps, episode length), :1355-1361 (gravity), src/sim.hpp:39-41 (capacities)
and the action -> force mappings of movementSystem / instantMovementSystem (src/sim.cpp:202-254).  Run in the build container:

    python tests/golden/gen_sim_constants_fixture.py

Only the .json travels (a dozen numbers).  tests/test_oracle_constants.py checks the oracle's BEHAVIOUR against them."""
import json
import os
import re

HERE = os.path.dirname(os.path.abspath(__file__))
REF = "/root/reference/src"


def num(expr):
    return float(eval(re.sub(r"(\d)f\b", r"\1", expr.replace(".f", ".0")), {"__builtins__": {}}))


def main():
    cpp = open(os.path.join(REF, "sim.cpp")).read()
    hpp = open(os.path.join(REF, "sim.hpp")).read()
    out = {"source": "src/sim.cpp:14-17, 202-254, 1355-1361; src/sim.hpp:39-41"}
    out["deltaT"] = num(re.search(r"constexpr inline float deltaT = ([^;]+);", cpp).group(1))
    for k in ("numPhysicsSubsteps", "numPrepSteps", "episodeLen"):
        out[k] = int(re.search(r"constexpr inline CountT %s = (\d+);" % k, cpp).group(1))
    for k in ("maxBoxes", "maxRamps", "maxAgents"):
        out[k] = int(re.search(r"constexpr int32_t %s = (\d+);" % k, hpp).group(1))
    out["gravity_z"] = num(re.search(r"numPhysicsSubsteps,\s*(-?[\d.]+)\s*\*\s*math::up", cpp).group(1))
    for name, fn in (("movement", "movementSystem"), ("instant_movement", "instantMovementSystem")):
        body = cpp[cpp.index("inline void %s(" % fn):]
        body = body[:body.index("\n}\n")]
        b = int(re.search(r"discrete_action_buckets = (\d+);", body).group(1))
        out[name] = {"buckets": b, "half_buckets": b // 2,
                     "move_max": num(re.search(r"move_discrete_action_max = ([\d.]+);", body).group(1)),
                     "turn_max": num(re.search(r"turn_discrete_action_max = ([\d.]+);", body).group(1)),
                     "centre": int(re.search(r"\(action\.x - (\d+)\)", body).group(1))}
    json.dump(out, open(os.path.join(HERE, "sim_constants.json"), "w"), indent=1, sort_keys=True)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
