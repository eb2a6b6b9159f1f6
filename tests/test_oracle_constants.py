"""The oracle's BEHAVIOUR against the constants the reference states in first-party source (tests/golden/sim_constants.json,
read out of src/sim.cpp / src/sim.hpp by tests/golden/gen_sim_constants_fixture.py): step length and substeps through free
fall, the action -> force mapping of both movement systems through the acceleration of an agent in mid-air, the episode
timeline, the slot capacities."""
import json
import os

import numpy as np
import pytest

from conftest import GOLDEN
from test_oracle_first_principles import edit_scene, make, put

K = json.load(open(os.path.join(GOLDEN, "sim_constants.json")))


def test_free_fall_follows_the_reference_step_and_gravity(oracle):
    """One step = numPhysicsSubsteps semi-implicit Euler substeps of deltaT / n under gravity: v = g deltaT, z drops by
    g h^2 n (n + 1) / 2."""
    ref = make(oracle)
    edit_scene(ref, lambda r: put(r["agents"][0], [0, 0, 20]))
    ref.tensor("action")[0] = [K["movement"]["centre"]] * 3 + [0, 0]
    ref.step()
    b, _ = ref.bodies()
    n, h = K["numPhysicsSubsteps"], K["deltaT"] / K["numPhysicsSubsteps"]
    assert abs(b[0, 11, 9] - K["gravity_z"] * K["deltaT"]) < 2e-3          # (velocities are pose differences in f32 at z = 20)
    assert abs(b[0, 11, 2] - (20 + K["gravity_z"] * h * h * n * (n + 1) / 2)) < 1e-4


@pytest.mark.parametrize("mode,flags", [("movement", 1 | 2), ("instant_movement", 1 | 2 | 8)])
def test_action_buckets_map_to_the_reference_forces(oracle, mode, flags):
    """An agent in mid-air (mass 1: tests/golden/object_table.json) accelerates by exactly force / mass: after one step
    v = (a - centre) * move_max / half_buckets * deltaT along its own axes, for every bucket."""
    M = K[mode]
    per_bucket = M["move_max"] / M["half_buckets"]
    for a in range(M["buckets"]):
        ref = make(oracle, sim_flags=flags)
        edit_scene(ref, lambda r: put(r["agents"][0], [0, 0, 50]))
        ref.tensor("action")[0] = [a, M["centre"], M["centre"], 0, 0]
        ref.step()
        b, _ = ref.bodies()
        want = (a - M["centre"]) * per_bucket * K["deltaT"]
        # (flags & 8 zeroes the agent's xy velocity after the step: read the displacement instead, x = f h^2 n (n + 1) / 2)
        n, h = K["numPhysicsSubsteps"], K["deltaT"] / K["numPhysicsSubsteps"]
        if flags & 8:
            assert abs(b[0, 11, 0] - (a - M["centre"]) * per_bucket * h * h * n * (n + 1) / 2) < 1e-4 * max(1.0, abs(want)), (a, b[0, 11, :3])
        else:
            assert abs(b[0, 11, 7] - want) < 1e-3 * max(1.0, abs(want)), (a, b[0, 11, 7], want)
        assert abs(b[0, 11, 8]) < 1e-6


def test_episode_timeline_and_capacities(oracle):
    ref = oracle.RefSim(4, rand_seed=2, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3)
    ref.init()
    assert ref.A == K["maxAgents"] and ref.tensor("box_data").shape[1] == K["maxBoxes"] and ref.tensor("ramp_data").shape[1] == K["maxRamps"]
    assert (ref.tensor("prep_counter") == K["numPrepSteps"]).all()
    for s in range(K["episodeLen"]):
        ref.step()
        if s == K["numPrepSteps"] - 2:
            assert (ref.tensor("reward") == 0).all(), "no reward while the seekers are frozen"
        if s == K["numPrepSteps"] - 1:
            assert (ref.tensor("reward") != 0).all() and (ref.tensor("prep_counter") == 0).all()
        assert (ref.tensor("done") == (1 if s == K["episodeLen"] - 1 else 0)).all(), s
