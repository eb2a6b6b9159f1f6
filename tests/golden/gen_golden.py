"""Generates the committed golden vectors from the CPU oracle IN THIS CONTAINER.

The reference ships no tests/fixtures and cannot be built, so these vectors pin
the build's own semantics: level layouts (integer/bit-exact) and a short scripted step sequence.
Run:  python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import hs_ref  # noqa: E402


def scripted_actions(step, rows):
    """Deterministic integer action stream (no RNG library involved)."""
    i = np.arange(rows, dtype=np.int64)
    a = np.zeros((rows, 5), np.int32)
    a[:, 0] = (i * 7 + step * 3) % 11
    a[:, 1] = (i * 5 + step * 2 + 1) % 11
    a[:, 2] = (i * 3 + step) % 11
    a[:, 3] = ((i + step) % 13 == 0).astype(np.int32)
    a[:, 4] = ((i * 2 + step) % 17 == 0).astype(np.int32)
    return a


def level_layouts(seed, flags, hiders, seekers):
    out = {}
    s = hs_ref.RefSim(8, sim_flags=flags, rand_seed=seed, min_hiders=1, max_hiders=hiders, min_seekers=1,
                      max_seekers=seekers)
    s.init()
    for ep in range(2):
        w, info = s.walls()
        b, m = s.bodies()
        out[f"walls_ep{ep}"] = w
        out[f"info_ep{ep}"] = info
        out[f"bodies_ep{ep}"] = b[:, :, :7]
        out[f"meta_ep{ep}"] = m
        out[f"seed_ep{ep}"] = s.tensor("seed").copy()
        out[f"self_type_ep{ep}"] = s.tensor("self_type").copy()
        out[f"self_mask_ep{ep}"] = s.tensor("self_mask").copy()
        s.tensor("reset")[:] = 1
        s.step()
    return out


def step_sequence():
    s = hs_ref.RefSim(8, sim_flags=0, rand_seed=3, min_hiders=2, max_hiders=3, min_seekers=1, max_seekers=3)
    s.init()
    rows = 8 * s.A
    keep = {}
    for step in range(48):
        s.tensor("action")[:] = scripted_actions(step, rows)
        s.step()
        if step in (0, 7, 23, 47):
            b, m = s.bodies()
            keep[f"bodies_{step}"] = b
            keep[f"meta_{step}"] = m
            for n in ("self_data", "lidar", "reward", "visible_boxes_mask", "visible_agents_mask", "box_data"):
                keep[f"{n}_{step}"] = s.tensor(n).copy()
    return keep


def main():
    np.savez_compressed(os.path.join(HERE, "levelgen_seed0.npz"), **level_layouts(0, 0, 3, 3))
    np.savez_compressed(os.path.join(HERE, "levelgen_seed5_flip_fixed.npz"), **level_layouts(5, 1 | 4, 3, 3))
    np.savez_compressed(os.path.join(HERE, "steps_seed3.npz"), **step_sequence())
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print(f, os.path.getsize(os.path.join(HERE, f)))


if __name__ == "__main__":
    main()
