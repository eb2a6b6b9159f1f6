"""ORACLE — TEST INFRASTRUCTURE ONLY.

ctypes driver for the CPU restatement (oracle/_build/libhs_ref.so).  Imported only by tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the product package.
PARITY UNPINNED against the real reference (engine source absent) — see DESIGN.md.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "_build", "libhs_ref.so")


def build(force=False):
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".hpp", ".cpp"))]
    if (not force and os.path.exists(_LIB_PATH)
            and os.path.getmtime(_LIB_PATH) >= max(os.path.getmtime(s) for s in srcs)):
        return _LIB_PATH
    subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB_PATH


class _Cfg(C.Structure):
    _fields_ = [("num_worlds", C.c_int32), ("sim_flags", C.c_uint32), ("rand_seed", C.c_uint32),
                ("min_hiders", C.c_int32), ("max_hiders", C.c_int32), ("min_seekers", C.c_int32),
                ("max_seekers", C.c_int32), ("world_offset", C.c_int32),
                ("skip_observations", C.c_int32), ("threads", C.c_int32)]


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        L = C.CDLL(_LIB_PATH)
        L.hsref_create.restype = C.c_void_p
        L.hsref_create.argtypes = [C.POINTER(_Cfg)]
        for n in ("hsref_destroy", "hsref_init", "hsref_step", "hsref_save_checkpoints", "hsref_load_checkpoints"):
            getattr(L, n).argtypes = [C.c_void_p]
            getattr(L, n).restype = None
        L.hsref_agents_per_world.argtypes = [C.c_void_p]
        L.hsref_agents_per_world.restype = C.c_int32
        L.hsref_tensor.argtypes = [C.c_void_p, C.c_int32]
        L.hsref_tensor.restype = C.c_void_p
        L.hsref_render.argtypes = [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_void_p]
        L.hsref_render.restype = None
        L.hsref_dump_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsref_dump_walls.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsref_threefry.argtypes = [C.c_uint32] * 4 + [C.c_void_p]
        L.hsref_rng_draws.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_void_p]
        L.hsref_sample_i32.argtypes = [C.c_uint32, C.c_uint32, C.c_int32, C.c_int32]
        L.hsref_sample_i32.restype = C.c_int32
        L.hsref_sincos.argtypes = [C.c_float, C.c_void_p, C.c_void_p]
        L.hsref_atan2.argtypes = [C.c_float, C.c_float]
        L.hsref_atan2.restype = C.c_float
        L.hsref_asin.argtypes = [C.c_float]
        L.hsref_asin.restype = C.c_float
        L.hsref_quat_to_euler.argtypes = [C.c_void_p, C.c_void_p]
        L.hsref_ray_body.argtypes = [C.c_int32] + [C.c_void_p] * 4
        L.hsref_ray_body.restype = C.c_float
        L.hsref_collide.argtypes = [C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p,
                                    C.c_void_p, C.c_void_p, C.c_void_p]
        L.hsref_collide.restype = C.c_int32
        L.hsref_hull_tables.argtypes = [C.c_int32] + [C.c_void_p] * 6
        L.hsref_hull_tables.restype = None
        L.hsref_object_params.argtypes = [C.c_int32, C.c_void_p]
        L.hsref_object_params.restype = None
        _lib = L
    return _lib


def object_params(obj):
    """(inv_mass, mu_s, mu_d, inv_inertia xyz) of SimObject `obj` as the oracle's solver uses them."""
    out = np.zeros(6, np.float32)
    lib().hsref_object_params(int(obj), out.ctypes.data)
    return out


def hull_tables(obj):
    """The oracle's hull of SimObject `obj` (identity pose; unit wall): dict of verts [nv,3], faces (list of index
    loops), normals [nf,3], edges [ne,3] = v0 v1 direction id, aabb (lo, hi)."""
    v = np.zeros((8, 3), np.float32); f = np.full((6, 4), -1, np.int32); c = np.zeros(4, np.int32)
    n = np.zeros((6, 3), np.float32); e = np.zeros((12, 3), np.int32); a = np.zeros(6, np.float32)
    lib().hsref_hull_tables(int(obj), v.ctypes.data, f.ctypes.data, c.ctypes.data, n.ctypes.data, e.ctypes.data, a.ctypes.data)
    nv, nf, ne, _ = c
    return {"verts": v[:nv], "faces": [[int(i) for i in row if i >= 0] for row in f[:nf]], "normals": n[:nf],
            "edges": e[:ne], "aabb": (a[:3].copy(), a[3:].copy())}


# name -> (export id, dtype, trailing shape, per-agent?)   (mgr.cpp:1062-1331)
TENSORS = {
    "reset": (0, np.int32, (1,), False),
    "prep_counter": (1, np.int32, (1,), True),
    "action": (2, np.int32, (5,), True),
    "self_data": (3, np.float32, (13,), True),
    "self_type": (4, np.int32, (1,), True),
    "self_mask": (5, np.float32, (1,), True),
    "agent_data": (6, np.float32, (5, 14), True),
    "box_data": (7, np.float32, (9, 17), True),
    "ramp_data": (8, np.float32, (2, 14), True),
    "visible_agents_mask": (9, np.float32, (5, 1), True),
    "visible_boxes_mask": (10, np.float32, (9, 1), True),
    "visible_ramps_mask": (11, np.float32, (2, 1), True),
    "lidar": (12, np.float32, (30,), True),
    "seed": (13, np.int32, (2,), True),
    "reward": (14, np.float32, (1,), True),
    "done": (15, np.int32, (1,), True),
    "global_positions": (16, np.float32, (17, 2), False),
    "policy_assignments": (17, np.int32, (1,), True),
    "episode_result": (18, np.float32, (2,), False),
    "ckpt_ctrl": (19, np.int32, (1,), False),          # CheckpointControl::trigger (the u8 [N,4] tensor viewed as i32)
    "ckpt": (20, np.uint8, (1392,), False),            # Checkpoint bytes (oracle/hs_ref_ckpt.hpp)
}


class RefSim:
    """CPU restatement with the HideAndSeekSimulator call shape (bindings.cpp:32-75)."""

    def __init__(self, num_worlds, sim_flags=0, rand_seed=0, min_hiders=2, max_hiders=2, min_seekers=2,
                 max_seekers=2, world_offset=0, skip_observations=False, threads=1):
        self.N = int(num_worlds)
        cfg = _Cfg(self.N, int(sim_flags), int(rand_seed), min_hiders, max_hiders, min_seekers, max_seekers,
                   int(world_offset), int(bool(skip_observations)), int(threads))
        self._h = lib().hsref_create(C.byref(cfg))
        if not self._h:
            raise ValueError("invalid oracle configuration")
        self.A = lib().hsref_agents_per_world(self._h)

    def close(self):
        if getattr(self, "_h", None):
            lib().hsref_destroy(self._h)
            self._h = None

    __del__ = close

    def init(self):
        lib().hsref_init(self._h)

    def step(self):
        lib().hsref_step(self._h)

    def tensor(self, name):
        """Zero-copy numpy view of an exported column."""
        eid, dt, tail, per_agent = TENSORS[name]
        rows = self.N * self.A if per_agent else self.N
        ptr = lib().hsref_tensor(self._h, eid)
        n = rows * int(np.prod(tail))
        ctype = {np.int32: C.c_int32, np.float32: C.c_float, np.uint8: C.c_uint8}[dt]
        arr = np.ctypeslib.as_array((ctype * n).from_address(ptr))
        return arr.reshape((rows,) + tail)

    def save_checkpoints(self):
        """SaveCheckpoints task graph for the worlds whose ckpt_ctrl trigger is set (sim.cpp:1315-1322)."""
        lib().hsref_save_checkpoints(self._h)

    def load_checkpoints(self):
        """LoadCheckpoints task graph (sim.cpp:1324-1333): restore triggered worlds, refresh observations."""
        lib().hsref_load_checkpoints(self._h)

    def render(self, width=64, height=64):
        """Agent views of the current state (hs_ref_render.hpp): depth [N*A,H,W,1] f32, rgb [N*A,H,W,4] u8."""
        depth = np.zeros((self.N * self.A, height, width, 1), np.float32)
        rgb = np.zeros((self.N * self.A, height, width, 4), np.uint8)
        lib().hsref_render(self._h, width, height, depth.ctypes.data, rgb.ctypes.data)
        return depth, rgb

    def bodies(self):
        b = np.zeros((self.N, 17, 13), np.float32)
        m = np.zeros((self.N, 17, 3), np.int32)
        lib().hsref_dump_bodies(self._h, b.ctypes.data, m.ctypes.data)
        return b, m

    def walls(self):
        w = np.zeros((self.N, 36, 4), np.float32)
        info = np.zeros((self.N, 8), np.int32)
        lib().hsref_dump_walls(self._h, w.ctypes.data, info.ctypes.data)
        return w, info
