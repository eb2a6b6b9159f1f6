"""gpu_hideseek — Python face of the MI355X-native batch hide-and-seek simulator.

Mirrors the reference's nanobind module (src/bindings.cpp:20-121): ``HideAndSeekSimulator`` with
the same keyword constructor, ``init`` / ``step``, the 21 tensor getters, ``SimFlags`` and
``madrona.ExecMode`` so that scripts/benchmark.py runs unchanged.  Everything is forwarded through
the C ABI of ``libhideseek.so`` (include/hideseek.h) with ctypes; tensors are zero-copy views of
the simulator's HBM buffers exported through DLPack (``Tensor.to_torch`` / ``Tensor.to_jax``).

There is no CPU execution path: if the HIP library or a GPU is missing the constructor raises.
"""
import atexit
import ctypes as C
import enum
import os
import weakref

__all__ = ["HideAndSeekSimulator", "SimFlags", "madrona", "Tensor", "library_path"]

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.environ.get("HS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libhideseek.so")


def library_path():
    return _LIB_PATH


class SimFlags(enum.IntFlag):
    """src/sim_flags.hpp:7-13 (nb::is_arithmetic enum, bindings.cpp:23-29)."""
    Default = 0
    UseFixedWorld = 1 << 0
    IgnoreEpisodeLength = 1 << 1
    RandomFlipTeams = 1 << 2
    ZeroAgentVelocity = 1 << 3
    # build-side extension (include/hideseek.h): skip the observation nodes
    ExtSkipObservations = 1 << 16


class _ExecMode(enum.IntEnum):
    CPU = 0
    CUDA = 1   # the reference's name for "GPU"; here it means HIP on gfx950


class _HsConfig(C.Structure):
    _fields_ = [("exec_mode", C.c_int32), ("gpu_id", C.c_int32), ("num_worlds", C.c_int32),
                ("sim_flags", C.c_uint32), ("rand_seed", C.c_uint32),
                ("min_hiders", C.c_int32), ("max_hiders", C.c_int32),
                ("min_seekers", C.c_int32), ("max_seekers", C.c_int32),
                ("num_pbt_policies", C.c_int32), ("enable_batch_renderer", C.c_int32),
                ("batch_render_width", C.c_int32), ("batch_render_height", C.c_int32),
                ("world_offset", C.c_int32)]


class _HsTensorDesc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 4), ("gpu_id", C.c_int32)]


# ---- DLPack (dlpack.h v0.8 ABI) built with ctypes: no torch types cross the C ABI ----
class _DLDevice(C.Structure):
    _fields_ = [("device_type", C.c_int32), ("device_id", C.c_int32)]


class _DLDataType(C.Structure):
    _fields_ = [("code", C.c_uint8), ("bits", C.c_uint8), ("lanes", C.c_uint16)]


class _DLTensor(C.Structure):
    _fields_ = [("data", C.c_void_p), ("device", _DLDevice), ("ndim", C.c_int32), ("dtype", _DLDataType),
                ("shape", C.POINTER(C.c_int64)), ("strides", C.POINTER(C.c_int64)), ("byte_offset", C.c_uint64)]


class _DLManagedTensor(C.Structure):
    pass


_DLDeleter = C.CFUNCTYPE(None, C.POINTER(_DLManagedTensor))
_DLManagedTensor._fields_ = [("dl_tensor", _DLTensor), ("manager_ctx", C.c_void_p), ("deleter", _DLDeleter)]

_kDLROCM = 10
_DTYPES = {0: (0, 32, "int32"), 1: (2, 32, "float32"), 2: (1, 8, "uint8")}   # id -> (code, bits, name)

# DLManagedTensor records handed out so far.  They are tiny, non-owning and must outlive every consumer
# (torch may run the deleter while the interpreter is shutting down, so the deleter is a C no-op in
# libhideseek and the records are simply kept for the life of the process).
_live_exports = []


class Tensor:
    """Counterpart of madrona::py::Tensor (src/mgr.cpp:824-842): a non-owning view."""

    def __init__(self, owner, desc):
        self._owner = owner            # keeps the simulator (and so the memory) alive
        self.ptr = desc.ptr
        self.dtype_id = desc.dtype
        self.shape = tuple(int(desc.dims[i]) for i in range(desc.ndim))
        self.gpu_id = desc.gpu_id

    @property
    def dtype(self):
        return _DTYPES[self.dtype_id][2]

    def __dlpack_device__(self):
        return (_kDLROCM, self.gpu_id)

    def __dlpack__(self, stream=None, **_):
        code, bits, _name = _DTYPES[self.dtype_id]
        nd = len(self.shape)
        shape = (C.c_int64 * nd)(*self.shape)
        mt = _DLManagedTensor()
        mt.dl_tensor.data = self.ptr
        mt.dl_tensor.device = _DLDevice(_kDLROCM, self.gpu_id)
        mt.dl_tensor.ndim = nd
        mt.dl_tensor.dtype = _DLDataType(code, bits, 1)
        mt.dl_tensor.shape = C.cast(shape, C.POINTER(C.c_int64))
        mt.dl_tensor.strides = None
        mt.dl_tensor.byte_offset = 0
        mt.manager_ctx = None
        mt.deleter = C.cast(_load().hs_dlpack_noop_deleter, _DLDeleter)
        _live_exports.append((mt, shape))
        new_capsule = C.pythonapi.PyCapsule_New
        new_capsule.restype = C.py_object
        new_capsule.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        return new_capsule(C.addressof(mt), b"dltensor", None)

    def to_torch(self):
        import torch
        t = torch.from_dlpack(self)
        return t

    def to_jax(self):
        import jax.dlpack
        return jax.dlpack.from_dlpack(self)

    def __repr__(self):
        return f"Tensor(shape={self.shape}, dtype={self.dtype}, gpu={self.gpu_id}, ptr=0x{self.ptr or 0:x})"


_lib = None


def _load():
    global _lib
    if _lib is not None:
        return _lib
    # PyTorch-ROCm wheels bundle their own HIP runtime (same SONAME as the system one).  If libhideseek pulled the
    # system copy in first, a later `import torch` would find "No HIP GPUs" — and the reference's scripts import
    # gpu_hideseek before torch (scripts/benchmark.py:1-2).  Loading torch first makes both share one runtime.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            f"{_LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). gpu_hideseek has no CPU fallback.")
    L = C.CDLL(_LIB_PATH)
    L.hs_create.argtypes = [C.POINTER(_HsConfig), C.POINTER(C.c_void_p)]
    L.hs_create.restype = C.c_int32
    L.hs_destroy.argtypes = [C.c_void_p]
    L.hs_destroy.restype = None
    for n in ("hs_init", "hs_step", "hs_save_checkpoints", "hs_load_checkpoints"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int32
    for n in ("hs_save_checkpoint", "hs_load_checkpoint"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int32]
        getattr(L, n).restype = C.c_int32
    for n in ("hs_jax_init", "hs_jax_step", "hs_jax_save_checkpoints", "hs_jax_load_checkpoints"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        getattr(L, n).restype = C.c_int32
    L.hs_step_async.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_step_async.restype = C.c_int32
    L.hs_get_tensor.argtypes = [C.c_void_p, C.c_int32, C.POINTER(_HsTensorDesc)]
    L.hs_get_tensor.restype = C.c_int32
    L.hs_trigger_reset.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.hs_trigger_reset.restype = C.c_int32
    L.hs_set_action.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.hs_set_action.restype = C.c_int32
    L.hs_agents_per_world.argtypes = [C.c_void_p]
    L.hs_agents_per_world.restype = C.c_int32
    L.hs_debug_dump_bodies.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_debug_dump_bodies.restype = C.c_int32
    L.hs_debug_dump_walls.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
    L.hs_debug_dump_walls.restype = C.c_int32
    L.hs_set_profiling.argtypes = [C.c_void_p, C.c_int32]
    L.hs_set_profiling.restype = C.c_int32
    L.hs_last_step_kernel_ms.argtypes = [C.c_void_p, C.POINTER(C.c_float * 3)]
    L.hs_last_step_kernel_ms.restype = C.c_int32
    L.hs_last_error.restype = C.c_char_p
    L.hs_version.restype = C.c_char_p
    _lib = L
    return L


def _check(rc):
    if rc != 0:
        msg = _load().hs_last_error().decode()
        if rc == 1:
            raise ValueError(msg)
        if rc == 3:
            raise NotImplementedError(msg)
        raise RuntimeError(f"libhideseek error {rc}: {msg}")


# ExportID (src/sim.hpp:45-68) + renderer outputs
_EXPORTS = dict(reset=0, prep_counter=1, action=2, self_data=3, self_type=4, self_mask=5, agent_data=6,
                box_data=7, ramp_data=8, visible_agents_mask=9, visible_boxes_mask=10, visible_ramps_mask=11,
                lidar=12, seed=13, reward=14, done=15, global_positions=16, policy_assignments=17,
                episode_result=18, ckpt_ctrl=19, ckpt=20, depth=21, rgb=22)


_live_sims = weakref.WeakSet()


@atexit.register
def _close_all():
    # destroy simulators while the HIP runtime is still loaded (its own teardown runs after Py_Finalize)
    for s in list(_live_sims):
        s.close()


class HideAndSeekSimulator:
    """gpu_hideseek.HideAndSeekSimulator (src/bindings.cpp:31-118)."""

    def __init__(self, exec_mode, gpu_id, num_worlds, sim_flags, rand_seed, min_hiders, max_hiders,
                 min_seekers, max_seekers, num_pbt_policies, enable_batch_renderer=False,
                 batch_render_width=64, batch_render_height=64, world_offset=0):
        L = _load()
        cfg = _HsConfig(int(exec_mode), int(gpu_id), int(num_worlds), int(sim_flags) & 0xFFFFFFFF,
                        int(rand_seed) & 0xFFFFFFFF, int(min_hiders), int(max_hiders), int(min_seekers),
                        int(max_seekers), int(num_pbt_policies), int(bool(enable_batch_renderer)),
                        int(batch_render_width), int(batch_render_height), int(world_offset))
        self._h = C.c_void_p()
        _check(L.hs_create(C.byref(cfg), C.byref(self._h)))
        self._L = L
        self.num_worlds = int(num_worlds)
        self.agents_per_world = L.hs_agents_per_world(self._h)
        _live_sims.add(self)

    def close(self):
        """Manager::~Manager (mgr.cpp:848-859).  Tensor views must not be used afterwards."""
        h = getattr(self, "_h", None)
        if h:
            self._L.hs_destroy(h)
            self._h = None

    def __del__(self):
        self.close()

    def init(self):
        _check(self._L.hs_init(self._h))

    def step(self):
        _check(self._L.hs_step(self._h))

    def step_async(self, hip_stream):
        """Enqueue one step on a caller-owned HIP stream (Manager::gpuJAXStep, mgr.cpp:1006-1022)."""
        _check(self._L.hs_step_async(self._h, C.c_void_p(int(hip_stream))))

    def _tensor(self, name):
        d = _HsTensorDesc()
        _check(self._L.hs_get_tensor(self._h, _EXPORTS[name], C.byref(d)))
        return Tensor(self, d)

    # the 21 getters of bindings.cpp:76-96
    def reset_tensor(self): return self._tensor("reset")
    def done_tensor(self): return self._tensor("done")
    def prep_counter_tensor(self): return self._tensor("prep_counter")
    def action_tensor(self): return self._tensor("action")
    def reward_tensor(self): return self._tensor("reward")
    def self_data_tensor(self): return self._tensor("self_data")
    def self_type_tensor(self): return self._tensor("self_type")
    def self_mask_tensor(self): return self._tensor("self_mask")
    def agent_data_tensor(self): return self._tensor("agent_data")
    def box_data_tensor(self): return self._tensor("box_data")
    def ramp_data_tensor(self): return self._tensor("ramp_data")
    def visible_agents_mask_tensor(self): return self._tensor("visible_agents_mask")
    def visible_boxes_mask_tensor(self): return self._tensor("visible_boxes_mask")
    def visible_ramps_mask_tensor(self): return self._tensor("visible_ramps_mask")
    def global_positions_tensor(self): return self._tensor("global_positions")
    def depth_tensor(self): return self._tensor("depth")
    def rgb_tensor(self): return self._tensor("rgb")
    def lidar_tensor(self): return self._tensor("lidar")
    def seed_tensor(self): return self._tensor("seed")
    def ckpt_ctrl_tensor(self): return self._tensor("ckpt_ctrl")
    def ckpt_tensor(self): return self._tensor("ckpt")
    # scripts/cpu_benchmark.py:77 calls a getter the reference binding lacks
    def agent_mask_tensor(self): return self._tensor("self_mask")
    # Manager::episodeResultTensor / policyAssignmentsTensor (mgr.cpp:1312-1331; JAX interface only)
    def episode_result_tensor(self): return self._tensor("episode_result")
    def policy_assignments_tensor(self): return self._tensor("policy_assignments")

    def trigger_reset(self, world_idx, level_idx):
        _check(self._L.hs_trigger_reset(self._h, int(world_idx), int(level_idx)))

    def set_action(self, agent_idx, x, y, r, g, l):
        _check(self._L.hs_set_action(self._h, int(agent_idx), int(x), int(y), int(r), int(g), int(l)))

    # ---- checkpoints: Manager::saveCheckpoint / loadCheckpoint / loadCheckpoints (mgr.cpp:905-985) ----
    def save_checkpoint(self, world_idx):
        _check(self._L.hs_save_checkpoint(self._h, int(world_idx)))

    def load_checkpoint(self, world_idx):
        _check(self._L.hs_load_checkpoint(self._h, int(world_idx)))

    def load_checkpoints(self):
        """Run the LoadCheckpoints graph for the triggers currently in ckpt_ctrl_tensor()."""
        _check(self._L.hs_load_checkpoints(self._h))

    def save_checkpoints(self):
        """Run the SaveCheckpoints graph for the triggers currently in ckpt_ctrl_tensor()."""
        _check(self._L.hs_save_checkpoints(self._h))

    # ---- the XLA-callable stream functions behind sim.jax() (bindings.cpp:97-118, mgr.cpp:362-436) ----
    # `buffers`: device pointers (ints) or objects with .data_ptr(), in the reference's order
    # (include/hideseek.h hs_jax_*).  Nothing is synchronised except stream_init.
    def _stream_call(self, fn, hip_stream, buffers):
        ptrs = [int(b.data_ptr()) if hasattr(b, "data_ptr") else int(b) for b in buffers]
        arr = (C.c_void_p * len(ptrs))(*ptrs)
        _check(fn(self._h, C.c_void_p(int(hip_stream)), arr))

    def stream_init(self, hip_stream, buffers):
        if len(buffers) != 11:
            raise ValueError("stream_init takes the 11 observation buffers")
        self._stream_call(self._L.hs_jax_init, hip_stream, buffers)

    def stream_step(self, hip_stream, buffers):
        if len(buffers) != 17:
            raise ValueError("stream_step takes actions, resets, policy_assignments, 11 observation buffers, "
                             "rewards, dones, episode_results")
        self._stream_call(self._L.hs_jax_step, hip_stream, buffers)

    def stream_save_checkpoints(self, hip_stream, buffers):
        if len(buffers) != 2:
            raise ValueError("stream_save_checkpoints takes ckpt_ctrl, ckpts")
        self._stream_call(self._L.hs_jax_save_checkpoints, hip_stream, buffers)

    def stream_load_checkpoints(self, hip_stream, buffers):
        if len(buffers) != 13:
            raise ValueError("stream_load_checkpoints takes ckpt_ctrl, ckpts, 11 observation buffers")
        self._stream_call(self._L.hs_jax_load_checkpoints, hip_stream, buffers)

    def jax(self, jax_gpu):
        raise NotImplementedError(
            "sim.jax(): the XLA custom-call registration needs jaxlib, which is not on the target (SURVEY §7 H8). "
            "The stream functions it would register exist natively (hs_jax_init/step/save_checkpoints/"
            "load_checkpoints, exposed as stream_init/stream_step/stream_save_checkpoints/stream_load_checkpoints); "
            "Tensor.to_jax() hands the buffers over through DLPack.")

    # ---- parity-test hooks (include/hideseek.h hs_debug_dump_*) ----
    def debug_bodies(self):
        import numpy as np
        b = np.zeros((self.num_worlds, 17, 13), np.float32)
        m = np.zeros((self.num_worlds, 17, 3), np.int32)
        _check(self._L.hs_debug_dump_bodies(self._h, b.ctypes.data, m.ctypes.data))
        return b, m

    def debug_walls(self):
        import numpy as np
        w = np.zeros((self.num_worlds, 36, 4), np.float32)
        info = np.zeros((self.num_worlds, 8), np.int32)
        _check(self._L.hs_debug_dump_walls(self._h, w.ctypes.data, info.ctypes.data))
        return w, info

    def set_profiling(self, enabled):
        _check(self._L.hs_set_profiling(self._h, int(bool(enabled))))

    def last_step_kernel_ms(self):
        out = (C.c_float * 3)()
        _check(self._L.hs_last_step_kernel_ms(self._h, C.byref(out)))
        return {"physics": out[0], "reset": out[1], "observe": out[2]}


class _MadronaNamespace:
    """`gpu_hideseek.madrona` submodule (bindings.cpp:21): only what scripts/ touch."""
    ExecMode = _ExecMode
    Tensor = Tensor


madrona = _MadronaNamespace()
