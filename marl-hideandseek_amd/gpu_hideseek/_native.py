"""ctypes binding of libhideseek.so (include/hideseek.h).  Internal: imported by the package's public modules."""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HS_LIB_PATH") or os.path.join(os.path.dirname(_HERE), "lib", "libhideseek.so")


class HsConfig(C.Structure):
    _fields_ = [("exec_mode", C.c_int32), ("gpu_id", C.c_int32), ("num_worlds", C.c_int32),
                ("sim_flags", C.c_uint32), ("rand_seed", C.c_uint32),
                ("min_hiders", C.c_int32), ("max_hiders", C.c_int32),
                ("min_seekers", C.c_int32), ("max_seekers", C.c_int32),
                ("num_pbt_policies", C.c_int32), ("enable_batch_renderer", C.c_int32),
                ("batch_render_width", C.c_int32), ("batch_render_height", C.c_int32),
                ("world_offset", C.c_int32)]


class HsTensorDesc(C.Structure):
    _fields_ = [("ptr", C.c_void_p), ("dtype", C.c_int32), ("ndim", C.c_int32),
                ("dims", C.c_int64 * 4), ("gpu_id", C.c_int32)]


class HsIfaceEntry(C.Structure):
    _fields_ = [("name", C.c_char_p), ("role", C.c_int32), ("export_id", C.c_int32)]


class HsDeviceStatus(C.Structure):
    _fields_ = [("dropped_dd_pairs", C.c_int64), ("dropped_static_pairs", C.c_int64),
                ("graphs_in_use", C.c_int32), ("reserved", C.c_int32),
                ("spilled_dd_pairs", C.c_int64), ("spilled_static_pairs", C.c_int64)]


# XLA custom-call targets (include/hideseek.h hs_xla_*): the key `sim.jax()` files each one under -> native symbol
XLA_TARGETS = {"init": "hs_xla_init", "step": "hs_xla_step", "save_ckpts": "hs_xla_save_checkpoints",
               "load_ckpts": "hs_xla_load_checkpoints"}

_lib = None


def load():
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
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python __graft_entry__.py build` "
            "(hipcc --offload-arch=gfx950). gpu_hideseek has no CPU fallback.")
    L = C.CDLL(LIB_PATH)
    L.hs_create.argtypes = [C.POINTER(HsConfig), C.POINTER(C.c_void_p)]
    L.hs_create.restype = C.c_int32
    L.hs_destroy.argtypes = [C.c_void_p]
    L.hs_destroy.restype = None
    for n in ("hs_init", "hs_step", "hs_step_begin", "hs_step_end", "hs_save_checkpoints", "hs_load_checkpoints", "hs_render"):
        getattr(L, n).argtypes = [C.c_void_p]
        getattr(L, n).restype = C.c_int32
    for n in ("hs_save_checkpoint", "hs_load_checkpoint"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_int32]
        getattr(L, n).restype = C.c_int32
    for n in ("hs_jax_init", "hs_jax_step", "hs_jax_save_checkpoints", "hs_jax_load_checkpoints"):
        getattr(L, n).argtypes = [C.c_void_p, C.c_void_p, C.POINTER(C.c_void_p)]
        getattr(L, n).restype = C.c_int32
    for n in XLA_TARGETS.values():
        getattr(L, n).argtypes = [C.c_void_p, C.POINTER(C.c_void_p), C.c_char_p, C.c_size_t]
        getattr(L, n).restype = None
    L.hs_xla_last_status.argtypes = [C.c_int32]
    L.hs_xla_last_status.restype = C.c_int32
    L.hs_step_async.argtypes = [C.c_void_p, C.c_void_p]
    L.hs_step_async.restype = C.c_int32
    L.hs_get_tensor.argtypes = [C.c_void_p, C.c_int32, C.POINTER(HsTensorDesc)]
    L.hs_get_tensor.restype = C.c_int32
    L.hs_trigger_reset.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
    L.hs_trigger_reset.restype = C.c_int32
    L.hs_set_action.argtypes = [C.c_void_p] + [C.c_int32] * 6
    L.hs_set_action.restype = C.c_int32
    L.hs_agents_per_world.argtypes = [C.c_void_p]
    L.hs_agents_per_world.restype = C.c_int32
    L.hs_train_interface.argtypes = [C.POINTER(C.POINTER(HsIfaceEntry))]
    L.hs_train_interface.restype = C.c_int32
    L.hs_get_device_status.argtypes = [C.c_void_p, C.POINTER(HsDeviceStatus)]
    L.hs_get_device_status.restype = C.c_int32
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


def check(rc):
    if rc != 0:
        msg = load().hs_last_error().decode()
        if rc == 1:
            raise ValueError(msg)
        if rc == 3:
            raise NotImplementedError(msg)
        raise RuntimeError(f"libhideseek error {rc}: {msg}")
