"""`gpu_hideseek.madrona` — the submodule the reference's binding imports from the engine's Python package
(src/bindings.cpp:21 `nb::module_::import_("madrona")`) and its scripts import from
(scripts/jax_train.py:16, scripts/jax_infer.py:13: `from gpu_hideseek.madrona import ExecMode`;
scripts/benchmark.py:22: `gpu_hideseek.madrona.ExecMode.CUDA`).  Only what scripts/ touch: `ExecMode`, `Tensor`.
"""
import ctypes as C
import enum

from . import _native

__all__ = ["ExecMode", "Tensor"]


class ExecMode(enum.IntEnum):
    """madrona::ExecMode (Manager::Config::execMode, src/mgr.hpp:17)."""
    CPU = 0
    CUDA = 1   # the reference's name for "GPU"; here it means HIP on gfx950


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

# One DLManagedTensor record per exported buffer for the life of the process: a consumer (torch) may run the
# record's deleter while the interpreter is shutting down, so the deleter is a C no-op in libhideseek and the
# records are never freed.  Keyed by (device, address, shape, dtype), so repeated `*_tensor().to_torch()` calls
# — scripts do that every step — reuse one record instead of growing this table.
_records = {}


class Tensor:
    """Counterpart of madrona::py::Tensor (src/mgr.cpp:824-842): a NON-OWNING view of simulator memory.

    Lifetime: the simulator owns the memory (src/mgr.hpp ownership convention) and neither a Tensor nor a torch / jax
    array made from it keeps the simulator alive: `del sim` (scripts/benchmark.py:94) frees the HBM at once, as
    `Manager::~Manager` does.  A Tensor knows its owner only weakly and refuses to hand out new views once the owner
    is closed or gone (`to_torch`, `to_jax`, `__dlpack__` raise); arrays made EARLIER are the caller's to drop."""

    def __init__(self, owner, desc):
        self._owner = owner            # weakref.proxy of the simulator: never extends its life
        self.ptr = desc.ptr
        self.dtype_id = desc.dtype
        self.shape = tuple(int(desc.dims[i]) for i in range(desc.ndim))
        self.gpu_id = desc.gpu_id
        self._torch = None

    @property
    def dtype(self):
        return _DTYPES[self.dtype_id][2]

    def _require_owner(self):
        try:
            if self._owner._h:
                return
        except ReferenceError:
            pass
        raise RuntimeError("the simulator that owns this tensor has been closed or deleted: its device memory is freed")

    def __dlpack_device__(self):
        return (_kDLROCM, self.gpu_id)

    def _record(self):
        key = (self.gpu_id, self.ptr, self.shape, self.dtype_id)
        rec = _records.get(key)
        if rec is None:
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
            mt.deleter = C.cast(_native.load().hs_dlpack_noop_deleter, _DLDeleter)
            rec = _records[key] = (mt, shape)
        return rec[0]

    def __dlpack__(self, stream=None, **_):
        self._require_owner()
        new_capsule = C.pythonapi.PyCapsule_New
        new_capsule.restype = C.py_object
        new_capsule.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        return new_capsule(C.addressof(self._record()), b"dltensor", None)

    def to_torch(self):
        """Zero-copy torch view on cuda:gpu_id (madrona Tensor.to_torch, scripts/benchmark.py:38-39,47)."""
        self._require_owner()
        if self._torch is None:
            import torch
            self._torch = torch.from_dlpack(self)
        return self._torch

    def to_jax(self):
        self._require_owner()
        import jax.dlpack
        return jax.dlpack.from_dlpack(self)

    def __repr__(self):
        return f"Tensor(shape={self.shape}, dtype={self.dtype}, gpu={self.gpu_id}, ptr=0x{self.ptr or 0:x})"
