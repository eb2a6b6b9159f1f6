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
import weakref

from . import _native
from . import madrona
from ._native import HsConfig as _HsConfig, HsTensorDesc as _HsTensorDesc, check as _check, load as _load
from .madrona import Tensor

__all__ = ["HideAndSeekSimulator", "ShardedSimulator", "SimFlags", "madrona", "Tensor", "library_path",
           "train_interface"]


def library_path():
    return _native.LIB_PATH


class SimFlags(enum.IntFlag):
    """src/sim_flags.hpp:7-13 (nb::is_arithmetic enum, bindings.cpp:23-29)."""
    Default = 0
    UseFixedWorld = 1 << 0
    IgnoreEpisodeLength = 1 << 1
    RandomFlipTeams = 1 << 2
    ZeroAgentVelocity = 1 << 3
    # build-side extension (include/hideseek.h): skip the observation nodes
    ExtSkipObservations = 1 << 16
    # build-side extension: render the agent views (depth / rgb) in every init / step when enable_batch_renderer is set
    ExtRender = 1 << 17


# ExportID (src/sim.hpp:45-68) + renderer outputs
_EXPORTS = dict(reset=0, prep_counter=1, action=2, self_data=3, self_type=4, self_mask=5, agent_data=6,
                box_data=7, ramp_data=8, visible_agents_mask=9, visible_boxes_mask=10, visible_ramps_mask=11,
                lidar=12, seed=13, reward=14, done=15, global_positions=16, policy_assignments=17,
                episode_result=18, ckpt_ctrl=19, ckpt=20, depth=21, rgb=22)


_EXPORT_NAMES = {v: k for k, v in _EXPORTS.items()}
_ROLES = {0: "actions", 1: "resets", 2: "sim_ctrl", 3: "pbt_inputs", 4: "observations", 5: "rewards", 6: "dones",
          7: "pbt_outputs", 8: "checkpoint_data"}


def train_interface():
    """Manager::trainInterface (src/mgr.cpp:1338-1375) as the library reports it (hs_train_interface): a list of
    (name, role, getter) in the reference's order, e.g. ("self_lidar", "observations", "lidar_tensor").  The getter
    is None for the empty simCtrl tensor (mgr.cpp:1333-1336)."""
    L = _load()
    tab = C.POINTER(_native.HsIfaceEntry)()
    n = L.hs_train_interface(C.byref(tab))
    out = []
    for i in range(n):
        e = tab[i]
        getter = None if e.export_id < 0 else _EXPORT_NAMES[e.export_id] + "_tensor"
        out.append((e.name.decode(), _ROLES[e.role], getter))
    return out


_live_sims = weakref.WeakSet()
_xla_registered = False          # sim.jax() registers the four XLA custom-call targets once per process


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
        self.world_offset = int(world_offset)
        self.gpu_id = int(gpu_id)
        self.agents_per_world = L.hs_agents_per_world(self._h)
        self._tensors = {}
        _live_sims.add(self)

    def close(self):
        """Manager::~Manager (mgr.cpp:848-859).  Tensor views — and torch / jax arrays made from them — must not be
        used afterwards: they are non-owning (madrona.Tensor)."""
        h = getattr(self, "_h", None)
        if h:
            self._tensors = {}
            self._L.hs_destroy(h)
            self._h = None

    def __del__(self):
        self.close()

    def init(self):
        _check(self._L.hs_init(self._h))

    def step(self):
        _check(self._L.hs_step(self._h))

    def render(self):
        """Render every agent's view of the current state into depth_tensor() / rgb_tensor() (hs_render;
        Manager::step's batchRender(), src/mgr.cpp:894-901).  Per step instead: SimFlags.ExtRender."""
        _check(self._L.hs_render(self._h))

    def step_begin(self):
        """Enqueue one step on this handle's own stream and return (hs_step_begin); pair with step_end()."""
        _check(self._L.hs_step_begin(self._h))

    def step_end(self):
        """Wait for the step started by step_begin() and report device-side failures (hs_step_end)."""
        _check(self._L.hs_step_end(self._h))

    def step_async(self, hip_stream):
        """Enqueue one step on a caller-owned HIP stream (Manager::gpuJAXStep, mgr.cpp:1006-1022)."""
        _check(self._L.hs_step_async(self._h, C.c_void_p(int(hip_stream))))

    def _tensor(self, name):
        t = self._tensors.get(name)
        if t is None:
            d = _HsTensorDesc()
            _check(self._L.hs_get_tensor(self._h, _EXPORTS[name], C.byref(d)))
            t = self._tensors[name] = Tensor(weakref.proxy(self), d)
        return t

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

    def train_interface(self):
        """Manager::trainInterface (mgr.cpp:1338-1375) bound to this simulator: {role: {name: Tensor}} with the
        reference's names and order — what `sim.jax()` hands to the learner."""
        out = {}
        for name, role, getter in train_interface():
            out.setdefault(role, {})[name] = getattr(self, getter)() if getter else None
        return out

    # ---- XLA custom calls (bindings.cpp:97-118: madrona::py::JAXInterface::buildEntry) ----
    def xla_opaque(self):
        """The descriptor an XLA custom call of this simulator carries: the 8 bytes of the native handle."""
        return int(self._h.value).to_bytes(8, "little")

    def xla_custom_call_targets(self):
        """{'init' | 'step' | 'save_ckpts' | 'load_ckpts': PyCapsule("xla._CUSTOM_CALL_TARGET")} around the native
        hs_xla_* functions (include/hideseek.h) — what xla_client.register_custom_call_target(name, capsule,
        platform="ROCM") takes."""
        new = C.pythonapi.PyCapsule_New
        new.restype = C.py_object
        new.argtypes = [C.c_void_p, C.c_char_p, C.c_void_p]
        return {k: new(C.cast(getattr(self._L, sym), C.c_void_p).value, b"xla._CUSTOM_CALL_TARGET", None)
                for k, sym in _native.XLA_TARGETS.items()}

    def xla_call_signatures(self):
        """Operand and result lists (name, shape, dtype) of the four custom calls, in buffer order
        (mgr.cpp:168-201, 362-436) — what a lowering rule needs besides the target and the opaque."""
        iface = self.train_interface()

        def sig(t):
            return (tuple(t.shape), t.dtype)
        obs = [(n, *sig(t)) for n, t in iface["observations"].items()]
        act = [("actions", *sig(self.action_tensor())), ("resets", *sig(self.reset_tensor())),
               ("policy_assignments", *sig(self.policy_assignments_tensor()))]
        out = [("rewards", *sig(self.reward_tensor())), ("dones", *sig(self.done_tensor())),
               ("episode_results", *sig(self.episode_result_tensor()))]
        ck = [("ckpt_ctrl", *sig(self.ckpt_ctrl_tensor())), ("ckpts", *sig(self.ckpt_tensor()))]
        return {"init": {"operands": [], "results": obs},
                "step": {"operands": act, "results": obs + out},
                "save_ckpts": {"operands": ck[:1], "results": ck[1:]},
                "load_ckpts": {"operands": ck, "results": obs}}

    def jax(self, jax_gpu):
        """bindings.cpp:97-118 (JAXInterface::buildEntry).  Everything XLA needs exists natively: the interface table
        (hs_train_interface), the four custom-call targets in XLA's own ABI (hs_xla_init / step / save_checkpoints /
        load_checkpoints — xla_custom_call_targets()), their descriptor (xla_opaque()) and their operand / result
        lists (xla_call_signatures()).  With jaxlib importable the targets are registered for the ROCm platform and
        the bundle is returned; without it (this image and the target: SURVEY §7 H8) the same bundle travels on the
        NotImplementedError, because the JAX-side lowering rules cannot be exercised here."""
        if not jax_gpu:
            raise NotImplementedError("sim.jax(jax_gpu=False): there is no CPU execution path (DESIGN.md §1)")
        bundle = {"train_interface": self.train_interface(), "targets": self.xla_custom_call_targets(),
                  "opaque": self.xla_opaque(), "signatures": self.xla_call_signatures(),
                  # one set of names per process: the targets are the same native functions for every simulator,
                  # the opaque descriptor says which one is meant
                  "target_names": {k: f"gpu_hideseek_{k}" for k in _native.XLA_TARGETS}}
        try:
            from jax.lib import xla_client
        except ImportError:
            err = NotImplementedError(
                "sim.jax(): jaxlib is not importable (SURVEY §7 H8), so the XLA custom-call targets cannot be "
                "registered here. They exist natively (hs_xla_*); the bundle a registration needs — capsules, "
                "opaque descriptor, operand / result signatures, interface table — is attached as .xla and "
                ".train_interface; stream_init / stream_step / ... call the same functions directly, and "
                "Tensor.to_jax() hands buffers over through DLPack.")
            err.train_interface = bundle["train_interface"]
            err.xla = bundle
            raise err
        global _xla_registered
        if not _xla_registered:
            for k, cap in bundle["targets"].items():
                xla_client.register_custom_call_target(bundle["target_names"][k], cap, platform="ROCM")
            _xla_registered = True
        return bundle

    def device_status(self):
        """hs_get_device_status: sticky device-side counters — candidate pairs that took the spill path of the physics
        kernel (beyond its LDS capacities; results unaffected), dropped pairs (always 0) — and whether graphs are in use."""
        st = _native.HsDeviceStatus()
        _check(self._L.hs_get_device_status(self._h, C.byref(st)))
        return {"dropped_dd_pairs": int(st.dropped_dd_pairs), "dropped_static_pairs": int(st.dropped_static_pairs),
                "dropped_candidate_pairs": int(st.dropped_dd_pairs + st.dropped_static_pairs),
                "spilled_dd_pairs": int(st.spilled_dd_pairs), "spilled_static_pairs": int(st.spilled_static_pairs),
                "spilled_candidate_pairs": int(st.spilled_dd_pairs + st.spilled_static_pairs),
                "graphs_in_use": bool(st.graphs_in_use)}

    def warning(self):
        """The library's last message for this thread."""
        return self._L.hs_last_error().decode()

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


from .sharded import ShardedSimulator  # noqa: E402
