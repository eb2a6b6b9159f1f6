"""ShardedSimulator — one front-end over the GPUs of a node (BASELINE.json north_star: "the batch
shards across the 8 GPUs of one node by world index with no RCCL on the step path (only a host-side concat of
observation tensors)").

The reference is single-GPU (`Manager::Config::gpuID`, src/mgr.hpp:18), so this has no counterpart there; it keeps
the HideAndSeekSimulator call shape (src/bindings.cpp:32-96).  Shard g owns the contiguous global worlds
[start_g, start_g + n_g) on its own device with its own HIP stream and is created with `world_offset = start_g`,
so every world draws from the RNG stream of its GLOBAL index (src/sim.cpp:107-113) and the sharded run equals the
monolithic one world for world.  `step()` starts every shard (hs_step_begin) before it waits for any
(hs_step_end): the devices run concurrently from one host thread, and there is no collective.

Tensors: `sim.action_tensor()` etc. return a ShardedTensor — `.shards` are the per-device zero-copy views,
`.gather()` concatenates them into ONE pinned host tensor (rows in global world order), `.scatter(t)` writes a
host tensor of that shape back to the devices (actions, resets).
"""
from . import madrona

_GETTERS = ["reset", "done", "prep_counter", "action", "reward", "self_data", "self_type", "self_mask", "agent_data",
            "box_data", "ramp_data", "visible_agents_mask", "visible_boxes_mask", "visible_ramps_mask",
            "global_positions", "depth", "rgb", "lidar", "seed", "ckpt_ctrl", "ckpt", "agent_mask", "episode_result",
            "policy_assignments"]


def shard_ranges(num_worlds, num_shards):
    """Contiguous world ranges [(start, count)] — the first `num_worlds % num_shards` shards take one world more."""
    if num_shards <= 0 or num_worlds < num_shards:
        raise ValueError("need at least one world per shard")
    base, extra = divmod(int(num_worlds), int(num_shards))
    out, start = [], 0
    for g in range(num_shards):
        n = base + (1 if g < extra else 0)
        out.append((start, n))
        start += n
    return out


def locate(ranges, world):
    """Global world index -> (shard, local world index)."""
    for g, (start, n) in enumerate(ranges):
        if start <= world < start + n:
            return g, world - start
    raise ValueError("world index out of range")


class ShardedTensor:
    def __init__(self, shards, rows_per_world, ranges):
        self.shards = shards                  # madrona.Tensor per shard, in world order
        self.rows_per_world = rows_per_world
        self.ranges = ranges
        self._host = None

    @property
    def shape(self):
        return (sum(t.shape[0] for t in self.shards),) + tuple(self.shards[0].shape[1:])

    def per_device(self):
        """The zero-copy device views (torch), one per shard."""
        return [t.to_torch() for t in self.shards]

    def row_range(self, shard):
        start, n = self.ranges[shard]
        return start * self.rows_per_world, (start + n) * self.rows_per_world

    def gather(self, out=None):
        """Host-side concat: device -> pinned host copies of every shard, rows in global world order."""
        import torch
        if out is None:
            if self._host is None:
                first = self.shards[0].to_torch()
                self._host = torch.empty(self.shape, dtype=first.dtype, pin_memory=True)
            out = self._host
        devs = self.per_device()
        for g, d in enumerate(devs):
            lo, hi = self.row_range(g)
            out[lo:hi].copy_(d, non_blocking=True)
        for d in devs:
            torch.cuda.synchronize(d.device)
        return out

    def scatter(self, host):
        """Write a host tensor of the gathered shape to the shards (actions / resets)."""
        import torch
        devs = self.per_device()
        for g, d in enumerate(devs):
            lo, hi = self.row_range(g)
            d.copy_(host[lo:hi], non_blocking=True)
        for d in devs:
            torch.cuda.synchronize(d.device)

    def to_torch(self):
        """A single handle aliases nothing across devices: one shard -> its zero-copy view, otherwise the gather."""
        return self.shards[0].to_torch() if len(self.shards) == 1 else self.gather()


class ShardedSimulator:
    def __init__(self, gpu_ids, num_worlds, **kw):
        from . import HideAndSeekSimulator
        kw.pop("gpu_id", None)
        kw.pop("world_offset", None)
        kw.setdefault("exec_mode", madrona.ExecMode.CUDA)
        self.gpu_ids = list(gpu_ids)
        self.num_worlds = int(num_worlds)
        self.ranges = shard_ranges(self.num_worlds, len(self.gpu_ids))
        self.shards = [HideAndSeekSimulator(gpu_id=g, num_worlds=n, world_offset=start, **kw)
                       for g, (start, n) in zip(self.gpu_ids, self.ranges)]
        self.agents_per_world = self.shards[0].agents_per_world

    def init(self):
        for s in self.shards:
            s.init()

    def step(self):
        started = []
        try:
            for s in self.shards:          # every device starts its step ...
                s.step_begin()
                started.append(s)
        finally:
            err = None
            for s in started:              # ... before the host waits for any of them
                try:
                    s.step_end()
                except Exception as e:     # noqa: BLE001 - finish the other shards, then report the first failure
                    err = err or e
            if err is not None:
                raise err

    def _tensor(self, name):
        ts = [getattr(s, name + "_tensor")() for s in self.shards]
        n0 = self.ranges[0][1]
        return ShardedTensor(ts, ts[0].shape[0] // n0, self.ranges)

    def trigger_reset(self, world_idx, level_idx):
        g, local = locate(self.ranges, int(world_idx))
        self.shards[g].trigger_reset(local, level_idx)

    def set_action(self, agent_idx, x, y, r, g, l):
        A = self.agents_per_world
        shard, local = locate(self.ranges, int(agent_idx) // A)
        self.shards[shard].set_action(local * A + int(agent_idx) % A, x, y, r, g, l)

    def device_status(self):
        out = {}
        for s in self.shards:
            for k, v in s.device_status().items():
                out[k] = (out.get(k, 0) + v) if not isinstance(v, bool) else (out.get(k, True) and v)
        return out

    def close(self):
        for s in self.shards:
            s.close()


def _make_getter(name):
    def getter(self):
        return self._tensor(name)
    getter.__name__ = name + "_tensor"
    return getter


for _n in _GETTERS:
    setattr(ShardedSimulator, _n + "_tensor", _make_getter(_n))
