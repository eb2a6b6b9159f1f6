"""N>1 path on CPU: two ranks (gloo) each own a contiguous world range with its global
world_offset; there is no collective on the step path, only a gather of per-shard checksums —
the same structure bench.py uses over RCCL.  Shard results must equal the single-process run."""
import hashlib
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _digest(sim):
    h = hashlib.sha256()
    b, m = sim.bodies()
    for a in (b, m, sim.tensor("self_data"), sim.tensor("lidar"), sim.tensor("reward"), sim.tensor("seed")):
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _run_shard(rank, world_size, port, n_total, steps, ret):
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import torch.distributed as dist
    import hs_ref
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world_size)
    n = n_total // world_size
    sim = hs_ref.RefSim(n, rand_seed=11, world_offset=rank * n)
    sim.init()
    for s in range(steps):
        rows = n * sim.A
        g = np.arange(rank * rows, (rank + 1) * rows)          # actions keyed by GLOBAL agent row
        a = sim.tensor("action")
        a[:, 0] = (g * 7 + s) % 11
        a[:, 1] = (g * 3 + 2 * s) % 11
        sim.step()
    dist.barrier()
    digests = [None] * world_size
    dist.all_gather_object(digests, (_digest(sim), sim.bodies()[0].tobytes()))
    if rank == 0:
        ret["digests"] = digests
    dist.destroy_process_group()


def test_two_rank_shards_equal_single_process(oracle):
    import torch.multiprocessing as mp
    n_total, steps = 16, 12
    port = _free_port()
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_run_shard, args=(2, port, n_total, steps, ret), nprocs=2, join=True)
    full = oracle.RefSim(n_total, rand_seed=11)
    full.init()
    for s in range(steps):
        rows = n_total * full.A
        g = np.arange(rows)
        a = full.tensor("action")
        a[:, 0] = (g * 7 + s) % 11
        a[:, 1] = (g * 3 + 2 * s) % 11
        full.step()
    fb = full.bodies()[0]
    half = n_total // 2
    d = ret["digests"]
    assert d[0][1] == fb[:half].tobytes() and d[1][1] == fb[half:].tobytes()
    assert d[0][0] != d[1][0]
