"""Replay logs: the `[steps][num_worlds]` raw `Checkpoint` records that the reference's scripts/jax_infer.py
appends per step (`np.asarray(ckpts).tofile(...)`, :125) and src/viewer.cpp plays back by copying one step's
records into the checkpoint tensor, setting every trigger and calling `loadCheckpoints()` (:13-26, 185-215).
Same file format here (record = include/hideseek.h `hs_checkpoint`, 1392 bytes)."""
import numpy as np

CHECKPOINT_BYTES = 1392


def record_step(sim, fileobj):
    """Save a checkpoint of every world and append the `[num_worlds]` records to an open binary file."""
    ctrl = sim.ckpt_ctrl_tensor().to_torch()
    ctrl.view(ctrl.dtype).copy_(_ones_like_i32(ctrl))
    sim.save_checkpoints()
    sim.ckpt_tensor().to_torch().cpu().numpy().tofile(fileobj)


def read_log(path, num_worlds):
    """-> uint8 array [steps, num_worlds, 1392] (trailing partial step dropped, as viewer.cpp:13-26 does)."""
    raw = np.fromfile(path, dtype=np.uint8)
    per_step = num_worlds * CHECKPOINT_BYTES
    steps = raw.size // per_step
    return raw[:steps * per_step].reshape(steps, num_worlds, CHECKPOINT_BYTES)


def replay_step(sim, log, step):
    """Restore every world to recorded step `step` (viewer.cpp:185-215): observations are recomputed."""
    import torch
    ck = sim.ckpt_tensor().to_torch()
    ck.copy_(torch.from_numpy(np.ascontiguousarray(log[step])).to(ck.device))
    ctrl = sim.ckpt_ctrl_tensor().to_torch()
    ctrl.copy_(_ones_like_i32(ctrl))
    sim.load_checkpoints()


def _ones_like_i32(ctrl_u8):
    """CheckpointControl::trigger = 1 for every world, as the [N, 4] uint8 view the tensor is exported as."""
    import torch
    ones = torch.ones(ctrl_u8.shape[0], 1, dtype=torch.int32, device=ctrl_u8.device)
    return ones.view(torch.uint8).reshape(ctrl_u8.shape)
