"""Oracle checkpoint save/load (oracle/hs_ref_ckpt.hpp; Checkpoint src/sim.hpp:283-313, systems
src/sim.cpp:956-1137): layout known answers and the restore property, on CPU."""
import ctypes as C
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class _Obj(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 4), ("lin", C.c_float * 3), ("ang", C.c_float * 3),
                ("team", C.c_uint32), ("is_locked", C.c_uint8), ("_pad", C.c_uint8 * 3)]


class _Agent(C.Structure):
    _fields_ = [("pos", C.c_float * 3), ("rot", C.c_float * 4), ("lin", C.c_float * 3), ("ang", C.c_float * 3),
                ("grab_idx", C.c_int32), ("grab_r1", C.c_float * 3), ("grab_r2", C.c_float * 3),
                ("attach_rot1", C.c_float * 4), ("attach_rot2", C.c_float * 4), ("separation", C.c_float)]


class Checkpoint(C.Structure):                     # include/hideseek.h hs_checkpoint
    _fields_ = [("episode_key", C.c_uint32 * 2), ("running_scores", C.c_int32 * 2), ("episode_step", C.c_int32),
                ("agents", _Agent * 6), ("boxes", _Obj * 9), ("ramps", _Obj * 2),
                ("num_hiders", C.c_int32), ("num_seekers", C.c_int32), ("num_boxes", C.c_int32), ("num_ramps", C.c_int32)]


def test_layout_is_1392_bytes():
    """ckpt tensor is [N, sizeof(Checkpoint) ~ 1392]; the C header must agree with the oracle's struct."""
    assert C.sizeof(Checkpoint) == 1392 and C.sizeof(_Agent) == 116 and C.sizeof(_Obj) == 60
    hdr = open(os.path.join(ROOT, "include", "hideseek.h")).read()
    for field in ("episode_key[2]", "running_scores[2]", "episode_step", "agents[6]", "boxes[9]", "ramps[2]",
                  "num_hiders, num_seekers, num_boxes, num_ramps"):
        assert field in hdr


def _run(ref, rng, steps, full=True):
    rows = ref.N * ref.A
    for _ in range(steps):
        a = ref.tensor("action")
        a[:, :3] = rng.integers(0, 11, (rows, 3))
        a[:, 3:] = rng.integers(0, 2, (rows, 2)) if full else 0
        ref.step()


def test_saved_record_fields(oracle):
    ref = oracle.RefSim(6, sim_flags=13, rand_seed=5, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3,
                        world_offset=100, threads=2)
    ref.init()
    _run(ref, np.random.default_rng(0), 50)
    ref.tensor("ckpt_ctrl")[[1, 4]] = 1
    ref.save_checkpoints()
    assert not ref.tensor("ckpt_ctrl").any()                                     # sim.cpp:1052
    raw = ref.tensor("ckpt")
    assert not raw[[0, 2, 3, 5]].any()                                           # untriggered worlds untouched
    bodies, meta = ref.bodies()
    _, info = ref.walls()
    for w in (1, 4):
        ck = Checkpoint.from_buffer_copy(raw[w].tobytes())
        assert list(ck.episode_key) == [0, 100 + w]                              # {episode idx, global world id} sim.cpp:107-110
        assert ck.episode_step == 50 and ck.num_hiders == 3 and ck.num_seekers == 3
        assert ck.num_boxes == info[w, 2] and ck.num_ramps == info[w, 3] == 2
        for i in range(ck.num_boxes):
            assert np.array_equal(np.array(ck.boxes[i].pos, np.float32), bodies[w, i, 0:3])
            assert ck.boxes[i].team == meta[w, i, 2] and ck.boxes[i].is_locked == (meta[w, i, 1] == 2)
        for i in range(2):
            assert np.array_equal(np.array(ck.ramps[i].rot, np.float32), bodies[w, 9 + i, 3:7])
        for i in range(6):
            a = ck.agents[i]
            assert -1 <= a.grab_idx < ck.num_boxes + ck.num_ramps
            if a.grab_idx >= 0:
                assert list(a.grab_r1) == [0.0, 1.25, 0.5] and list(a.attach_rot1) == [1.0, 0.0, 0.0, 0.0]
            else:
                assert not any(a.grab_r1) and not any(a.grab_r2) and a.separation == 0.0   # sim.cpp:1071-1074
    ref.close()


def test_restore_then_replay_is_identical(oracle):
    ref = oracle.RefSim(12, sim_flags=0, rand_seed=9, min_hiders=1, max_hiders=3, min_seekers=1, max_seekers=3, threads=2)
    ref.init()
    _run(ref, np.random.default_rng(1), 40)
    ref.tensor("ckpt_ctrl")[:] = 1
    ref.save_checkpoints()
    names = [k for k in oracle.TENSORS if k not in ("ckpt", "ckpt_ctrl", "action")]
    at_save = {k: ref.tensor(k).copy() for k in names}
    b_save = ref.bodies()
    _run(ref, np.random.default_rng(2), 25)
    b_end = ref.bodies()
    end = {k: ref.tensor(k).copy() for k in names}
    ref.tensor("ckpt_ctrl")[:] = 1
    ref.load_checkpoints()
    assert (ref.tensor("ckpt_ctrl") == 1).all()                                  # sim.cpp:963 leaves the trigger set
    assert (ref.tensor("action")[ref.tensor("self_mask")[:, 0] == 1] == [2, 2, 2, 0, 0]).all()   # makeAgent level_gen.cpp:26-32
    for k in names:
        if k in ("reward", "done", "episode_result"):
            continue                                                             # not part of the load graph
        assert np.array_equal(ref.tensor(k).view(np.int32), at_save[k].view(np.int32)), k
    assert np.array_equal(ref.bodies()[0].view(np.int32), b_save[0].view(np.int32))
    assert np.array_equal(ref.bodies()[1], b_save[1])
    ref.tensor("ckpt_ctrl")[:] = 0
    _run(ref, np.random.default_rng(2), 25)
    assert np.array_equal(ref.bodies()[0].view(np.int32), b_end[0].view(np.int32))
    for k in names:
        assert np.array_equal(ref.tensor(k).view(np.int32), end[k].view(np.int32)), k
    ref.close()


def test_load_tolerates_a_garbage_record(oracle):
    ref = oracle.RefSim(2, threads=1)
    ref.init()
    ref.tensor("ckpt")[:] = 0xFF
    ref.tensor("ckpt_ctrl")[0] = 1
    ref.load_checkpoints()
    _, info = ref.walls()
    assert info[0, 4] == 0 and info[0, 5] == 0 and info[1, 4] == 2              # counts clamped; world 1 untouched
    ref.close()
