/*
 * hideseek.h — C ABI of the MI355X-native batch hide-and-seek simulator (libhideseek.so).
 *
 * This is the drop-in boundary for the reference's `Manager` class (src/mgr.hpp:14-99): one opaque
 * simulator handle per shard of worlds, `init` / `step`, and non-owning descriptors of the
 * exported columns that the reference hands to Python as `madrona::py::Tensor`
 * (src/mgr.cpp:824-842, 1062-1336).  No C++ or torch types cross this boundary: plain pointers,
 * sizes and status codes.  Errors are returned as status codes instead of the reference's
 * FATAL()/abort (src/mgr.cpp:466,573,762); `hs_last_error()` gives the message.
 *
 * All tensors live in device (HBM) memory of `gpu_id`, are row-major contiguous and stay valid
 * and at the same address until `hs_destroy` (src/mgr.hpp ownership convention; the scripts write
 * `action` and `reset` in place: scripts/benchmark.py:64-65,82-84).
 */
#ifndef HIDESEEK_H
#define HIDESEEK_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct hs_sim hs_sim;   /* replaces Manager::Impl (src/mgr.hpp:94-98) */

/* Status codes (the reference aborts instead). */
enum {
    HS_OK = 0,
    HS_ERR_INVALID_ARG = 1,
    HS_ERR_NO_DEVICE = 2,      /* no HIP device / HIP runtime failure */
    HS_ERR_UNSUPPORTED = 3,    /* e.g. exec_mode CPU: this library has no CPU execution path */
    HS_ERR_HIP = 4
};

/* madrona::ExecMode as used by Manager::Config::execMode (src/mgr.hpp:17). */
enum { HS_EXEC_CPU = 0, HS_EXEC_GPU = 1 /* "CUDA" in the reference's Python enum */ };

/* SimFlags (src/sim_flags.hpp:7-13). Bits 16+ are build-side extensions. */
enum {
    HS_FLAG_DEFAULT = 0,
    HS_FLAG_USE_FIXED_WORLD = 1 << 0,
    HS_FLAG_IGNORE_EPISODE_LENGTH = 1 << 1,
    HS_FLAG_RANDOM_FLIP_TEAMS = 1 << 2,
    HS_FLAG_ZERO_AGENT_VELOCITY = 1 << 3,
    /* extension: skip the observation task-graph nodes (sim.cpp:1232-1293) — the physics-only
       roofline configuration of BASELINE.json configs[2]; not a reference flag. */
    HS_FLAG_EXT_SKIP_OBSERVATIONS = 1 << 16,
    /* extension: render the agent views (depth + RGB, hs_render below) as part of every init / step when
       enable_batch_renderer is set.  Without it the renderer outputs stay allocated and unwritten, which is what the
       reference scripts' arguments get (scripts/benchmark.py:32,47 only takes the tensor). */
    HS_FLAG_EXT_RENDER = 1 << 17
};

/* Manager::Config (src/mgr.hpp:16-32) + the shard placement of SURVEY §8e. */
typedef struct hs_config {
    int32_t exec_mode;            /* HS_EXEC_GPU */
    int32_t gpu_id;               /* Manager::Config::gpuID */
    int32_t num_worlds;           /* worlds simulated by THIS handle */
    uint32_t sim_flags;
    uint32_t rand_seed;
    int32_t min_hiders, max_hiders, min_seekers, max_seekers;
    int32_t num_pbt_policies;
    int32_t enable_batch_renderer;   /* rgb/depth tensors; rendered by hs_render / under HS_FLAG_EXT_RENDER */
    int32_t batch_render_width, batch_render_height;
    int32_t world_offset;         /* global index of local world 0 (RNG keys use global ids) */
} hs_config;

/* ExportID (src/sim.hpp:45-68), same numbering, plus the two renderer outputs
 * (src/mgr.cpp:1241-1263) and debug dumps of internal state used by the parity tests. */
enum {
    HS_EXPORT_RESET = 0,
    HS_EXPORT_PREP_COUNTER = 1,
    HS_EXPORT_ACTION = 2,
    HS_EXPORT_SELF_OBS = 3,
    HS_EXPORT_SELF_TYPE = 4,
    HS_EXPORT_SELF_MASK = 5,
    HS_EXPORT_AGENT_OBS = 6,
    HS_EXPORT_BOX_OBS = 7,
    HS_EXPORT_RAMP_OBS = 8,
    HS_EXPORT_AGENT_VIS_MASKS = 9,
    HS_EXPORT_BOX_VIS_MASKS = 10,
    HS_EXPORT_RAMP_VIS_MASKS = 11,
    HS_EXPORT_LIDAR = 12,
    HS_EXPORT_SEED = 13,
    HS_EXPORT_REWARD = 14,
    HS_EXPORT_DONE = 15,
    HS_EXPORT_GLOBAL_DEBUG_POSITIONS = 16,
    HS_EXPORT_AGENT_POLICY = 17,
    HS_EXPORT_EPISODE_RESULT = 18,
    HS_EXPORT_CHECKPOINT_CONTROL = 19,
    HS_EXPORT_CHECKPOINT = 20,
    HS_EXPORT_DEPTH = 21,
    HS_EXPORT_RGB = 22,
    HS_NUM_EXPORTS = 23
};

enum { HS_DTYPE_I32 = 0, HS_DTYPE_F32 = 1, HS_DTYPE_U8 = 2 };

/* madrona::py::Tensor (src/mgr.cpp:824-842): pointer, element type, dimensions, device. */
typedef struct hs_tensor_desc {
    void *ptr;
    int32_t dtype;
    int32_t ndim;
    int64_t dims[4];
    int32_t gpu_id;
} hs_tensor_desc;

/* Checkpoint (src/sim.hpp:283-313): the per-world snapshot behind `ckpt_tensor()` ([N, 1392] u8) and the
 * replay log of scripts/jax_infer.py:125.  Field order follows the reference struct; Quat is w,x,y,z,
 * Velocity is linear then angular.  The engine's JointConstraint::Fixed is not in the reference tree:
 * {attachRot1, attachRot2, separation} is assumed, which reproduces the 1392-byte record. */
typedef struct hs_ckpt_object {          /* Checkpoint::DynObjectState (sim.hpp:294-297) */
    float pos[3], rot[4], lin[3], ang[3];
    uint32_t team;                        /* OwnerTeam: 0 None, 1 Seeker, 2 Hider, 3 Unownable (sim.hpp:127-132) */
    uint8_t is_locked, _pad[3];
} hs_ckpt_object;
typedef struct hs_ckpt_agent {           /* Checkpoint::AgentState (sim.hpp:299-304) */
    float pos[3], rot[4], lin[3], ang[3];
    int32_t grab_idx;                     /* box index, or numBoxes + ramp index, or -1 */
    float grab_r1[3], grab_r2[3];
    float attach_rot1[4], attach_rot2[4], separation;
} hs_ckpt_agent;
typedef struct hs_checkpoint {
    uint32_t episode_key[2];              /* curEpisodeRNDCounter = {episode index, world id} */
    int32_t running_scores[2];            /* EpisodeStats (sim.hpp:109-111) */
    int32_t episode_step;
    hs_ckpt_agent agents[6];              /* hiders, then seekers */
    hs_ckpt_object boxes[9];
    hs_ckpt_object ramps[2];
    int32_t num_hiders, num_seekers, num_boxes, num_ramps;
} hs_checkpoint;

/* Manager::Manager (src/mgr.cpp:844-846 -> Impl::make :674-822). */
int32_t hs_create(const hs_config *cfg, hs_sim **out);
/* Manager::~Manager (src/mgr.cpp:848-859). */
void hs_destroy(hs_sim *sim);
/* Manager::init (src/mgr.cpp:861-881): runs the Init task graph (sim.cpp:1295-1305); blocking. */
int32_t hs_init(hs_sim *sim);
/* Manager::step (src/mgr.cpp:883-903): runs the Step task graph once (sim.cpp:1307-1313); blocking. */
int32_t hs_step(hs_sim *sim);
/* The two halves of hs_step, for a front-end that owns several handles (one per GPU of the node): every
 * handle has its own HIP stream; hs_step_begin orders that stream after the work already queued on the device's
 * legacy default stream (where torch writes `action`), enqueues the step and returns; hs_step_end waits for it and
 * reports device-side failures (hs_get_device_status).  hs_step == hs_step_begin + hs_step_end.  The reference is
 * single-GPU (src/mgr.hpp:18); these replace the executor's run() for the sharded case. */
int32_t hs_step_begin(hs_sim *sim);
int32_t hs_step_end(hs_sim *sim);
/* Manager::gpuJAXStep / CUDAImpl::gpuStreamStep (src/mgr.cpp:379-398, 1006-1022): enqueue one step on a
 * caller-supplied hipStream_t (passed as void*) without synchronising. */
int32_t hs_step_async(hs_sim *sim, void *hip_stream);

/* Render every agent's view of the current state into the depth / rgb exports (Manager::step's
 * renderMgr->batchRender(), src/mgr.cpp:894-901, with the camera of src/sim.cpp:1400-1403: 100 degrees vertical field
 * of view, z-near 0.001, 0.5 above the agent's origin, looking along the agent's forward axis).  depth [N*A,H,W,1] f32
 * = view-space depth of the closest hit (0: nothing hit / inactive agent); rgb [N*A,H,W,4] u8 = base colour of the hit
 * object (src/mgr.cpp:621-647, textures not reproduced) x (0.3 + 0.7 Lambert term of the light of :657-659), alpha 255.
 * Blocking.  Madrona's renderer is absent from the reference snapshot: the image is this build's own. */
int32_t hs_render(hs_sim *sim);
/* The 21 Manager::*Tensor() getters + policyAssignmentsTensor / episodeResultTensor
 * (src/mgr.cpp:1062-1336). */
int32_t hs_get_tensor(hs_sim *sim, int32_t export_id, hs_tensor_desc *out);
/* Manager::triggerReset (src/mgr.cpp:1265-1281). */
int32_t hs_trigger_reset(hs_sim *sim, int32_t world_idx, int32_t level_idx);
/* Manager::setAction (src/mgr.cpp:1283-1305). */
int32_t hs_set_action(hs_sim *sim, int32_t agent_idx, int32_t x, int32_t y, int32_t r, int32_t g, int32_t l);
/* Manager::saveCheckpoint (src/mgr.cpp:905-929): set world's CheckpointControl trigger, run the
 * SaveCheckpoints graph (sim.cpp:1315-1322: every triggered world writes its hs_checkpoint and clears the
 * trigger); blocking. */
int32_t hs_save_checkpoint(hs_sim *sim, int32_t world_idx);
/* Manager::loadCheckpoint (src/mgr.cpp:931-963): set the trigger, run the LoadCheckpoints graph. */
int32_t hs_load_checkpoint(hs_sim *sim, int32_t world_idx);
/* Manager::loadCheckpoints (src/mgr.cpp:965-985): run the LoadCheckpoints graph (sim.cpp:1324-1333) for
 * the triggers currently in the ckpt_ctrl tensor: triggered worlds regenerate their level from the saved
 * episode key and restore body / joint / episode state (sim.cpp:956-1044, trigger left at 1 as :963 does);
 * then observations are recomputed for all worlds. */
int32_t hs_load_checkpoints(hs_sim *sim);
/* CUDAImpl::saveCheckpoints (src/mgr.cpp:316-319): run the SaveCheckpoints graph for the current triggers. */
int32_t hs_save_checkpoints(hs_sim *sim);

/* The XLA-callable entry points behind `sim.jax()` (src/bindings.cpp:97-118): enqueue on the caller's
 * hipStream_t, device buffers in the reference's order, no synchronisation except hs_jax_init.
 *   obs block (JAXIOObservations, mgr.cpp:168-201): prep_counter, self_data, self_type, self_mask, lidar,
 *   agent_data, box_data, ramp_data, vis_agents_mask, vis_boxes_mask, vis_ramps_mask.
 * hs_jax_init  (gpuStreamInit mgr.cpp:362-376):  buffers = obs block (out).
 * hs_jax_step  (gpuStreamStep mgr.cpp:379-398):  buffers = actions, resets, policy_assignments (in), obs block,
 *                                                rewards, dones, episode_results (out).
 * hs_jax_save_checkpoints (mgr.cpp:400-416):     buffers = ckpt_ctrl (in), ckpts (out).
 * hs_jax_load_checkpoints (mgr.cpp:418-436):     buffers = ckpt_ctrl, ckpts (in), obs block (out). */
int32_t hs_jax_init(hs_sim *sim, void *hip_stream, void **buffers);
int32_t hs_jax_step(hs_sim *sim, void *hip_stream, void **buffers);
int32_t hs_jax_save_checkpoints(hs_sim *sim, void *hip_stream, void **buffers);
int32_t hs_jax_load_checkpoints(hs_sim *sim, void *hip_stream, void **buffers);

/* The same four functions as XLA GPU custom-call targets — what madrona::py::JAXInterface::buildEntry
 * (src/bindings.cpp:97-118) registers with XLA: `void target(stream, buffers, opaque, opaque_len)`, the original
 * status-less custom-call ABI; XLA passes its hipStream_t, the operand and result device buffers in call order (the
 * orders above) and the descriptor the Python side attached — the 8 bytes of the hs_sim* handle.  A failure cannot be
 * returned through this ABI: it is kept, hs_xla_last_status(clear) reports it, and the next blocking call on the
 * handle returns HS_ERR_HIP.  `sim.jax()` wraps the four addresses in PyCapsules named "xla._CUSTOM_CALL_TARGET". */
void hs_xla_init(void *hip_stream, void **buffers, const char *opaque, size_t opaque_len);
void hs_xla_step(void *hip_stream, void **buffers, const char *opaque, size_t opaque_len);
void hs_xla_save_checkpoints(void *hip_stream, void **buffers, const char *opaque, size_t opaque_len);
void hs_xla_load_checkpoints(void *hip_stream, void **buffers, const char *opaque, size_t opaque_len);
int32_t hs_xla_last_status(int32_t clear);

/* Manager::trainInterface (src/mgr.cpp:1338-1375): the names and roles under which `sim.jax()` hands the exported
 * tensors to the learner, in the reference's order (which is also the buffer order of hs_jax_step).  `export_id` is
 * an HS_EXPORT_* value, or -1 for the empty simCtrl tensor (mgr.cpp:1333-1336).  Returns the number of entries and
 * points *entries at a static table. */
enum {
    HS_ROLE_ACTION = 0,        /* TrainInterface inputs.actions */
    HS_ROLE_RESET = 1,         /* inputs.resets */
    HS_ROLE_SIM_CTRL = 2,      /* inputs.simCtrl (empty) */
    HS_ROLE_PBT_INPUT = 3,     /* inputs.pbt */
    HS_ROLE_OBSERVATION = 4,   /* outputs.observations */
    HS_ROLE_REWARD = 5,        /* outputs.rewards */
    HS_ROLE_DONE = 6,          /* outputs.dones */
    HS_ROLE_PBT_OUTPUT = 7,    /* outputs.pbt */
    HS_ROLE_CHECKPOINT = 8     /* TrainCheckpointingInterface.checkpointData */
};
typedef struct hs_iface_entry {
    const char *name;
    int32_t role;
    int32_t export_id;
} hs_iface_entry;
int32_t hs_train_interface(const hs_iface_entry **entries);

/* Device-side conditions the reference has no channel for (it asserts or aborts).
 *   spilled_dd_pairs / spilled_static_pairs: broadphase candidate pairs of a world beyond what the physics kernel keeps
 *     in LDS per substep (16 body-body, 24 body-static).  They are NOT lost: they go through a global spill list sized
 *     for the worst case (17 bodies: 136 pairs; 17 x 38 statics — the reference sizes for every entity too,
 *     src/sim.cpp:1356-1361) and are tested and solved in the same order as every other pair, on a slower path; results
 *     are identical to an unbounded solver.  Sticky totals since hs_create; a performance signal, not an error.
 *   dropped_dd_pairs / dropped_static_pairs: candidate pairs discarded.  Always 0 (nothing can overflow the spill
 *     lists); kept so that callers can assert it (bench.py, tools/train_config_bench.py do).
 *   graphs_in_use: 1 when HS_GRAPH=1 took effect and steps are replayed as HIP graphs. */
typedef struct hs_device_status {
    int64_t dropped_dd_pairs;
    int64_t dropped_static_pairs;
    int32_t graphs_in_use;
    int32_t reserved;
    int64_t spilled_dd_pairs;
    int64_t spilled_static_pairs;
} hs_device_status;
int32_t hs_get_device_status(hs_sim *sim, hs_device_status *out);

/* maxAgentsPerWorld (src/mgr.cpp:684). */
int32_t hs_agents_per_world(const hs_sim *sim);

/* Internal-state dumps (not in the reference; used by the parity tests and by INTEGRATION.md's
 * shard-equivalence check).  Host output buffers:
 *   bodies [num_worlds][17][13] f32 = pos3 rot4(wxyz) lin3 ang3, meta [num_worlds][17][3] i32 =
 *   objType, response, owner;  walls [num_worlds][36][4] f32 = cx cy hx hy,
 *   info [num_worlds][8] i32 = numWalls numPlanes numActiveBoxes numActiveRamps numHiders numSeekers
 *   curEpisodeStep seekersFirst. */
int32_t hs_debug_dump_bodies(hs_sim *sim, float *bodies, int32_t *meta);
int32_t hs_debug_dump_walls(hs_sim *sim, float *walls, int32_t *info);

/* Milliseconds of device time of the last `hs_step`, measured with HIP events on the launch stream when
 * profiling is enabled: [0] k_physics (movement / actions, 4 XPBD substeps, rewards), [1] k_reset,
 * [2] k_observe. */
int32_t hs_set_profiling(hs_sim *sim, int32_t enabled);
int32_t hs_last_step_kernel_ms(hs_sim *sim, float out_ms[3]);

/* Development aid: per-phase wall-clock ticks (100 MHz) accumulated by every workgroup of the physics kernel,
 * [groups][10] = pre, integrate, detect, sat, dd_pos, body_pos, dd_vel, body_vel, post, -; all zero unless the
 * library was built with -DHS_PHASE_TIMING.  Returns the number of groups written (<= max_groups). */
int32_t hs_debug_phase_ticks(hs_sim *sim, int64_t *out, int32_t max_groups);
/* Development aid: work counters per WORLD, accumulated since init, [worlds][8] = body-body candidates, body-static
 * candidates, accepted body-body manifolds, body-body solver rounds the world was pending in, the physics wave the world
 * lives in, the solver rounds that wave ran (first world of a wave only), -, -; all zero unless the library was built with
 * -DHS_LOAD_STUDY (tools/load_study.py).  Returns the number of worlds written (<= max_worlds). */
int32_t hs_debug_load_study(hs_sim *sim, int64_t *out, int32_t max_worlds);
/* The same for k_observe: ticks per section, summed over all waves: stage (incl. the schedule's wait), per-agent table,
 * ray setup, walls, planes, hull cull, exact hull tests, ray results, observation rows. */
int32_t hs_debug_observe_ticks(hs_sim *sim, int64_t out[16]);
/* ... and the work counters of the convex tests, summed over all waves and substeps: calls, box-shaped items, wedge
 * items, rounds of 32 pairs, colliding pairs, contact-generation rounds. */
int32_t hs_debug_sat_counters(hs_sim *sim, int64_t out[16]);

/* The DEVICE's hull tables for one SimObject (src/sim.hpp:78-88; 3 = wall, unit size), evaluated by a one-lane kernel through
 * the very functions the convex tests use (csrc/hs_collide.h: packed topology, closed-form vertices) at the identity
 * pose: verts [8][3], faces [6][4] (-1 padded), counts {nv, nf, ne, ned}, normals [6][3], edges [12][3] = v0 v1 dir,
 * local [8][3] = hull_local_vertex (the contact points of plane manifolds).  tests/test_gpu_hulls.py pins them to
 * tests/golden/hulls.npz, i.e. to data/{cube,wall,agent,ramp,elongated}_collision.obj of the reference. */
int32_t hs_debug_dump_hull(int32_t obj, float *verts, int32_t *faces, int32_t *counts, float *normals, int32_t *edges,
                           float *local);

/* The DEVICE's object table for one SimObject: out[6] = inverse mass, static / dynamic friction coefficient, inverse inertia x y z
 * (object frame) as the kernels use them (csrc/hs_dev.h).  tests/test_gpu_hulls.py pins the first three and the zeroed
 * inertia axes of the agents to tests/golden/object_table.json (src/mgr.cpp:441-588). */
int32_t hs_debug_object_params(int32_t obj, float *out);

/* Profiling aid: one dword-per-lane coalesced copy of `bytes` bytes (read + write), used to calibrate the
 * rocprofv3 FETCH_SIZE / WRITE_SIZE counters for the simulator's access pattern. */
int32_t hs_debug_calibrate(int64_t bytes);

/* No-op `DLManagedTensor::deleter` for the non-owning DLPack views built by language bindings. */
void hs_dlpack_noop_deleter(void *managed_tensor);

const char *hs_last_error(void);
const char *hs_version(void);

#ifdef __cplusplus
}
#endif
#endif /* HIDESEEK_H */
