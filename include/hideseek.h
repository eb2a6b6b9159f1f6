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
    HS_FLAG_EXT_SKIP_OBSERVATIONS = 1 << 16
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
    int32_t enable_batch_renderer;   /* accepted; rgb/depth tensors are allocated, never rendered */
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

/* Manager::Manager (src/mgr.cpp:844-846 -> Impl::make :674-822). */
int32_t hs_create(const hs_config *cfg, hs_sim **out);
/* Manager::~Manager (src/mgr.cpp:848-859). */
void hs_destroy(hs_sim *sim);
/* Manager::init (src/mgr.cpp:861-881): runs the Init task graph (sim.cpp:1295-1305); blocking. */
int32_t hs_init(hs_sim *sim);
/* Manager::step (src/mgr.cpp:883-903): runs the Step task graph once (sim.cpp:1307-1313); blocking. */
int32_t hs_step(hs_sim *sim);
/* Manager::gpuJAXStep / CUDAImpl::gpuStreamStep (src/mgr.cpp:379-398, 1006-1022): enqueue one step on a
 * caller-supplied hipStream_t (passed as void*) without synchronising. */
int32_t hs_step_async(hs_sim *sim, void *hip_stream);
/* The 21 Manager::*Tensor() getters + policyAssignmentsTensor / episodeResultTensor
 * (src/mgr.cpp:1062-1336). */
int32_t hs_get_tensor(hs_sim *sim, int32_t export_id, hs_tensor_desc *out);
/* Manager::triggerReset (src/mgr.cpp:1265-1281). */
int32_t hs_trigger_reset(hs_sim *sim, int32_t world_idx, int32_t level_idx);
/* Manager::setAction (src/mgr.cpp:1283-1305). */
int32_t hs_set_action(hs_sim *sim, int32_t agent_idx, int32_t x, int32_t y, int32_t r, int32_t g, int32_t l);
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
 * profiling is enabled: [0] the physics pipeline (k_pre, 4 x 9 substep kernels, k_post), [1] k_reset,
 * [2] k_observe. */
int32_t hs_set_profiling(hs_sim *sim, int32_t enabled);
int32_t hs_last_step_kernel_ms(hs_sim *sim, float out_ms[3]);

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
