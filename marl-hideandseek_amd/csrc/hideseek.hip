// libhideseek.so — host side of the C ABI declared in include/hideseek.h.
// Owns the HBM allocations, launches the two kernels of a step on the handle's HIP stream and
// hands out non-owning tensor descriptors (replaces Manager::Impl, src/mgr.cpp:86-437, 674-822).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <string>
#include <vector>
#include <atomic>

#include "../../include/hideseek.h"
#include "hs_state.h"
#include "hs_k_reset.h"
#include "hs_k_observe.h"
#include "hs_k_render.h"
#include "hs_k_physics.h"
#include "hs_k_balance.h"
#include "hs_solver.h"

namespace {

thread_local std::string g_err;

int fail(int code, const std::string &msg) { g_err = msg; return code; }

#define HS_HIP(expr)                                                                          \
    do {                                                                                      \
        hipError_t e_ = (expr);                                                               \
        if (e_ != hipSuccess)                                                                 \
            return fail(HS_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(e_));       \
    } while (0)

}  // namespace

struct hs_sim {
    hs_config cfg;
    hs::SimState S;
    int A;
    std::vector<void *> allocs;
    hs_tensor_desc exports[HS_NUM_EXPORTS];
    bool profiling = false;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    float last_ms[3] = {0.f, 0.f, 0.f};
    bool initialised = false;
    bool use_graph = false;                // env HS_GRAPH=1: replay the step as HIP graphs (measured 2 % slower than two direct launches)
    hipGraphExec_t graph_exec[2] = {nullptr, nullptr};     // physics, observe
    hipStream_t step_stream = nullptr;     // the stream of the open step
    bool blocking_own_stream = false;      // HS_STREAM=own: hs_step uses the handle's stream as hs_step_begin does
    int step_idx = 0;                      // physics launches mod 3 (SimState::tickSum)
    int tile = 8;                          // worlds per physics wave: 8 (two waves per SIMD) or 4 (four waves per SIMD); HS_TILE
    // Load balancing between the physics waves (hs_k_balance.h; HS_BALANCE=0 turns it off, HS_BALANCE_PERIOD sets the steps between deals)
    bool balance = true;
    int balance_period = 32, steps_since_balance = 0;
    int *bal_hist = nullptr, *bal_cursor = nullptr, *bal_new_slot = nullptr;
    void *bal_tmp = nullptr;               // a second arena: the deal copies the columns there and moves them back to their new slots
    void *col_arena = nullptr; size_t col_arena_bytes = 0, col_slots = 0;      // the tiled columns, one after the other
    hs::BalanceCols bal_cols;              // first arena row of each column (+ the total)
    hipStream_t stream = nullptr;          // this handle's own stream: hs_init / hs_step / checkpoints run here
    hipEvent_t evIn = nullptr;             // orders `stream` after the device's legacy default stream (torch's writes to `action`)
    bool step_open = false;                // hs_step_begin without its hs_step_end
    std::atomic<int32_t> async_error{0};   // a failed XLA custom call on this handle (hs_xla_*: the ABI has no status channel; XLA's thread)

    template <typename T> int dalloc(T **p, size_t n, int fill_byte = 0) {
        void *d = nullptr;
        HS_HIP(hipMalloc(&d, n * sizeof(T) > 0 ? n * sizeof(T) : 4));
        HS_HIP(hipMemset(d, fill_byte, n * sizeof(T) > 0 ? n * sizeof(T) : 4));
        allocs.push_back(d);
        *p = (T *)d;
        return HS_OK;
    }
};

namespace {

void set_desc(hs_sim *s, int id, void *ptr, int dtype, std::initializer_list<int64_t> dims) {
    hs_tensor_desc &d = s->exports[id];
    d.ptr = ptr; d.dtype = dtype; d.ndim = (int32_t)dims.size(); d.gpu_id = s->cfg.gpu_id;
    int i = 0;
    for (int64_t v : dims) d.dims[i++] = v;
    for (; i < 4; ++i) d.dims[i] = 1;
}

// The renderer outputs (Manager::depthTensor / rgbTensor, src/mgr.cpp:1241-1263): allocated on first need, zero-filled.
int ensure_render_buffers(hs_sim *s) {
    if (s->exports[HS_EXPORT_RGB].ptr && s->exports[HS_EXPORT_DEPTH].ptr) return HS_OK;
    const int64_t r = (int64_t)s->S.N * s->A;
    const int64_t H = s->cfg.batch_render_height > 0 ? s->cfg.batch_render_height : 64;
    const int64_t Wd = s->cfg.batch_render_width > 0 ? s->cfg.batch_render_width : 64;
    int rc;
    if (!s->exports[HS_EXPORT_RGB].ptr) {
        uint8_t *p; if ((rc = s->dalloc(&p, (size_t)(r * H * Wd * 4))) != HS_OK) return rc;
        set_desc(s, HS_EXPORT_RGB, p, HS_DTYPE_U8, {r, H, Wd, 4});
    }
    if (!s->exports[HS_EXPORT_DEPTH].ptr) {
        float *p; if ((rc = s->dalloc(&p, (size_t)(r * H * Wd))) != HS_OK) return rc;
        set_desc(s, HS_EXPORT_DEPTH, p, HS_DTYPE_F32, {r, H, Wd, 1});
    }
    return HS_OK;
}
// k_render over every view (hs_k_render.h); the buffers exist (ensure_render_buffers).
void launch_render(hs_sim *s, hipStream_t strm) {
    const hs_tensor_desc &d = s->exports[HS_EXPORT_DEPTH];
    const int nslots = (s->S.N + hs::kTile - 1) / hs::kTile * hs::kTile;
    hipLaunchKernelGGL(hs::k_render, dim3(nslots * s->A), dim3(hs::kRenderThreads), 0, strm, s->S, (float *)d.ptr,
                       (unsigned *)s->exports[HS_EXPORT_RGB].ptr, (int)d.dims[2], (int)d.dims[1]);
}

void launch_observe(hs_sim *s, hipStream_t strm) {
    hs::SimState S = s->S;
    if (S.flags & hs::FLAG_EXT_SKIP_OBSERVATIONS) return;
    struct RenderAfter {        // HS_FLAG_EXT_RENDER: the agent views are part of every step's observations
        hs_sim *s; hipStream_t strm;
        ~RenderAfter() { if ((s->S.flags & hs::FLAG_EXT_RENDER) && s->exports[HS_EXPORT_DEPTH].ptr) launch_render(s, strm); }
    } render_after{s, strm};
    // one workgroup per world; the grid covers whole groups of 8 octets (k_observe's block -> world mapping)
    const int N = ((S.N + hs::kTile - 1) / hs::kTile + 7) / 8 * 64;
    const int nt = hs::obs_threads(s->A);                          // a lane per ray (hs_k_observe.h)
    if (nt <= 128) hipLaunchKernelGGL(hs::k_observe<128>, dim3(N), dim3(128), 0, strm, S);
    else if (nt <= 192) hipLaunchKernelGGL(hs::k_observe<192>, dim3(N), dim3(192), 0, strm, S);
    else hipLaunchKernelGGL(hs::k_observe<320>, dim3(N), dim3(320), 0, strm, S);
}

// One step = k_physics (movement + actionSystem, 4 XPBD substeps, rewards / dones / episode results, reset: one
// kernel, a wave per octet of 8 worlds, hs_k_physics.h) and k_observe.  Manager::init = k_reset then k_observe.
// `stages`: 1 physics, 2 reset (init only), 4 observe.
int launch_step_eager(hs_sim *s, hipStream_t strm, bool first, bool prof, int stages = 7, bool capture = false) {
    hs::SimState S = s->S;
    const int N = S.N, noct = (N + hs::kTile - 1) / hs::kTile;
    // (a captured launch keeps its arguments for ever: it gets no step index, and the kernel skips the per-wave priority hint)
    S.stepIdx = capture ? -1 : s->step_idx; if (!capture && !first && (stages & 1)) s->step_idx = (s->step_idx + 1) % 3;
    if (prof) HS_HIP(hipEventRecord(s->ev[0], strm));
    if (!first && (stages & 1)) {
        // rounds of 64 lanes over the tile's bodies: 16 body slots per world with up to 5 agents, 17 with 6
        const bool six = s->A > hs::kMaxAgents - 1;
        if (s->tile == 8) {
            if (six) hipLaunchKernelGGL((hs::k_physics<3, 8>), dim3(noct), dim3(hs::kPhysThreads), hs::kPhysDynLds, strm, S);
            else hipLaunchKernelGGL((hs::k_physics<2, 8>), dim3(noct), dim3(hs::kPhysThreads), hs::kPhysDynLds, strm, S);
        } else {
            if (six) hipLaunchKernelGGL((hs::k_physics<2, 4>), dim3(2 * noct), dim3(hs::kPhysThreads), hs::kPhysDynLds / 2, strm, S);
            else hipLaunchKernelGGL((hs::k_physics<1, 4>), dim3(2 * noct), dim3(hs::kPhysThreads), hs::kPhysDynLds / 2, strm, S);
        }
    }
    if (prof) HS_HIP(hipEventRecord(s->ev[1], strm));
    // in a step the reset is the tail of k_physics; only Manager::init launches it on its own
    if (first && (stages & 2)) hipLaunchKernelGGL(hs::k_reset, dim3((N + 31) / 32), dim3(32), 0, strm, S);     // half-filled waves: the generator diverges per world
    if (prof) HS_HIP(hipEventRecord(s->ev[2], strm));
    if (stages & 4) launch_observe(s, strm);
    if (prof) HS_HIP(hipEventRecord(s->ev[3], strm));
    HS_HIP(hipGetLastError());
    return HS_OK;
}

// Deal the worlds to the octets by contact load (hs_k_balance.h), every balance_period steps: histogram, scan, deal, one copy
// of the column arena, one move kernel over all columns, commit (six launches and a copy; it was 26 launches and 11 copies).
int balance_worlds(hs_sim *s, hipStream_t strm) {
    const hs::SimState &S = s->S;
    const int nfull = S.N / hs::kTile * hs::kTile;
    if (nfull < 2 * hs::kTile) return HS_OK;
    const dim3 grid((nfull + 255) / 256), blk(256);
    hipLaunchKernelGGL(hs::k_balance_hist, grid, blk, 0, strm, S, nfull, s->bal_hist);
    hipLaunchKernelGGL(hs::k_balance_scan, dim3(1), dim3(hs::kBalanceBins), 0, strm, s->bal_hist, s->bal_cursor);
    hipLaunchKernelGGL(hs::k_balance_deal, grid, blk, 0, strm, S, nfull, s->bal_cursor, s->bal_new_slot, s->tile);
    HS_HIP(hipMemcpyAsync(s->bal_tmp, s->col_arena, s->col_arena_bytes, hipMemcpyDeviceToDevice, strm));
    const size_t n = (size_t)nfull * s->bal_cols.base[hs::kBalanceCols];
    hipLaunchKernelGGL(hs::k_balance_move_all, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, strm, (int *)s->col_arena, (const int *)s->bal_tmp,
                       s->bal_cols, s->col_slots, (const int *)s->S.slotOfWorld, (const int *)s->bal_new_slot, nfull);
    hipLaunchKernelGGL(hs::k_balance_commit, grid, blk, 0, strm, S, nfull, (const int *)s->bal_new_slot);
    HS_HIP(hipGetLastError());
    return HS_OK;
}

// Optional (HS_GRAPH=1): the step as two HIP graphs (physics, observe), captured once on a private stream (the
// legacy stream cannot be captured) and replayed on the caller's stream; the profiling events stay ordinary stream
// events between the graph launches.  It paid off while physics was ~40 launches per step; with the persistent
// physics kernel a step is two launches and the direct launches are faster.  When capture or instantiation fails
// the handle falls back to direct launches and hs_get_device_status reports graphs_in_use = 0.
int launch_step(hs_sim *s, hipStream_t strm, bool first) {
    if (!first && s->balance && ++s->steps_since_balance >= s->balance_period) {
        s->steps_since_balance = 0;
        int rc = balance_worlds(s, strm);
        if (rc != HS_OK) return rc;
    }
    if (first || !s->use_graph) return launch_step_eager(s, strm, first, s->profiling);
    const bool skip_obs = (s->S.flags & hs::FLAG_EXT_SKIP_OBSERVATIONS) != 0;
    const int ngraphs = skip_obs ? 1 : 2;
    if (!s->graph_exec[0]) {
        hipStream_t cap = nullptr;
        HS_HIP(hipStreamCreateWithFlags(&cap, hipStreamNonBlocking));
        bool ok = true;
        for (int g = 0; g < ngraphs && ok; ++g) {
            hipGraph_t gr = nullptr;
            ok = hipStreamBeginCapture(cap, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) ok = launch_step_eager(s, cap, false, false, g == 0 ? 1 : 4, true) == HS_OK;
            if (hipStreamEndCapture(cap, &gr) != hipSuccess) ok = false;
            if (ok) ok = hipGraphInstantiate(&s->graph_exec[g], gr, nullptr, nullptr, 0) == hipSuccess;
            if (gr) hipGraphDestroy(gr);
        }
        hipStreamDestroy(cap);
        if (!ok) {
            for (auto &e : s->graph_exec) { if (e) hipGraphExecDestroy(e); e = nullptr; }
            s->use_graph = false; (void)hipGetLastError();
            return launch_step_eager(s, strm, first, s->profiling);
        }
    }
    const bool prof = s->profiling;
    if (prof) HS_HIP(hipEventRecord(s->ev[0], strm));
    HS_HIP(hipGraphLaunch(s->graph_exec[0], strm));
    if (prof) { HS_HIP(hipEventRecord(s->ev[1], strm)); HS_HIP(hipEventRecord(s->ev[2], strm)); }
    if (!skip_obs) HS_HIP(hipGraphLaunch(s->graph_exec[1], strm));
    if (prof) HS_HIP(hipEventRecord(s->ev[3], strm));
    return HS_OK;
}

// A failure of an earlier asynchronous entry point (the XLA custom calls have no status channel) surfaces at the next call
// that can report it.  Nothing here touches the device: the sticky counters of hs_get_device_status (candidate pairs
// that took the spill path) are read only on request, so no step ever pays a blocking copy for them.
int poll_status(hs_sim *s) {
    const int32_t e = s->async_error.exchange(0);
    if (e != 0) return fail(HS_ERR_HIP, "an XLA custom call on this simulator failed earlier with status " + std::to_string(e));
    return HS_OK;
}

}  // namespace

namespace {
// Host copy of a tiled column (hs_state.h Col): element (row, world) at ((w / 8) * ROWS + row) * 8 + w % 8.
template <typename T, int ROWS>
struct HostCol {
    std::vector<T> v;
    int load(const hs::Col<T, ROWS> &c, size_t n) {
        v.resize((n + hs::kTile - 1) / hs::kTile * hs::kTile * ROWS);
        HS_HIP(hipMemcpy(v.data(), c.p, v.size() * sizeof(T), hipMemcpyDeviceToHost));
        return HS_OK;
    }
    T operator()(size_t row, size_t p) const { return v[((p >> 3) * ROWS + row) * hs::kTile + (p & 7)]; }     // p = slot
};
int load_slots(hs_sim *s, std::vector<int32_t> &slot) {
    slot.resize(s->S.N);
    HS_HIP(hipMemcpy(slot.data(), s->S.slotOfWorld, slot.size() * 4, hipMemcpyDeviceToHost));
    return HS_OK;
}
}  // namespace

extern "C" {

const char *hs_last_error(void) { return g_err.c_str(); }
const char *hs_version(void) { return "hideseek-mi355x 0.1 (gfx950)"; }

int32_t hs_create(const hs_config *cfg, hs_sim **out) {
    if (!cfg || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    *out = nullptr;
    if (cfg->exec_mode != HS_EXEC_GPU)
        return fail(HS_ERR_UNSUPPORTED,
                    "exec_mode CPU is not provided by libhideseek: the HIP path is the only execution path");
    const int A = cfg->max_hiders + cfg->max_seekers;
    if (cfg->num_worlds <= 0) return fail(HS_ERR_INVALID_ARG, "num_worlds must be > 0");
    if (A <= 0 || A > hs::kMaxAgents || cfg->max_hiders > 3 || cfg->max_seekers > 3 || cfg->min_hiders < 0 ||
        cfg->min_seekers < 0 || cfg->min_hiders > cfg->max_hiders || cfg->min_seekers > cfg->max_seekers)
        return fail(HS_ERR_INVALID_ARG, "hider/seeker counts out of range (<= 3 each, src/sim.hpp:338-341)");
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(HS_ERR_NO_DEVICE, "no HIP device visible: libhideseek has no CPU fallback");
    if (cfg->gpu_id < 0 || cfg->gpu_id >= ndev) return fail(HS_ERR_INVALID_ARG, "gpu_id out of range");
    HS_HIP(hipSetDevice(cfg->gpu_id));

    hs_sim *s = new hs_sim();
    s->cfg = *cfg;
    s->A = A;
    hs::SimState &S = s->S;
    std::memset(&S, 0, sizeof(S));
    const size_t N = (size_t)cfg->num_worlds, R = N * (size_t)A;
    S.N = (int)N; S.A = A; S.flags = cfg->sim_flags;
    S.initKey = {cfg->rand_seed, 0u};          // rand::initKey (mgr.cpp:678)
    S.minHiders = cfg->min_hiders; S.maxHiders = cfg->max_hiders;
    S.minSeekers = cfg->min_seekers; S.maxSeekers = cfg->max_seekers;
    S.worldOffset = cfg->world_offset;
    const int D = hs::kNumDSlots, AG = hs::kMaxAgents;
    int rc = HS_OK;
#define HS_ALLOC(ptr, n) if ((rc = s->dalloc(&(ptr), (n))) != HS_OK) { hs_destroy(s); return rc; }
    // tiled columns (hs_state.h Col): whole octets, the padding worlds stay zero = empty slots
    const size_t NP = (N + hs::kTile - 1) / hs::kTile * hs::kTile;
    // the tiled columns are consecutive pieces of ONE arena (so that the periodic deal moves them in one launch): all have
    // 4-byte elements; a column of ROWS rows takes ROWS * NP of them
    {
        const int rows[hs::kBalanceCols] = {S.bpos.kRows, S.brot.kRows, S.blin.kRows, S.bang.kRows, S.bmeta.kRows, S.aforce.kRows, S.walls.kRows,
                                            S.planes.kRows, S.runningScores.kRows, S.grabOther.kRows, S.grabData.kRows};
        s->bal_cols.base[0] = 0;
        for (int k = 0; k < hs::kBalanceCols; ++k) s->bal_cols.base[k + 1] = s->bal_cols.base[k] + rows[k];
        s->col_slots = NP;
        s->col_arena_bytes = (size_t)s->bal_cols.base[hs::kBalanceCols] * NP * 4;
        char *arena, *tmp;
        HS_ALLOC(arena, s->col_arena_bytes); HS_ALLOC(tmp, s->col_arena_bytes);
        s->col_arena = arena; s->bal_tmp = tmp;
        int k = 0;
#define HS_ARENA_COL(col) (col).p = (decltype((col).p))(arena + (size_t)s->bal_cols.base[k++] * NP * 4);
        HS_ARENA_COL(S.bpos) HS_ARENA_COL(S.brot) HS_ARENA_COL(S.blin) HS_ARENA_COL(S.bang) HS_ARENA_COL(S.bmeta) HS_ARENA_COL(S.aforce)
        HS_ARENA_COL(S.walls) HS_ARENA_COL(S.planes) HS_ARENA_COL(S.runningScores) HS_ARENA_COL(S.grabOther) HS_ARENA_COL(S.grabData)
#undef HS_ARENA_COL
    }
    HS_ALLOC(S.numWalls, N); HS_ALLOC(S.numPlanes, N);
    HS_ALLOC(S.curWorldEpisode, N); HS_ALLOC(S.rngKeyA, N); HS_ALLOC(S.rngKeyB, N); HS_ALLOC(S.rngCount, N);
    HS_ALLOC(S.curEpisodeStep, N); HS_ALLOC(S.hiderTeamReward, N); HS_ALLOC(S.counts, N); HS_ALLOC(S.teams, N);
    HS_ALLOC(S.epKeyA, N); HS_ALLOC(S.epKeyB, N); HS_ALLOC(S.xCkptCtrl, N); HS_ALLOC(S.xCkpt, N * sizeof(hs_checkpoint));
    HS_ALLOC(S.xReset, N); HS_ALLOC(S.xPrep, R); HS_ALLOC(S.xAction, R * 5); HS_ALLOC(S.xSelfType, R);
    HS_ALLOC(S.xSeed, R * 2); HS_ALLOC(S.xDone, R); HS_ALLOC(S.xPolicy, R);
    HS_ALLOC(S.xSelfObs, R * 13); HS_ALLOC(S.xSelfMask, R); HS_ALLOC(S.xAgentObs, R * 5 * 14);
    HS_ALLOC(S.xBoxObs, R * 9 * 17); HS_ALLOC(S.xRampObs, R * 2 * 14); HS_ALLOC(S.xVisAgents, R * 5);
    HS_ALLOC(S.xVisBoxes, R * 9); HS_ALLOC(S.xVisRamps, R * 2); HS_ALLOC(S.xLidar, R * 30);
    HS_ALLOC(S.xReward, R); HS_ALLOC(S.xGlobalPos, N * 34); HS_ALLOC(S.xEpisodeResult, N * 2);
    // (every possible pair of every world: 92 KB per world, 1.5 GB at 16 000 worlds, of which a step touches a few MB)
    { char *p; HS_ALLOC(p, NP * hs::kAllDD * sizeof(hs::ManDD)); S.wsDD = p; HS_ALLOC(p, NP * hs::kAllSC * sizeof(hs::ManS)); S.wsSC = p; }
    HS_ALLOC(S.spPair, NP * (hs::kAllDD + hs::kAllSC)); HS_ALLOC(S.spInfo, NP * hs::kSpInfoWords);
    HS_ALLOC(S.phaseTicks, hs::phase_ticks_study_base((int)N) + N * hs::kStudyWords);   // + k_observe's section counters + the per-world work counters
    HS_ALLOC(S.slotOfWorld, N); HS_ALLOC(S.worldOfSlot, NP); HS_ALLOC(S.loadAcc, N); HS_ALLOC(S.wallHist, N);
    if ((rc = s->dalloc(&S.slotHdr, NP, 0xFF)) != HS_OK) { hs_destroy(s); return rc; }      // world id -1: empty slot
    HS_ALLOC(S.lidarSinCos, 60);
    HS_ALLOC(S.octTicks, NP / 4); HS_ALLOC(S.tickSum, 3);
    HS_ALLOC(s->bal_hist, hs::kBalanceBins); HS_ALLOC(s->bal_cursor, hs::kBalanceBins); HS_ALLOC(s->bal_new_slot, N);
    HS_ALLOC(S.status, 4);
#undef HS_ALLOC
    // Sim::Sim (sim.cpp:1346-1408): resetLevel = 1 for every world, no grab joints
    {
        std::vector<int32_t> ones(N, 1), neg(AG * NP, -1), ident(NP);
        for (size_t i = 0; i < NP; ++i) ident[i] = i < N ? (int32_t)i : -1;      // slot == world until k_balance deals them
        if (hipMemcpy(S.slotOfWorld, ident.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(S.worldOfSlot, ident.data(), NP * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(S.xReset, ones.data(), N * 4, hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(S.grabOther.p, neg.data(), AG * NP * 4, hipMemcpyHostToDevice) != hipSuccess) {
            hs_destroy(s);
            return fail(HS_ERR_HIP, "initial upload failed");
        }
    }
    for (auto &e : s->ev) {
        if (hipEventCreate(&e) != hipSuccess) { hs_destroy(s); return fail(HS_ERR_HIP, "hipEventCreate failed"); }
    }
    if (const char *e = getenv("HS_GRAPH")) s->use_graph = atoi(e) != 0;
    if (const char *e = getenv("HS_STREAM")) s->blocking_own_stream = std::strcmp(e, "own") == 0;
    if (const char *e = getenv("HS_TILE")) { const int v = atoi(e); if (v == 4 || v == 8) s->tile = v; }
    if (const char *e = getenv("HS_BALANCE")) s->balance = atoi(e) != 0;
    if (const char *e = getenv("HS_BALANCE_PERIOD")) { const int v = atoi(e); if (v > 0) s->balance_period = v; }
    if (hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking) != hipSuccess ||
        hipEventCreateWithFlags(&s->evIn, hipEventDisableTiming) != hipSuccess) { hs_destroy(s); return fail(HS_ERR_HIP, "stream/event creation failed"); }
    S.wbeg = 0; S.wcnt = (int)N;
    hipLaunchKernelGGL(hs::k_lidar_table, dim3(1), dim3(64), 0, s->stream, S);
    if (hipStreamSynchronize(s->stream) != hipSuccess) { hs_destroy(s); return fail(HS_ERR_HIP, "k_lidar_table failed"); }
    std::memset(s->exports, 0, sizeof(s->exports));
    const int64_t n = (int64_t)N, r = (int64_t)R;
    set_desc(s, HS_EXPORT_RESET, S.xReset, HS_DTYPE_I32, {n, 1});
    set_desc(s, HS_EXPORT_PREP_COUNTER, S.xPrep, HS_DTYPE_I32, {r, 1});
    set_desc(s, HS_EXPORT_ACTION, S.xAction, HS_DTYPE_I32, {r, 5});
    set_desc(s, HS_EXPORT_SELF_OBS, S.xSelfObs, HS_DTYPE_F32, {r, 13});
    set_desc(s, HS_EXPORT_SELF_TYPE, S.xSelfType, HS_DTYPE_I32, {r, 1});
    set_desc(s, HS_EXPORT_SELF_MASK, S.xSelfMask, HS_DTYPE_F32, {r, 1});
    set_desc(s, HS_EXPORT_AGENT_OBS, S.xAgentObs, HS_DTYPE_F32, {r, 5, 14});
    set_desc(s, HS_EXPORT_BOX_OBS, S.xBoxObs, HS_DTYPE_F32, {r, 9, 17});
    set_desc(s, HS_EXPORT_RAMP_OBS, S.xRampObs, HS_DTYPE_F32, {r, 2, 14});
    set_desc(s, HS_EXPORT_AGENT_VIS_MASKS, S.xVisAgents, HS_DTYPE_F32, {r, 5, 1});
    set_desc(s, HS_EXPORT_BOX_VIS_MASKS, S.xVisBoxes, HS_DTYPE_F32, {r, 9, 1});
    set_desc(s, HS_EXPORT_RAMP_VIS_MASKS, S.xVisRamps, HS_DTYPE_F32, {r, 2, 1});
    set_desc(s, HS_EXPORT_LIDAR, S.xLidar, HS_DTYPE_F32, {r, 30});
    set_desc(s, HS_EXPORT_SEED, S.xSeed, HS_DTYPE_I32, {r, 2});
    set_desc(s, HS_EXPORT_REWARD, S.xReward, HS_DTYPE_F32, {r, 1});
    set_desc(s, HS_EXPORT_DONE, S.xDone, HS_DTYPE_I32, {r, 1});
    set_desc(s, HS_EXPORT_GLOBAL_DEBUG_POSITIONS, S.xGlobalPos, HS_DTYPE_F32, {n, 17, 2});
    set_desc(s, HS_EXPORT_AGENT_POLICY, S.xPolicy, HS_DTYPE_I32, {r, 1});
    set_desc(s, HS_EXPORT_EPISODE_RESULT, S.xEpisodeResult, HS_DTYPE_F32, {n, 2});
    // raw bytes, as the reference exports them (mgr.cpp:1209-1227)
    set_desc(s, HS_EXPORT_CHECKPOINT_CONTROL, S.xCkptCtrl, HS_DTYPE_U8, {n, (int64_t)sizeof(int32_t)});
    set_desc(s, HS_EXPORT_CHECKPOINT, S.xCkpt, HS_DTYPE_U8, {n, (int64_t)sizeof(hs_checkpoint)});
    if ((S.flags & hs::FLAG_EXT_RENDER) && cfg->enable_batch_renderer) {      // rendered by every init / step
        const int rc = ensure_render_buffers(s);
        if (rc != HS_OK) { hs_destroy(s); return rc; }
    } else {
        s->S.flags &= ~(uint32_t)hs::FLAG_EXT_RENDER;                         // (the flag needs the renderer switched on)
    }
    *out = s;
    return HS_OK;
}

void hs_destroy(hs_sim *s) {
    if (!s) return;
    hipDeviceSynchronize();
    for (void *p : s->allocs) hipFree(p);
    for (auto &e : s->ev) if (e) hipEventDestroy(e);
    if (s->evIn) hipEventDestroy(s->evIn);
    if (s->stream) hipStreamDestroy(s->stream);
    for (auto &e : s->graph_exec) if (e) hipGraphExecDestroy(e);
    delete s;
}

int32_t hs_agents_per_world(const hs_sim *s) { return s ? s->A : 0; }

namespace {
// The handle's own stream starts after everything already queued on the device's legacy default stream: that is
// where torch (and scripts/benchmark.py:82-84) writes `action` / `reset` between steps.
int order_after_default_stream(hs_sim *s) {
    HS_HIP(hipEventRecord(s->evIn, nullptr));
    HS_HIP(hipStreamWaitEvent(s->stream, s->evIn, 0));
    return HS_OK;
}
}  // namespace

int32_t hs_init(hs_sim *s) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = order_after_default_stream(s);
    if (rc == HS_OK) rc = launch_step(s, s->stream, true);
    if (rc != HS_OK) return rc;
    HS_HIP(hipStreamSynchronize(s->stream));
    s->initialised = true;
    return HS_OK;
}

namespace {
// A blocking hs_step runs on the device's legacy default stream itself: it is ordered after the caller's writes to
// `action` / `reset` there without a cross-stream event (which costs tens of microseconds per step), and the caller
// waits for it anyway.  hs_step_begin / hs_step_end use the handle's own stream, so that the steps of several handles
// (one per GPU) run side by side.
int step_begin(hs_sim *s, bool own_stream) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    if (s->step_open) return fail(HS_ERR_INVALID_ARG, "hs_step_begin: the previous step was not ended");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = HS_OK;
    s->step_stream = own_stream ? s->stream : nullptr;
    if (own_stream) rc = order_after_default_stream(s);
    if (rc == HS_OK) rc = launch_step(s, s->step_stream, false);
    if (rc != HS_OK) return rc;
    s->step_open = true;
    return HS_OK;
}
}  // namespace

int32_t hs_step_begin(hs_sim *s) { return step_begin(s, true); }

int32_t hs_step_end(hs_sim *s) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    if (!s->step_open) return fail(HS_ERR_INVALID_ARG, "hs_step_end without hs_step_begin");
    s->step_open = false;
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipStreamSynchronize(s->step_stream));
    if (s->profiling) {
        HS_HIP(hipEventElapsedTime(&s->last_ms[0], s->ev[0], s->ev[1]));
        HS_HIP(hipEventElapsedTime(&s->last_ms[1], s->ev[1], s->ev[2]));
        HS_HIP(hipEventElapsedTime(&s->last_ms[2], s->ev[2], s->ev[3]));
    }
    return poll_status(s);
}

int32_t hs_step(hs_sim *s) {
    int rc = step_begin(s, s && s->blocking_own_stream);
    return rc != HS_OK ? rc : hs_step_end(s);
}

int32_t hs_step_async(hs_sim *s, void *hip_stream) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = poll_status(s);               // a failure of an earlier asynchronous step surfaces here
    return rc != HS_OK ? rc : launch_step(s, (hipStream_t)hip_stream, false);
}

int32_t hs_get_tensor(hs_sim *s, int32_t id, hs_tensor_desc *out) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    if (id < 0 || id >= HS_NUM_EXPORTS) return fail(HS_ERR_INVALID_ARG, "export id out of range");
    if (!s->exports[id].ptr) {
        // renderer outputs are allocated on first request; written only by hs_render / under HS_FLAG_EXT_RENDER
        if (id != HS_EXPORT_RGB && id != HS_EXPORT_DEPTH) return fail(HS_ERR_INVALID_ARG, "export not available");
        HS_HIP(hipSetDevice(s->cfg.gpu_id));
        const int rc = ensure_render_buffers(s);
        if (rc != HS_OK) return rc;
    }
    *out = s->exports[id];
    return HS_OK;
}

int32_t hs_render(hs_sim *s) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    if (!s->initialised) return fail(HS_ERR_INVALID_ARG, "hs_render before hs_init");
    if (s->step_open) return fail(HS_ERR_INVALID_ARG, "hs_render inside an open step");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = ensure_render_buffers(s);
    if (rc == HS_OK) rc = order_after_default_stream(s);
    if (rc != HS_OK) return rc;
    launch_render(s, s->stream);
    HS_HIP(hipGetLastError());
    HS_HIP(hipStreamSynchronize(s->stream));
    return HS_OK;
}

int32_t hs_trigger_reset(hs_sim *s, int32_t world, int32_t level) {
    if (!s || world < 0 || world >= s->S.N) return fail(HS_ERR_INVALID_ARG, "world index out of range");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipMemcpy(s->S.xReset + world, &level, sizeof(int32_t), hipMemcpyHostToDevice));
    return HS_OK;
}

int32_t hs_set_action(hs_sim *s, int32_t agent, int32_t x, int32_t y, int32_t r, int32_t g, int32_t l) {
    if (!s || agent < 0 || agent >= s->S.N * s->A) return fail(HS_ERR_INVALID_ARG, "agent index out of range");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int32_t a[5] = {x, y, r, g ? 1 : 0, l ? 1 : 0};
    HS_HIP(hipMemcpy(s->S.xAction + (size_t)agent * 5, a, sizeof(a), hipMemcpyHostToDevice));
    return HS_OK;
}

// ---- checkpoints (sim.cpp:956-1137, 1315-1333) ----
namespace {
int launch_save_ckpts(hs_sim *s, hipStream_t strm) {
    hipLaunchKernelGGL(hs::k_save_ckpt, dim3((s->S.N + 63) / 64), dim3(64), 0, strm, s->S);
    HS_HIP(hipGetLastError());
    return HS_OK;
}
int launch_load_ckpts(hs_sim *s, hipStream_t strm) {
    hipLaunchKernelGGL(hs::k_load_ckpt, dim3((s->S.N + 63) / 64), dim3(64), 0, strm, s->S);
    launch_observe(s, strm);              // postGenTasks + observationsTasks for every world (sim.cpp:1331-1332)
    HS_HIP(hipGetLastError());
    return HS_OK;
}
int set_ckpt_trigger(hs_sim *s, int32_t world) {
    if (world < 0 || world >= s->S.N) return fail(HS_ERR_INVALID_ARG, "world index out of range");
    const int32_t one = 1;
    HS_HIP(hipMemcpy(s->S.xCkptCtrl + world, &one, sizeof(one), hipMemcpyHostToDevice));
    return HS_OK;
}
}  // namespace

int32_t hs_save_checkpoints(hs_sim *s) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = order_after_default_stream(s);
    if (rc == HS_OK) rc = launch_save_ckpts(s, s->stream);
    if (rc != HS_OK) return rc;
    HS_HIP(hipStreamSynchronize(s->stream));
    return HS_OK;
}
int32_t hs_load_checkpoints(hs_sim *s) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = order_after_default_stream(s);
    if (rc == HS_OK) rc = launch_load_ckpts(s, s->stream);
    if (rc != HS_OK) return rc;
    HS_HIP(hipStreamSynchronize(s->stream));
    return HS_OK;
}
int32_t hs_save_checkpoint(hs_sim *s, int32_t world) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = set_ckpt_trigger(s, world);
    return rc != HS_OK ? rc : hs_save_checkpoints(s);
}
int32_t hs_load_checkpoint(hs_sim *s, int32_t world) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int rc = set_ckpt_trigger(s, world);
    return rc != HS_OK ? rc : hs_load_checkpoints(s);
}

// ---- stream entry points with the reference's JAX buffer order (mgr.cpp:168-201, 362-436) ----
namespace {
int copy_dd(void *dst, const void *src, size_t bytes, hipStream_t strm) {
    if (!dst || !src) return fail(HS_ERR_INVALID_ARG, "null device buffer");
    if (dst == src) return HS_OK;                // the caller passed the simulator's own tensor
    HS_HIP(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToDevice, strm));
    return HS_OK;
}
// copyOutObservations (mgr.cpp:340-360); returns the advanced buffer cursor through *pp
int copy_out_observations(hs_sim *s, hipStream_t strm, void ***pp) {
    const hs::SimState &S = s->S;
    const size_t R = (size_t)S.N * s->A;
    const struct { const void *src; size_t bytes; } obs[11] = {
        {S.xPrep, R * 4}, {S.xSelfObs, R * 13 * 4}, {S.xSelfType, R * 4}, {S.xSelfMask, R * 4}, {S.xLidar, R * 30 * 4},
        {S.xAgentObs, R * 5 * 14 * 4}, {S.xBoxObs, R * 9 * 17 * 4}, {S.xRampObs, R * 2 * 14 * 4},
        {S.xVisAgents, R * 5 * 4}, {S.xVisBoxes, R * 9 * 4}, {S.xVisRamps, R * 2 * 4}};
    for (const auto &o : obs) {
        int rc = copy_dd(*(*pp)++, o.src, o.bytes, strm);
        if (rc != HS_OK) return rc;
    }
    return HS_OK;
}
}  // namespace

int32_t hs_jax_init(hs_sim *s, void *hip_stream, void **buffers) {
    if (!s || !buffers) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    hipStream_t strm = (hipStream_t)hip_stream;
    int rc = launch_step(s, strm, true);
    if (rc == HS_OK) rc = copy_out_observations(s, strm, &buffers);
    if (rc != HS_OK) return rc;
    HS_HIP(hipStreamSynchronize(strm));          // gpuStreamInit synchronises (mgr.cpp:376)
    s->initialised = true;
    return HS_OK;
}
int32_t hs_jax_step(hs_sim *s, void *hip_stream, void **buffers) {
    if (!s || !buffers) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    hipStream_t strm = (hipStream_t)hip_stream;
    const hs::SimState &S = s->S;
    const size_t N = (size_t)S.N, R = N * s->A;
    int rc = poll_status(s);               // a failure of an earlier asynchronous step surfaces here
    if (rc == HS_OK) rc = copy_dd(S.xAction, *buffers++, R * 5 * 4, strm);
    if (rc == HS_OK) rc = copy_dd(S.xReset, *buffers++, N * 4, strm);
    if (rc == HS_OK) rc = copy_dd(S.xPolicy, *buffers++, R * 4, strm);
    if (rc == HS_OK) rc = launch_step(s, strm, false);
    if (rc == HS_OK) rc = copy_out_observations(s, strm, &buffers);
    if (rc == HS_OK) rc = copy_dd(*buffers++, S.xReward, R * 4, strm);
    if (rc == HS_OK) rc = copy_dd(*buffers++, S.xDone, R * 4, strm);
    if (rc == HS_OK) rc = copy_dd(*buffers++, S.xEpisodeResult, N * 2 * 4, strm);
    return rc;
}
int32_t hs_jax_save_checkpoints(hs_sim *s, void *hip_stream, void **buffers) {
    if (!s || !buffers) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    hipStream_t strm = (hipStream_t)hip_stream;
    const size_t N = (size_t)s->S.N;
    int rc = copy_dd(s->S.xCkptCtrl, buffers[0], N * 4, strm);
    if (rc == HS_OK) rc = launch_save_ckpts(s, strm);
    if (rc == HS_OK) rc = copy_dd(buffers[1], s->S.xCkpt, N * sizeof(hs_checkpoint), strm);
    return rc;
}
int32_t hs_jax_load_checkpoints(hs_sim *s, void *hip_stream, void **buffers) {
    if (!s || !buffers) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    hipStream_t strm = (hipStream_t)hip_stream;
    const size_t N = (size_t)s->S.N;
    int rc = copy_dd(s->S.xCkptCtrl, *buffers++, N * 4, strm);
    if (rc == HS_OK) rc = copy_dd(s->S.xCkpt, *buffers++, N * sizeof(hs_checkpoint), strm);
    if (rc == HS_OK) rc = launch_load_ckpts(s, strm);
    if (rc == HS_OK) rc = copy_out_observations(s, strm, &buffers);
    return rc;
}

// ---- XLA custom-call targets (the original, status-less GPU custom-call ABI) ----
// What madrona::py::JAXInterface registers with XLA for the four Manager functions above (src/bindings.cpp:97-118):
// XLA calls target(stream, buffers, opaque, opaque_len) from its own stream-executor thread; `opaque` is the
// descriptor the Python side attached to the call — here the 8 bytes of the simulator handle.  The ABI has no status
// channel: a failure is kept (hs_xla_last_status, hs_last_error on the calling thread is not the user's thread) and
// surfaces as HS_ERR_HIP from the next blocking call on the handle, like a failed asynchronous step.
namespace {
std::atomic<int32_t> g_xla_status{HS_OK};
void xla_call(int32_t (*FN)(hs_sim *, void *, void **), void *stream, void **buffers, const char *opaque, size_t opaque_len) {
    hs_sim *s = nullptr;
    if (opaque && opaque_len == sizeof(s)) memcpy(&s, opaque, sizeof(s));
    const int32_t rc = s ? FN(s, stream, buffers) : (int32_t)HS_ERR_INVALID_ARG;
    if (rc != HS_OK) { g_xla_status.store(rc); if (s) s->async_error.store(rc); }
}
}  // namespace
void hs_xla_init(void *stream, void **buffers, const char *opaque, size_t opaque_len) { xla_call(hs_jax_init, stream, buffers, opaque, opaque_len); }
void hs_xla_step(void *stream, void **buffers, const char *opaque, size_t opaque_len) { xla_call(hs_jax_step, stream, buffers, opaque, opaque_len); }
void hs_xla_save_checkpoints(void *stream, void **buffers, const char *opaque, size_t opaque_len) { xla_call(hs_jax_save_checkpoints, stream, buffers, opaque, opaque_len); }
void hs_xla_load_checkpoints(void *stream, void **buffers, const char *opaque, size_t opaque_len) { xla_call(hs_jax_load_checkpoints, stream, buffers, opaque, opaque_len); }
int32_t hs_xla_last_status(int32_t clear) { return clear ? g_xla_status.exchange(HS_OK) : g_xla_status.load(); }

// Development aid (HS_PHASE_TIMING builds): accumulated wall-clock ticks per phase per workgroup of k_physics.
int32_t hs_debug_phase_ticks(hs_sim *s, int64_t *out, int32_t max_groups) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    int nb = (s->S.N + hs::kTile - 1) / hs::kTile * (hs::kTile / s->tile);          // one workgroup (wave) per tile
    if (nb > max_groups) nb = max_groups;
    HS_HIP(hipMemcpy(out, s->S.phaseTicks, (size_t)nb * 10 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return nb;
}

// Per-world work counters (HS_LOAD_STUDY builds): [worlds][8], see hs_state.h phase_ticks_study_base.
int32_t hs_debug_load_study(hs_sim *s, int64_t *out, int32_t max_worlds) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    const int n = s->S.N < max_worlds ? s->S.N : max_worlds;
    HS_HIP(hipMemcpy(out, s->S.phaseTicks + hs::phase_ticks_study_base(s->S.N), (size_t)n * hs::kStudyWords * sizeof(int64_t), hipMemcpyDeviceToHost));
    return n;
}

// The same for k_observe: ticks per section summed over all waves (HS_PHASE_TIMING builds).
int32_t hs_debug_observe_ticks(hs_sim *s, int64_t out[16]) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    std::vector<int64_t> part(16 * 1024);
    HS_HIP(hipMemcpy(part.data(), s->S.phaseTicks + hs::phase_ticks_obs_base(s->S.N), part.size() * sizeof(int64_t),
                     hipMemcpyDeviceToHost));
    for (int i = 0; i < 16; ++i) { out[i] = 0; for (int b = 0; b < 1024; ++b) out[i] += part[(size_t)b * 16 + i]; }
    return HS_OK;
}
// ... and the work counters of the convex tests: calls, box items, wedge items, rounds, colliding pairs, contact rounds.
int32_t hs_debug_sat_counters(hs_sim *s, int64_t out[16]) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipMemcpy(out, s->S.phaseTicks + hs::phase_ticks_obs_base(s->S.N) + 16 * 1024, 16 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return HS_OK;
}

int32_t hs_debug_dump_bodies(hs_sim *s, float *bodies, int32_t *meta) {
    if (!s || !bodies || !meta) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipDeviceSynchronize());
    const size_t N = s->S.N, D = hs::kNumDSlots;
    HostCol<float, 3 * hs::kNumDSlots> pos, lin, ang; HostCol<float, 4 * hs::kNumDSlots> rot; HostCol<int, hs::kNumDSlots> m;
    int rc;
    if ((rc = pos.load(s->S.bpos, N)) != HS_OK || (rc = rot.load(s->S.brot, N)) != HS_OK || (rc = lin.load(s->S.blin, N)) != HS_OK ||
        (rc = ang.load(s->S.bang, N)) != HS_OK || (rc = m.load(s->S.bmeta, N)) != HS_OK) return rc;
    std::vector<int32_t> slot;
    if ((rc = load_slots(s, slot)) != HS_OK) return rc;
    for (size_t w = 0; w < N; ++w)
        for (size_t i = 0; i < D; ++i) {
            const size_t p = (size_t)slot[w];
            float *o = bodies + (w * D + i) * 13;
            for (size_t c = 0; c < 3; ++c) { o[c] = pos(c * D + i, p); o[7 + c] = lin(c * D + i, p); o[10 + c] = ang(c * D + i, p); }
            for (size_t c = 0; c < 4; ++c) o[3 + c] = rot(c * D + i, p);
            int32_t mm = m(i, p);
            int32_t *om = meta + (w * D + i) * 3;
            if (mm == 0) { om[0] = -1; om[1] = 2; om[2] = 0; }
            else { om[0] = (mm & 0xff) - 1; om[1] = (mm >> 8) & 0xff; om[2] = (mm >> 16) & 0xff; }
        }
    return HS_OK;
}

int32_t hs_debug_dump_walls(hs_sim *s, float *walls, int32_t *info) {
    if (!s || !walls || !info) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipDeviceSynchronize());
    const size_t N = s->S.N, K = hs::kMaxWalls;
    HostCol<float, 4 * hs::kMaxWalls> wl;
    int rc;
    if ((rc = wl.load(s->S.walls, N)) != HS_OK) return rc;
    std::vector<int32_t> nw(N), np(N), cnt(N), step(N);
    HS_HIP(hipMemcpy(nw.data(), s->S.numWalls, N * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(np.data(), s->S.numPlanes, N * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(cnt.data(), s->S.counts, N * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(step.data(), s->S.curEpisodeStep, N * 4, hipMemcpyDeviceToHost));
    std::vector<int32_t> slot;
    if ((rc = load_slots(s, slot)) != HS_OK) return rc;
    for (size_t w = 0; w < N; ++w) {
        for (size_t k = 0; k < K; ++k)
            for (size_t c = 0; c < 4; ++c)
                walls[(w * K + k) * 4 + c] = (int)k < nw[w] ? wl(c * K + k, (size_t)slot[w]) : 0.f;
        int32_t *m = info + w * 8;
        const int c = cnt[w];
        m[0] = nw[w]; m[1] = np[w]; m[2] = (c >> 12) & 15; m[3] = (c >> 16) & 15; m[4] = c & 15; m[5] = (c >> 4) & 15;
        m[6] = step[w]; m[7] = (c >> 20) & 1;
    }
    return HS_OK;
}

// PMC calibration (MI355X_MICROARCH.md: FETCH_SIZE is uncalibrated for narrow accesses): copies `bytes`
// with the access pattern of the simulator's SoA columns — one coalesced dword per lane.
__global__ void k_calib_copy_dword(const float *in, float *out, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = in[i] + 1.f;
}
int32_t hs_debug_calibrate(int64_t bytes) {
    if (bytes <= 0) return fail(HS_ERR_INVALID_ARG, "bytes must be > 0");
    float *a = nullptr, *b = nullptr;
    HS_HIP(hipMalloc((void **)&a, (size_t)bytes));
    HS_HIP(hipMalloc((void **)&b, (size_t)bytes));
    HS_HIP(hipMemset(a, 0, (size_t)bytes));
    hipLaunchKernelGGL(k_calib_copy_dword, dim3(4096), dim3(256), 0, nullptr, a, b, (size_t)bytes / 4);
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipFree(a)); HS_HIP(hipFree(b));
    return HS_OK;
}

// hs_debug_dump_hull: the hull tables as the kernels see them (hs_collide.h), one lane.
__global__ void k_dump_hull(int obj, float *verts, int *faces, int *counts, float *normals, int *edges, float *local) {
    using namespace hs;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const HullRef h = obj == OBJ_WALL ? hull_ref_wall(0.f, 0.f, 1.f, 1.f) : hull_ref_body(obj, V3{0.f, 0.f, 0.f}, Q{1.f, 0.f, 0.f, 0.f});
    const int nv = hull_nv(h), nf = hull_nf(h), ne = hull_ne(h);
    counts[0] = nv; counts[1] = nf; counts[2] = ne; counts[3] = hull_ned(h);
    for (int i = 0; i < nv; ++i) {
        const V3 v = hull_v(h, i);
        verts[i * 3] = v.x; verts[i * 3 + 1] = v.y; verts[i * 3 + 2] = v.z;
        const V3 l = obj == OBJ_WALL ? V3{0.f, 0.f, 0.f} : hull_local_vertex(obj, i);
        local[i * 3] = l.x; local[i * 3 + 1] = l.y; local[i * 3 + 2] = l.z;
    }
    for (int f = 0; f < nf; ++f) {
        for (int k = 0; k < 4; ++k) faces[f * 4 + k] = k < hull_fcnt(h, f) ? hull_fidx(h, f, k) : -1;
        const V3 n = hull_fn(h, f);
        normals[f * 3] = n.x; normals[f * 3 + 1] = n.y; normals[f * 3 + 2] = n.z;
    }
    for (int e = 0; e < ne; ++e) { int v0, v1, d; hull_edge(h, e, &v0, &v1, &d); edges[e * 3] = v0; edges[e * 3 + 1] = v1; edges[e * 3 + 2] = d; }
}
int32_t hs_debug_dump_hull(int32_t obj, float *verts, int32_t *faces, int32_t *counts, float *normals, int32_t *edges, float *local) {
    if (!verts || !faces || !counts || !normals || !edges || !local) return fail(HS_ERR_INVALID_ARG, "null argument");
    if (obj != hs::OBJ_CUBE && obj != hs::OBJ_WALL && obj != hs::OBJ_HIDER && obj != hs::OBJ_SEEKER && obj != hs::OBJ_RAMP && obj != hs::OBJ_BOX)
        return fail(HS_ERR_INVALID_ARG, "no collision hull for this SimObject");
    char *d = nullptr;
    const size_t nb = (24 + 24 + 8 + 18 + 36 + 24) * 4;
    HS_HIP(hipMalloc((void **)&d, nb));
    HS_HIP(hipMemset(d, 0, nb));
    float *dv = (float *)d; int *df = (int *)(dv + 24), *dc = df + 24; float *dn = (float *)(dc + 8); int *de = (int *)(dn + 18); float *dl = (float *)(de + 36);
    hipLaunchKernelGGL(k_dump_hull, dim3(1), dim3(64), 0, nullptr, (int)obj, dv, df, dc, dn, de, dl);
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(verts, dv, 24 * 4, hipMemcpyDeviceToHost)); HS_HIP(hipMemcpy(faces, df, 24 * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(counts, dc, 4 * 4, hipMemcpyDeviceToHost)); HS_HIP(hipMemcpy(normals, dn, 18 * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipMemcpy(edges, de, 36 * 4, hipMemcpyDeviceToHost)); HS_HIP(hipMemcpy(local, dl, 24 * 4, hipMemcpyDeviceToHost));
    HS_HIP(hipFree(d));
    return HS_OK;
}

__global__ void k_object_params(int obj, float *out) {
    using namespace hs;
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const V3 i = obj_inv_inertia(obj);
    out[0] = obj_inv_mass(obj); out[1] = obj_mu_s(obj); out[2] = obj_mu_d(obj); out[3] = i.x; out[4] = i.y; out[5] = i.z;
}
int32_t hs_debug_object_params(int32_t obj, float *out) {
    if (!out || obj < 0 || obj > hs::OBJ_BOX) return fail(HS_ERR_INVALID_ARG, "bad argument");
    float *d = nullptr;
    HS_HIP(hipMalloc((void **)&d, 6 * sizeof(float)));
    hipLaunchKernelGGL(k_object_params, dim3(1), dim3(64), 0, nullptr, (int)obj, d);
    HS_HIP(hipDeviceSynchronize());
    HS_HIP(hipMemcpy(out, d, 6 * sizeof(float), hipMemcpyDeviceToHost));
    HS_HIP(hipFree(d));
    return HS_OK;
}

// DLPack deleter for the non-owning tensor views handed to Python: the simulator owns the memory, the
// binding keeps the DLManagedTensor records alive, so there is nothing to free (and nothing here may call
// back into an interpreter that is shutting down).
void hs_dlpack_noop_deleter(void *) {}

int32_t hs_train_interface(const hs_iface_entry **entries) {
    // Manager::trainInterface (mgr.cpp:1338-1375), in its order
    static const hs_iface_entry table[] = {
        {"actions", HS_ROLE_ACTION, HS_EXPORT_ACTION},
        {"resets", HS_ROLE_RESET, HS_EXPORT_RESET},
        {"sim_ctrl", HS_ROLE_SIM_CTRL, -1},
        {"policy_assignments", HS_ROLE_PBT_INPUT, HS_EXPORT_AGENT_POLICY},
        {"prep_counter", HS_ROLE_OBSERVATION, HS_EXPORT_PREP_COUNTER},
        {"self_data", HS_ROLE_OBSERVATION, HS_EXPORT_SELF_OBS},
        {"self_type", HS_ROLE_OBSERVATION, HS_EXPORT_SELF_TYPE},
        {"self_mask", HS_ROLE_OBSERVATION, HS_EXPORT_SELF_MASK},
        {"self_lidar", HS_ROLE_OBSERVATION, HS_EXPORT_LIDAR},
        {"agent_data", HS_ROLE_OBSERVATION, HS_EXPORT_AGENT_OBS},
        {"box_data", HS_ROLE_OBSERVATION, HS_EXPORT_BOX_OBS},
        {"ramp_data", HS_ROLE_OBSERVATION, HS_EXPORT_RAMP_OBS},
        {"vis_agents_mask", HS_ROLE_OBSERVATION, HS_EXPORT_AGENT_VIS_MASKS},
        {"vis_boxes_mask", HS_ROLE_OBSERVATION, HS_EXPORT_BOX_VIS_MASKS},
        {"vis_ramps_mask", HS_ROLE_OBSERVATION, HS_EXPORT_RAMP_VIS_MASKS},
        {"rewards", HS_ROLE_REWARD, HS_EXPORT_REWARD},
        {"dones", HS_ROLE_DONE, HS_EXPORT_DONE},
        {"episode_results", HS_ROLE_PBT_OUTPUT, HS_EXPORT_EPISODE_RESULT},
        {"checkpoint_data", HS_ROLE_CHECKPOINT, HS_EXPORT_CHECKPOINT},
    };
    if (entries) *entries = table;
    return (int32_t)(sizeof(table) / sizeof(table[0]));
}

int32_t hs_get_device_status(hs_sim *s, hs_device_status *out) {
    if (!s || !out) return fail(HS_ERR_INVALID_ARG, "null argument");
    HS_HIP(hipSetDevice(s->cfg.gpu_id));
    HS_HIP(hipDeviceSynchronize());
    int st[4];
    HS_HIP(hipMemcpy(st, s->S.status, sizeof(st), hipMemcpyDeviceToHost));
    out->spilled_dd_pairs = st[0];
    out->spilled_static_pairs = st[1];
    out->dropped_dd_pairs = 0;             // nothing can overflow the spill lists (hs_k_physics.h: sized for every pair)
    out->dropped_static_pairs = 0;
    out->graphs_in_use = (s->use_graph && s->graph_exec[0]) ? 1 : 0;
    out->reserved = 0;
    return HS_OK;
}

int32_t hs_set_profiling(hs_sim *s, int32_t enabled) {
    if (!s) return fail(HS_ERR_INVALID_ARG, "null sim");
    s->profiling = enabled != 0;
    return HS_OK;
}

int32_t hs_last_step_kernel_ms(hs_sim *s, float out_ms[3]) {
    if (!s || !out_ms) return fail(HS_ERR_INVALID_ARG, "null argument");
    for (int i = 0; i < 3; ++i) out_ms[i] = s->last_ms[i];
    return HS_OK;
}

}  // extern "C"
