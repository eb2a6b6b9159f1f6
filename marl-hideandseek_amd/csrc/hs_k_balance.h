// Load balancing between the physics waves.
//
// k_physics gives one wave to an octet of 8 worlds and ends with its slowest wave; an octet's time follows the number
// of contact candidates its worlds hold (1.8 x between the quietest and the busiest octet of a random assignment),
// and that number persists for most of an episode.  Every kBalancePeriod steps the worlds are therefore dealt to the
// octets like cards: sorted by the candidate pairs they showed since the last deal (SimState::loadAcc, counted by the
// broadphase), rank r goes to octet r mod #octets — in alternating direction from row to row — so every octet gets
// one world of every load class.  Moving a world = moving its rows of the tiled columns to another slot
// (slotOfWorld / worldOfSlot, hs_state.h); per-world scalars and the exported tensors are indexed by world id and
// stay where they are.  Worlds do not interact, so WHICH octet a world lives in changes no result — only the time.
#pragma once
#include "hs_state.h"
#include "hs_k_reset.h"

namespace hs {

constexpr int kBalanceBins = 1024;

// 1. histogram of the load classes (descending: class 0 = busiest)
HSD int load_class(int load) { const int c = load >> 2; return kBalanceBins - 1 - (c < kBalanceBins - 1 ? c : kBalanceBins - 1); }
__global__ void __launch_bounds__(256) k_balance_hist(SimState S, int nfull, int *hist) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w < nfull) atomicAdd(&hist[load_class(S.loadAcc[w])], 1);
}
// 2. exclusive prefix over the classes (one workgroup), cursors start at the class bases
__global__ void __launch_bounds__(kBalanceBins) k_balance_scan(int *hist, int *cursor) {
    __shared__ int sh[kBalanceBins];
    const int t = threadIdx.x;
    sh[t] = hist[t];
    __syncthreads();
    for (int d = 1; d < kBalanceBins; d <<= 1) {
        const int v = t >= d ? sh[t - d] : 0;
        __syncthreads();
        sh[t] += v;
        __syncthreads();
    }
    cursor[t] = sh[t] - hist[t];
    hist[t] = 0;
}
// 3. rank of every world (order inside a class does not matter), its new slot; the load counters start over
// (`tile`: worlds per physics wave, 8 or 4 — the groups the load is equalised over)
__global__ void __launch_bounds__(256) k_balance_deal(SimState S, int nfull, int *cursor, int *newSlot, int tile) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nfull) return;
    const int r = atomicAdd(&cursor[load_class(S.loadAcc[w])], 1);
    const int noct = nfull / tile;
    const int row = r / noct, col = r - row * noct;
    newSlot[w] = ((row & 1) ? noct - 1 - col : col) * tile + row;
    S.loadAcc[w] = 0;
}
// 4. every row of every tiled column moves from the old slot to the new one, in ONE launch: the columns are consecutive
// pieces of one arena (hideseek.hip), `src` is a copy of the arena, `base` the first arena row of each column (+ the total).
constexpr int kBalanceCols = 11;
struct BalanceCols { int base[kBalanceCols + 1]; };
__global__ void __launch_bounds__(256) k_balance_move_all(int *dst, const int *src, BalanceCols cols, size_t slots, const int *oldSlot,
                                                          const int *newSlot, int nfull) {
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;          // (arena row, world) with the world fastest
    if (i >= (size_t)nfull * cols.base[kBalanceCols]) return;
    const int grow = (int)(i / nfull), w = (int)(i - (size_t)grow * nfull);
    int c = 0;
#pragma unroll
    for (int k = 1; k < kBalanceCols; ++k) c += grow >= cols.base[k] ? 1 : 0;
    const int row = grow - cols.base[c], rows = cols.base[c + 1] - cols.base[c];
    const size_t off = (size_t)cols.base[c] * slots;                           // the column's first element in the arena
    const int a = oldSlot[w], b = newSlot[w];
    dst[off + ((size_t)(b >> 3) * rows + row) * kTile + (b & 7)] = src[off + ((size_t)(a >> 3) * rows + row) * kTile + (a & 7)];
}
// 5. the maps
__global__ void __launch_bounds__(256) k_balance_commit(SimState S, int nfull, const int *newSlot) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= nfull) return;
    const int b = newSlot[w];
    S.slotOfWorld[w] = b;
    S.worldOfSlot[b] = w;
    write_slot_hdr(S, w, b);
}

}  // namespace hs
