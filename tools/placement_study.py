"""Where the dispatcher puts the physics waves (development aid; HS_LOAD_STUDY build): which workgroups share a SIMD, and
whether the placement repeats from launch to launch.   python tools/placement_study.py [worlds] [steps]"""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))
import build
lib = build.build_lib(out=os.path.join(build.HERE, "lib", "libhideseek_study.so"), defines=("HS_PHASE_TIMING", "HS_LOAD_STUDY"))
os.environ["HS_LIB_PATH"] = lib
import torch, gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 40
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
    min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
act = sim.action_tensor().to_torch()
sim.init()
nb = (N + 7) // 8
L = sim._L
L.hs_debug_load_study.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
L.hs_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
st = np.zeros((N, 8), np.int64); pt = np.zeros((nb, 10), np.int64); ppt = pt.copy()
places = []; times = []
for i in range(steps):
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    sim.step()
    L.hs_debug_load_study(sim._h, st.ctypes.data, N); L.hs_debug_phase_ticks(sim._h, pt.ctypes.data, nb)
    first = st[:, 6] != 0
    blk = st[first, 4]; hw = st[first, 6]; xcc = st[first, 7] & 15
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7; wv = hw & 15
    place = np.full(nb, -1, np.int64)
    place[blk] = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
    places.append(place); times.append(((pt - ppt) / 100.0).sum(axis=1)); ppt[:] = pt
    st[:, 6] = 0
    if i == steps - 1:
        print("last step: workgroup -> xcc se sh cu simd wave")
        o = np.argsort(blk)
        for k in list(range(0, 24)) + list(range(1020, 1030)) + list(range(1990, 2000)):
            j = o[k]; print(f"  wg {blk[j]:5d}: xcc {xcc[j]} se {se[j]} sh {sh[j]} cu {cu[j]:2d} simd {simd[j]} wave {wv[j]}")
P = np.stack(places); T = np.stack(times)
print(f"{steps} launches, {nb} workgroups; distinct SIMDs used in the last launch: {len(np.unique(P[-1]))}")
cnt = np.bincount(np.unique(P[-1], return_inverse=True)[1]); print("waves per used SIMD: " + ", ".join(f"{k}: {(cnt == k).sum()}" for k in range(1, cnt.max() + 1)))
same = [(P[i] == P[i + 1]).mean() for i in range(5, steps - 1)]
print(f"share of workgroups on the same SIMD as in the previous launch: {np.mean(same):.3f}")
# partner of a workgroup: the other workgroup on its SIMD
def partners(p):
    o = np.argsort(p, kind="stable"); ps = p[o]; part = np.full(nb, -1)
    eq = ps[1:] == ps[:-1]
    part[o[1:][eq]] = o[:-1][eq]; part[o[:-1][eq]] = o[1:][eq]
    return part
pa = partners(P[-1]); pb = partners(P[-2])
print(f"share of workgroups with the same partner as in the previous launch: {(pa == pb).mean():.3f}; without a partner: {(pa < 0).mean():.3f}")
d = pa - np.arange(nb); v, c = np.unique(d[pa >= 0], return_counts=True); top = np.argsort(-c)[:8]
print("partner - own workgroup index, most frequent: " + ", ".join(f"{v[k]}: {c[k]}" for k in top))
# does a wave's time depend on its partner's?
t = T[-1]; ok = pa >= 0
print(f"corr(time, partner's time) {np.corrcoef(t[ok], t[pa[ok]])[0, 1]:.2f}; mean time with partner {t[ok].mean():.1f}, alone {t[~ok].mean() if (~ok).any() else float('nan'):.1f}")
