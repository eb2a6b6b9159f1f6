"""What a physics wave's time depends on, and how well the past predicts it (development aid):
  python tools/load_study.py [worlds] [steps]      (builds its own library with -DHS_PHASE_TIMING -DHS_LOAD_STUDY)
Per step: the waves' times (phase ticks) and per-world work counters (hs_debug_load_study).  Prints (1) a linear cost
model of a wave's time in its worlds' counters, (2) how much of the persistent spread between waves it explains,
(3) how well a world's counters over one deal period predict the next period's."""
import os, sys, ctypes as C, numpy as np
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))
import build
lib = build.build_lib(out=os.path.join(build.HERE, "lib", "libhideseek_study.so"), defines=("HS_PHASE_TIMING", "HS_LOAD_STUDY"))
os.environ["HS_LIB_PATH"] = lib
import torch, gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 256
period = int(os.environ.get("HS_BALANCE_PERIOD", "32"))
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
    min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
act = sim.action_tensor().to_torch()
sim.init()
nb = (N + 7) // 8
L = sim._L
L.hs_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
L.hs_debug_load_study.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
pt = np.zeros((nb, 10), np.int64); st = np.zeros((N, 8), np.int64)
L.hs_debug_phase_ticks(sim._h, pt.ctypes.data, nb); L.hs_debug_load_study(sim._h, st.ctypes.data, N)
ppt, pst = pt.copy(), st.copy()
T = np.zeros((steps, nb)); DD = np.zeros((steps, nb)); F = np.zeros((steps, N, 4), np.int32); O = np.zeros((steps, N), np.int32); WR = np.zeros((steps, nb))
for i in range(steps):
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    sim.step()
    L.hs_debug_phase_ticks(sim._h, pt.ctypes.data, nb); L.hs_debug_load_study(sim._h, st.ctypes.data, N)
    d = (pt - ppt) / 100.0
    T[i] = d.sum(axis=1); DD[i] = d[:, 4] + d[:, 6]
    F[i] = (st - pst)[:, :4]; O[i] = st[:, 4]
    np.add.at(WR[i], st[:, 4], (st - pst)[:, 5])
    ppt[:] = pt; pst[:] = st
ok = T.max(axis=1) < 1500
def octsum(i, k): return np.bincount(O[i], weights=F[i, :, k], minlength=nb)
def octmax(i, k):
    m = np.zeros(nb); np.maximum.at(m, O[i], F[i, :, k]); return m
# (1) instantaneous cost model across waves and steps
rows, ys, ydd = [], [], []
for i in np.nonzero(ok)[0][period:]:
    rows.append(np.stack([np.ones(nb), octsum(i, 0), octsum(i, 1), octsum(i, 2), WR[i], octmax(i, 3)], axis=1)); ys.append(T[i]); ydd.append(DD[i])
X = np.concatenate(rows); y = np.concatenate(ys); ydd = np.concatenate(ydd)
names = ["const", "dd candidates", "static candidates", "dd manifolds", "wave's dd rounds", "max world dd rounds"]
def fit(cols, yy, label):
    A = X[:, cols]; c, *_ = np.linalg.lstsq(A, yy, rcond=None); r = yy - A @ c
    print(f"{label}: R2 {1 - r.var() / yy.var():.3f}  " + ", ".join(f"{names[k]} {v:.2f}" for k, v in zip(cols, c)))
    return c
print(f"{ok.sum()} steps, {nb} waves; mean wave {y.mean():.1f} us, sd {y.std():.1f}; per wave and step: dd cand {X[:,1].mean():.1f}, static cand {X[:,2].mean():.1f}, dd manifolds {X[:,3].mean():.1f}, dd rounds {X[:,4].mean():.1f}")
fit([0, 1, 2], y, "time ~ candidates            ")
fit([0, 1, 2, 3], y, "time ~ cand + manifolds      ")
cm = fit([0, 1, 2, 3, 4], y, "time ~ cand + manif + rounds ")
fit([0, 3, 4], ydd, "dd phases ~ manifolds, rounds")
fit([0, 4], ydd, "dd phases ~ rounds           ")
# (2) persistent part per period
for cols, label in (([0, 1, 2], "candidates"), ([0, 1, 2, 3, 4], "cand + manifolds + rounds")):
    num = den = 0.0
    for s0 in range(period, steps - period + 1, period):
        sel = [i for i in range(s0, s0 + period) if ok[i]]
        Tm = np.mean([T[i] - T[i].mean() for i in sel], axis=0)
        A = np.mean([np.stack([np.ones(nb), octsum(i, 0), octsum(i, 1), octsum(i, 2), WR[i], octmax(i, 3)], axis=1) for i in sel], axis=0)[:, cols]
        c, *_ = np.linalg.lstsq(A, Tm, rcond=None)
        num += (Tm - A @ c).var(); den += Tm.var()
    print(f"persistent spread over a period explained by the period's {label}: {1 - num / den:.2f}")
# (3) world-level persistence from one period to the next
per = [np.arange(s0, s0 + period) for s0 in range(period, steps - period + 1, period)]
W = np.stack([F[p].sum(axis=0) for p in per]).astype(float)           # [period, world, 4]
for k, nm in enumerate(["dd candidates", "static candidates", "dd manifolds", "dd rounds pending"]):
    c = [np.corrcoef(W[j, :, k], W[j + 1, :, k])[0, 1] for j in range(len(per) - 1)]
    print(f"world's {nm} over a period vs the next period: corr {np.mean(c):.2f}; mean {W[:, :, k].mean():.1f}, sd over worlds {W[:, :, k].std(axis=1).mean():.1f}")
np.savez_compressed(os.path.join(ROOT, "gpurun_out", "load_study.npz"), T=T.astype(np.float32), F=F.astype(np.int16), O=O.astype(np.int16), WR=WR.astype(np.int16))
