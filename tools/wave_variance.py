"""Where the spread between the physics waves of a step comes from (development aid; needs the HS_PHASE_TIMING build):
  HS_LIB_PATH=$PWD/marl-hideandseek_amd/lib/libhideseek_timing.so python tools/wave_variance.py [worlds] [steps]
Per step and wave the phase ticks are differenced; the spread of the waves' totals is split into the part that persists
over a deal period (32 steps: what a better load estimate could remove) and the part that changes from step to step."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "marl-hideandseek_amd"))
import gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 240
period = int(os.environ.get("HS_BALANCE_PERIOD", "32"))
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
    min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
act = sim.action_tensor().to_torch()
sim.init()
nb = (N + 7) // 8
L = sim._L
L.hs_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
prev = np.zeros((nb, 10), np.int64); cur = np.zeros((nb, 10), np.int64)
L.hs_debug_phase_ticks(sim._h, prev.ctypes.data, nb)
names = ["pre", "integrate", "detect", "sat", "dd_pos", "body_pos", "dd_vel", "body_vel", "post+store", "load"]
D = np.zeros((steps, nb, 10))
for i in range(steps):
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    sim.step()
    L.hs_debug_phase_ticks(sim._h, cur.ctypes.data, nb)
    D[i] = (cur - prev) / 100.0
    prev[:] = cur
T = D.sum(axis=2)                                     # [step, wave] us
ok = T.max(axis=1) < 1500                             # (not the step on which all worlds regenerate)
print(f"{ok.sum()} of {steps} steps; mean wave {T[ok].mean():.1f} us, slowest {T[ok].max(axis=1).mean():.1f}, sd over waves {T[ok].std(axis=1).mean():.1f}")
# deal periods: the k-th deal happens before step k * period (counted from init)
pers, resid, tot = [], [], []
pcov = np.zeros(10); pvar = 0.0
for s0 in range(period, steps - period + 1, period):
    sel = np.arange(s0, s0 + period)[ok[s0:s0 + period]]
    if len(sel) < period // 2: continue
    X = T[sel] - T[sel].mean(axis=1, keepdims=True)    # deviation from the step's mean wave
    m = X.mean(axis=0)                                 # a wave's persistent deviation over the period
    pers.append(m.var()); resid.append((X - m).var()); tot.append(X.var())
    Pm = (D[sel] - D[sel].mean(axis=1, keepdims=True)).mean(axis=0)          # [wave, phase] persistent deviation per phase
    pcov += (Pm * m[:, None]).mean(axis=0); pvar += m.var()
print(f"variance of a wave's deviation from the step mean: total {np.mean(tot):.0f} us^2 = persistent over a period {np.mean(pers):.0f} + step-to-step {np.mean(resid):.0f}")
print("share of the persistent part by phase: " + ", ".join(f"{nm} {pcov[i] / pvar:.2f}" for i, nm in enumerate(names)))
X = T[ok] - T[ok].mean(axis=1, keepdims=True)
print("phase        mean     sd over waves   covariance with the wave's total / variance of the total")
for i, nm in enumerate(names):
    P = D[ok][:, :, i]; Pc = P - P.mean(axis=1, keepdims=True)
    print(f"{nm:10s} {P.mean():7.1f} {Pc.std():10.1f} {((Pc * X).mean() / X.var()):18.2f}")
# step-to-step correlation of a wave's deviation (inside deal periods)
c = []
for s in range(period, steps - 1):
    if (s + 1) % period == 0 or not (ok[s] and ok[s + 1]): continue
    a = T[s] - T[s].mean(); b = T[s + 1] - T[s + 1].mean()
    c.append((a * b).mean() / (a.std() * b.std()))
print(f"correlation of a wave's deviation between consecutive steps: {np.mean(c):.2f}")
for lag in (4, 16):
    c = []
    for s in range(period, steps - lag):
        if s // period != (s + lag) // period or not (ok[s] and ok[s + lag]): continue
        a = T[s] - T[s].mean(); b = T[s + lag] - T[s + lag].mean()
        c.append((a * b).mean() / (a.std() * b.std()))
    print(f"  at lag {lag}: {np.mean(c):.2f}")
