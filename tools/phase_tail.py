"""What the slowest physics wave of a step spends its time on (development aid; needs the HS_PHASE_TIMING build):
  HS_LIB_PATH=$PWD/marl-hideandseek_amd/lib/libhideseek_timing.so python tools/phase_tail.py [worlds] [steps]
Per step the per-wave phase ticks are differenced; the table compares the mean wave with the slowest one."""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "marl-hideandseek_amd"))
import gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 240
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
    min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
act = sim.action_tensor().to_torch()
sim.init()
nb = (N + 7) // 8
L = sim._L
L.hs_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
prev = np.zeros((nb, 10), np.int64)
cur = np.zeros((nb, 10), np.int64)
L.hs_debug_phase_ticks(sim._h, prev.ctypes.data, nb)
names = ["pre", "integrate", "detect", "sat", "dd_pos", "body_pos", "dd_vel", "body_vel", "post+store", "load"]
mean_acc = np.zeros(10); max_acc = np.zeros(10); p99_acc = np.zeros(10); ratios = []; used = 0
for i in range(steps):
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    sim.step()
    L.hs_debug_phase_ticks(sim._h, cur.ctypes.data, nb)
    d = (cur - prev) / 100.0          # us
    prev[:] = cur
    tot = d.sum(axis=1)
    if tot.max() > 1500:              # the step on which all worlds regenerate
        continue
    used += 1
    mean_acc += d.mean(axis=0)
    j = int(tot.argmax())
    max_acc += d[j]
    order = np.argsort(tot)
    p99_acc += d[order[int(0.99 * nb)]]
    ratios.append(tot.max() / tot.mean())
print(f"{used} steps; slowest wave / mean wave: median {np.median(ratios):.2f}, p10 {np.percentile(ratios, 10):.2f}, p90 {np.percentile(ratios, 90):.2f}")
print("phase        mean wave   p99 wave   slowest wave   (us per step)")
for i, nm in enumerate(names):
    print(f"{nm:10s} {mean_acc[i] / used:10.1f} {p99_acc[i] / used:10.1f} {max_acc[i] / used:12.1f}")
print(f"{'total':10s} {mean_acc.sum() / used:10.1f} {p99_acc.sum() / used:10.1f} {max_acc.sum() / used:12.1f}")
