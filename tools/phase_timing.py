"""Per-phase time of the physics kernel and per-section time of k_observe (development aid).

Needs the timing build of the library (python marl-hideandseek_amd/build.py --timing [--sat-counters]):
  HS_LIB_PATH=$PWD/marl-hideandseek_amd/lib/libhideseek_timing.so python tools/phase_timing.py 16000
--sat-counters adds the work counters of the convex tests (their atomics disturb the phase times).
"""
import os, sys, ctypes as C, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "marl-hideandseek_amd"))
import gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0,
    min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
act = sim.action_tensor().to_torch()
sim.init()
steps = 240
for i in range(steps):
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    sim.step()
nb = (N + 7) // 8          # one wave (workgroup) per octet of 8 worlds
out = np.zeros((nb, 10), np.int64)
L = sim._L
L.hs_debug_phase_ticks.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
n = L.hs_debug_phase_ticks(sim._h, out.ctypes.data, nb)
us = out[:n] / 100.0 / steps     # 100 MHz ticks -> us per step
names = ["pre", "integrate", "detect", "sat", "dd_pos", "body_pos", "dd_vel", "body_vel", "post+store", "load"]
print("phase      mean_us  min_us  max_us   (per step, over %d waves = octets)" % n)
for i, nm in enumerate(names):
    print(f"{nm:10s} {us[:, i].mean():7.1f} {us[:, i].min():7.1f} {us[:, i].max():7.1f}")
tot = us.sum(axis=1)
print(f"total      {tot.mean():7.1f} {tot.min():7.1f} {tot.max():7.1f}")

# k_observe sections (ticks summed over all waves; 3 waves per world at 4 agents; sections inside the ray loop are
# reported by the waves whose first lane has a ray)
ob = np.zeros(16, np.int64)
L.hs_debug_observe_ticks.argtypes = [C.c_void_p, C.c_void_p]
if L.hs_debug_observe_ticks(sim._h, ob.ctypes.data) == 0:
    launches = steps + 1
    waves = N * ((4 * 46 + 63) // 64)
    onames = ["stage(+wait)", "agent table", "ray setup", "walls", "planes", "cull(+barrier)", "hull tests", "ray results", "obs rows"]
    tot = ob[:9].sum()
    print("k_observe section    us per wave per launch   share")
    for i, nm in enumerate(onames):
        print(f"{nm:18s} {ob[i] / 100.0 / launches / waves:10.2f} {100.0 * ob[i] / max(tot, 1):8.1f} %")
    print(f"pairs per world per launch: {ob[9] / launches / N:.1f} box-shaped, {ob[10] / launches / N:.1f} ramps")
sc = np.zeros(16, np.int64)
L.hs_debug_sat_counters.argtypes = [C.c_void_p, C.c_void_p]
if L.hs_debug_sat_counters(sim._h, sc.ctypes.data) == 0 and sc[0] > 0:
    c = float(sc[0])
    print(f"convex tests per wave and substep: {sc[1] / c:.1f} box-shaped items + {sc[2] / c:.1f} wedge items in {sc[3] / c:.2f} rounds of 32; "
          f"{sc[4] / c:.1f} colliding pairs in {sc[5] / c:.2f} contact rounds")
    print(f"  time per call: {sc[7] / c / 100:.2f} us, of which contact generation {sc[6] / c / 100:.2f} us, wedge rounds {sc[8] / c / 100:.2f} us")
