"""Developer tool: per-phase cycle breakdown of k_physics (needs lib/libhideseek_timing.so built with
-DHS_PHASE_TIMING).  Usage: python tools/phase_timing.py [worlds] [steps]"""
import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))
import gpu_hideseek
gpu_hideseek._LIB_PATH = os.path.join(ROOT, "marl-hideandseek_amd", "lib", "libhideseek_timing.so")
import torch
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=1, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0, min_hiders=2,
                                        max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1)
sim.init()
L = gpu_hideseek._load()
L.hs_debug_phase_cycles.argtypes = [C.c_void_p, C.POINTER(C.c_uint64 * 16), C.c_int32]
act = sim.action_tensor().to_torch(); move = act[..., 0:2]
for _ in range(100):
    sim.step(); torch.randint(-5, 5, move.shape, out=move, dtype=torch.int32, device=move.device)
out = (C.c_uint64 * 16)()
L.hs_debug_phase_cycles(sim._h, C.byref(out), 1)
for _ in range(steps):
    sim.step(); torch.randint(-5, 5, move.shape, out=move, dtype=torch.int32, device=move.device)
L.hs_debug_phase_cycles(sim._h, C.byref(out), 0)
names = ["stage", "move+action", "P1 integrate+aabb", "P2 candidates+lists", "P3a ground collide", "P3b SAT (packed)",
         "P4a DD pos (packed)", "P4b ground pos", "P4c walls pos (packed)", "P5 derive + P6a DD vel", "P6b ground vel",
         "P6c walls vel (packed)"]
tot = sum(out[:12])
for i, n in enumerate(names):
    print(f"{n:28s} {out[i] / tot * 100:6.2f}%   {out[i] / (steps * ((N + 7) // 8 * 2)):10.0f} cycles/wave-step")
print("total cycles/wave-step", tot / (steps * ((N + 7) // 8 * 2)))
