"""Per-step kernel times of the benchmark workload (HIP events): distribution of k_physics over an episode."""
import os, sys, numpy as np, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "marl-hideandseek_amd"))
import gpu_hideseek
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=1, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0, min_hiders=2, max_hiders=2,
                                        min_seekers=2, max_seekers=2, num_pbt_policies=1)
sim.init(); sim.set_profiling(True)
act = sim.action_tensor().to_torch()
ph, ob = [], []
for i in range(485):
    sim.step()
    act[:, :2] = torch.randint(-5, 5, (N * 4, 2), dtype=torch.int32, device="cuda")
    k = sim.last_step_kernel_ms(); ph.append(k["physics"]); ob.append(k["observe"])
ph, ob = np.array(ph) * 1e3, np.array(ob) * 1e3
ep = np.arange(485) % 240
for name, a in (("physics", ph), ("observe", ob)):
    reg = a[ep == 239]; rest = a[ep != 239]
    print(f"{name}: regen steps {reg.round(0)} us; other steps min {rest.min():.0f} p10 {np.percentile(rest,10):.0f} median {np.median(rest):.0f} "
          f"p90 {np.percentile(rest,90):.0f} max {rest.max():.0f} mean {rest.mean():.0f}; prep-phase mean {a[(ep<95)].mean():.0f}, after {a[(ep>=95)&(ep!=239)].mean():.0f}")
