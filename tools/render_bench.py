"""Time of k_render (agent-view depth / RGB) at benchmark size: render_bench.py [worlds] [W] [H] [repeats]."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))
import torch  # noqa: E402
import gpu_hideseek  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
W = int(sys.argv[2]) if len(sys.argv) > 2 else 64
H = int(sys.argv[3]) if len(sys.argv) > 3 else 64
R = int(sys.argv[4]) if len(sys.argv) > 4 else 10
sim = gpu_hideseek.HideAndSeekSimulator(
    exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N, sim_flags=0, rand_seed=0, min_hiders=2,
    max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1, enable_batch_renderer=True, batch_render_width=W,
    batch_render_height=H)
sim.init()
act = sim.action_tensor().to_torch()
for _ in range(120):
    act[:, 0:2] = torch.randint(-5, 5, (act.shape[0], 2), device=act.device, dtype=torch.int32)
    sim.step()
sim.render()
torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(R):
    sim.render()
dt = (time.perf_counter() - t0) / R
views = N * 4
out_bytes = views * W * H * 8
print(f"{N} worlds x 4 views x {W}x{H}: {dt * 1e3:.3f} ms per render, {views * W * H / dt / 1e9:.2f} G rays/s, "
      f"output {out_bytes / 1e6:.0f} MB -> {out_bytes / dt / 1e9:.0f} GB/s written ({out_bytes / dt / 8e12 * 100:.1f} % of 8 TB/s)")
d = sim.depth_tensor().to_torch()
print("sky fraction", float((d == 0).float().mean()), "mean depth", float(d[d > 0].mean()))
