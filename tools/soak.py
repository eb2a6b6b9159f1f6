import os, sys, time
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "marl-hideandseek_amd"))
import torch, gpu_hideseek
for cfg in (dict(h=2, s=2, flags=0, seed=0), dict(h=3, s=3, flags=13, seed=5)):
    sim = gpu_hideseek.HideAndSeekSimulator(exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=16000, sim_flags=cfg["flags"],
        rand_seed=cfg["seed"], min_hiders=cfg["h"], max_hiders=cfg["h"], min_seekers=cfg["s"], max_seekers=cfg["s"], num_pbt_policies=1)
    sim.init()
    act = sim.action_tensor().to_torch()
    t0 = time.time()
    for i in range(6000):
        if cfg["flags"]:
            act[:, 0:3] = torch.randint(0, 5, (act.shape[0], 3), device=act.device, dtype=torch.int32)
            act[:, 3:5] = torch.randint(0, 2, (act.shape[0], 2), device=act.device, dtype=torch.int32)
        else:
            act[:, 0:3] = torch.randint(0, 11, (act.shape[0], 3), device=act.device, dtype=torch.int32)
            act[:, 3:5] = torch.randint(0, 2, (act.shape[0], 2), device=act.device, dtype=torch.int32)
        sim.step()
    ok = all(bool(torch.isfinite(getattr(sim, n + "_tensor")().to_torch()).all()) for n in ("self_data", "agent_data", "box_data", "ramp_data", "lidar", "reward"))
    b, m = sim.debug_bodies()
    import numpy as np
    print(cfg, "6000 steps in %.1f s" % (time.time() - t0), "finite", ok, "bodies finite", bool(np.isfinite(b).all()), "max |pos|", float(np.abs(b[:, :, :3]).max()), sim.device_status())
    del sim
