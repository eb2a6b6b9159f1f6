"""The simulator side of BASELINE configs[4] (scripts/jax_train.py:69-81, 146-148) without jaxlib: 16 000 worlds, 3 hiders + 3
seekers, RandomFlipTeams | UseFixedWorld | ZeroAgentVelocity, seed 5, actions drawn from [0,5)^3 x [0,2)^2 every step,
stepped through the stream entry point sim.jax() registers (hs_jax_step = stream_step) with caller-owned buffers on a
caller-owned stream.  A random policy stands in for the PPO network: what is timed is the simulator and its bindings.
    train_config_bench.py [worlds] [steps]"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))
import torch  # noqa: E402
import gpu_hideseek  # noqa: E402

N = int(sys.argv[1]) if len(sys.argv) > 1 else 16000
STEPS = int(sys.argv[2]) if len(sys.argv) > 2 else 960
sim = gpu_hideseek.HideAndSeekSimulator(
    exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=0, num_worlds=N,
    sim_flags=gpu_hideseek.SimFlags.RandomFlipTeams | gpu_hideseek.SimFlags.UseFixedWorld | gpu_hideseek.SimFlags.ZeroAgentVelocity,
    rand_seed=5, min_hiders=3, max_hiders=3, min_seekers=3, max_seekers=3, num_pbt_policies=1)
A = sim.agents_per_world
names = [n for n in sim.train_interface()["observations"]]
obs = [torch.zeros_like(t.to_torch()) for t in sim.train_interface()["observations"].values()]
rew = torch.zeros_like(sim.reward_tensor().to_torch()); done = torch.zeros_like(sim.done_tensor().to_torch())
epres = torch.zeros_like(sim.episode_result_tensor().to_torch())
act = torch.zeros_like(sim.action_tensor().to_torch()); resets = torch.zeros_like(sim.reset_tensor().to_torch())
pol = torch.zeros_like(sim.policy_assignments_tensor().to_torch())
strm = torch.cuda.Stream()
sim.stream_init(strm.cuda_stream, obs)
bufs = [act, resets, pol] + obs + [rew, done, epres]


def one_step():
    with torch.cuda.stream(strm):
        act[:, 0:3] = torch.randint(0, 5, (act.shape[0], 3), device=act.device, dtype=torch.int32)
        act[:, 3:5] = torch.randint(0, 2, (act.shape[0], 2), device=act.device, dtype=torch.int32)
    sim.stream_step(strm.cuda_stream, bufs)


for _ in range(10):
    one_step()
strm.synchronize()
t0 = time.perf_counter()
for _ in range(STEPS):
    one_step()
strm.synchronize()
dt = time.perf_counter() - t0
st = sim.device_status()
assert st["dropped_candidate_pairs"] == 0, st          # pairs beyond the LDS capacities spill (counted), none is dropped
print(json.dumps({"config": "BASELINE configs[4] simulator side: %d worlds, 3+3 agents, flags 13, seed 5, random policy, stream_step with caller buffers" % N,
                  "steps": STEPS, "ms_per_step": dt / STEPS * 1e3, "world_steps_per_s": N * STEPS / dt,
                  "agent_steps_per_s": N * A * STEPS / dt, "dropped_candidate_pairs": st["dropped_candidate_pairs"],
                  "spilled_candidate_pairs": st["spilled_candidate_pairs"],
                  "finite": bool(all(torch.isfinite(o).all() for o in obs if o.dtype == torch.float32))}))
