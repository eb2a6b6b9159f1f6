#!/usr/bin/env python3
"""Headline benchmark: world-steps/s of Manager.step() at 16 000 worlds per GPU.

Mirrors scripts/benchmark.py:67-92 of the reference: 5 untimed warm-up steps, then K timed
iterations of `sim.step()` + `torch.randint(-5, 5)` into action columns 0-1, synthetic inputs
resident in HBM.  One process per GPU (torch.distributed / RCCL only for the barrier and the
max-over-ranks reduction — worlds are independent, there is no collective on the step path);
rank r simulates global worlds [r*16000, (r+1)*16000) (weak scaling).

Prints ONE JSON line (rank 0) with the bench contract fields plus
  "roofline":     dominant kernel vs the HBM roofline (algorithmic bytes / HIP-event kernel time)
  "cpu_baseline": the CPU oracle (own restatement, NOT the Madrona CPU backend) on the host cores
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(ROOT, "marl-hideandseek_amd"))

WORLDS_PER_GPU = 16000
HBM_PEAK_GBPS = 8000.0        # /opt/skills/guides/MI355X_MICROARCH.md: 8.0 TB/s spec


def algorithmic_bytes(sim, A, kernel):
    """SURVEY §8(d) per-world-step bytes, summed over the actual worlds of this shard.
    physics: 2*56*D + 16*S + 2*64 + 2*72*J + 2*20*A ; observe: 1268*A + 144 (+ reads of 56*D + 16*S)."""
    import numpy as np
    _, meta = sim.debug_bodies()
    _, info = sim.debug_walls()
    D = (meta[:, :, 0] >= 0).sum(axis=1).astype(np.float64)
    S = info[:, 0].astype(np.float64)
    n = float(len(D))
    if kernel == "physics":
        per_world = 2 * 56 * D + 16 * S + 2 * 64 + 2 * 20 * A
    elif kernel == "observe":
        per_world = 56 * D + 16 * S + 64 + (1268 * A + 144)
    else:                               # "step": SURVEY §8(d)'s B for the whole step (state r+w once, exports written once)
        per_world = 2 * 56 * D + 16 * S + 2 * 64 + 2 * 20 * A + (1268 * A + 144 if kernel == "step" else 0)
    return float(per_world.sum()), float(per_world.sum() / n)


def cpu_baseline(seconds_budget=20.0):
    """Oracle timed on the host cores over a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import numpy as np
    import hs_ref
    hs_ref.build()
    cores = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    cores = min(cores, 16)      # a one-GPU box's CPU share is 16 cores
    nworlds = 2000
    ref = hs_ref.RefSim(nworlds, sim_flags=0, rand_seed=0, threads=cores)
    ref.init()
    rng = np.random.default_rng(0)
    act = ref.tensor("action")
    for _ in range(2):
        ref.step()
    t0 = time.time()
    steps = 0
    while True:
        ref.step()
        act[:, 0:2] = rng.integers(-5, 5, size=(act.shape[0], 2))
        steps += 1
        if time.time() - t0 > seconds_budget or steps >= 480:
            break
    dt = time.time() - t0
    return {"value": nworlds * steps / dt, "unit": "world-steps/s", "cores": cores, "kind": "port",
            "sample": f"{nworlds} worlds x {steps} steps (scripts/cpu_benchmark.py args), own CPU restatement "
                      f"(oracle/), {cores} threads over worlds; not the Madrona CPU backend"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--worlds-per-gpu", type=int, default=WORLDS_PER_GPU,
                    help="16000 = BASELINE configs[1] (default); 16384 = one rank of configs[3] (131072 worlds over 8 GPUs)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=20.0)
    ap.add_argument("--profile-every", type=int, default=0,
                    help="steps between two steps that carry HIP events (0: 7, or 4 for runs of at most 64 steps)")
    args = ap.parse_args()

    import torch
    import gpu_hideseek

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the HIP path is the only execution path")
    # Rehearsal switch for a one-GPU box: HS_BENCH_REHEARSE=1 maps every rank to GPU 0 and uses gloo for the barrier /
    # max-reduction (RCCL refuses two ranks on one device).  The driver's runs never set it.
    rehearse = os.environ.get("HS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world_size > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist.init_process_group(backend="gloo")
        else:
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    N = args.worlds_per_gpu
    torch.manual_seed(rank)
    sim = gpu_hideseek.HideAndSeekSimulator(
        exec_mode=gpu_hideseek.madrona.ExecMode.CUDA, gpu_id=local_rank, num_worlds=N, sim_flags=args.flags,
        rand_seed=0, min_hiders=2, max_hiders=2, min_seekers=2, max_seekers=2, num_pbt_policies=1,
        world_offset=rank * N)
    sim.init()
    A = sim.agents_per_world
    actions = sim.action_tensor().to_torch()
    move = actions[..., 0:2]
    move.copy_(torch.zeros_like(move))
    dev = actions.device

    def one_step():
        sim.step()
        torch.randint(-5, 5, move.shape, out=move, dtype=torch.int32, device=dev)

    for _ in range(args.warmup):
        one_step()

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Kernel durations: HIP events on the launch stream around the kernels of every P-th step of the timed region (recording
    # and resolving them costs 12 us per step, 2.7 % of a step, so they are not put on every step; 7 is coprime to the
    # 240-step episode, so level-regeneration steps are sampled in proportion).
    P = args.profile_every if args.profile_every > 0 else (7 if args.steps > 64 else 4 if args.steps >= 8 else 1)
    nsamp = 0
    sim.set_profiling(False)
    # the kernels a step launches: k_physics (its tail is the per-step reset) and, unless skipped, k_observe
    skip_obs = bool(args.flags & (1 << 16))
    kms = {"physics": 0.0} if skip_obs else {"physics": 0.0, "observe": 0.0}
    overlapped = False
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        if i % P:
            one_step()
            continue
        sim.set_profiling(True)
        one_step()
        k = sim.last_step_kernel_ms()
        sim.set_profiling(False)
        nsamp += 1
        kms["physics"] += k["physics"]
        if not skip_obs:
            overlapped = overlapped or k["observe"] < 0
            kms["observe"] += max(k["observe"], 0.0)
    barrier()
    dt = time.perf_counter() - t0
    obs_pass_steps = 0
    if overlapped:
        # Under the dependency schedule (HS_OVERLAP=1) k_observe runs beside k_physics and has no duration of its own in
        # the timed region; its kernel time comes from an extra, untimed pass with the two kernels launched one after
        # the other (same results).  k_physics' events above are from the timed region.
        sim.set_overlap(False)
        sim.set_profiling(True)
        kms["observe"] = 0.0
        obs_pass_steps = min(args.steps, 240)
        for _ in range(obs_pass_steps):
            one_step()
            kms["observe"] += sim.last_step_kernel_ms()["observe"]
        kms["observe"] *= nsamp / max(obs_pass_steps, 1)
        sim.set_profiling(False)
        sim.set_overlap(True)
    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device="cpu" if rehearse else dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())

    # Roofline position of the dominant kernel.  A step is two kernels: k_physics (persistent: movement / actions,
    # 4 XPBD substeps, rewards, per-step reset) and k_observe.  Durations are HIP events on the launch stream
    # (hs_set_profiling); profiles/ holds the rocprofv3 --kernel-trace --stats summary of this command, whose
    # per-kernel averages agree with them.  Only kernels that were launched are listed.
    names = {"physics": "k_physics", "observe": "k_observe"}
    per_stage = {}
    for n in kms:
        avg_ms = kms[n] / max(nsamp, 1)
        total_bytes, per_world = algorithmic_bytes(sim, A, n)
        ach = total_bytes / (avg_ms * 1e-3) / 1e9 if avg_ms > 0 else 0.0
        per_stage[n] = {"kernel": names[n], "avg_ms": avg_ms, "algorithmic_bytes_per_world_step": per_world,
                        "achieved_GBps": ach, "frac": ach / HBM_PEAK_GBPS}
        assert 0.0 <= per_stage[n]["frac"] <= 1.0, f"roofline fraction of {n} out of range: {per_stage[n]}"
    dom = max(kms, key=lambda n: kms[n])
    traffic = None
    # PMC FETCH_SIZE / WRITE_SIZE passes (tools/pmc.sh): the newest round's file
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
    if tfiles and N == WORLDS_PER_GPU:
        try:
            traffic = json.load(open(os.path.join(ROOT, "profiles", tfiles[-1]))).get(dom, {}).get("hbm_bytes_per_step")
        except Exception:
            traffic = None
    step_bytes, step_per_world = algorithmic_bytes(sim, A, "physics_only_step" if skip_obs else "step")
    step_ms = dt / args.steps * 1e3
    roofline = {"bound": "hbm", "kernel": per_stage[dom]["kernel"], "achieved": per_stage[dom]["achieved_GBps"],
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": per_stage[dom]["frac"], "traffic": traffic,
                "avg_kernel_ms": per_stage[dom]["avg_ms"],
                "algorithmic_bytes_per_world_step": per_stage[dom]["algorithmic_bytes_per_world_step"],
                "stages": per_stage,
                # the whole step against the roofline: SURVEY §8(d)'s B per world-step over the wall time per step
                "step": {"algorithmic_bytes_per_world_step": step_per_world,
                         "achieved_GBps": step_bytes / (step_ms * 1e-3) / 1e9,
                         "frac": step_bytes / (step_ms * 1e-3) / 1e9 / HBM_PEAK_GBPS},
                "kernel_ms_per_step": {n: kms[n] / max(nsamp, 1) for n in kms},
                "kernel_time_samples": {"every": P, "count": nsamp, "how": "HIP events on the launch stream around the kernels of "
                                        "every P-th step of the timed region"},
                "schedule": ("k_observe beside k_physics, octets in finish order (dependency schedule); k_physics timed "
                             "in the timed region, k_observe in an extra pass of %d sequential steps" % obs_pass_steps)
                if overlapped else "k_physics then k_observe on one stream"}

    if rank == 0:
        total_worlds = N * world_size
        out = {
            "metric": "world-steps/sec (agent-steps/sec derived) at 16K worlds, 1/2/4/8 GPU",
            "value": total_worlds * args.steps / dt,
            "unit": "world-steps/s",
            "n_gpus": world_size, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N} worlds/GPU x {args.steps} steps, 2 hiders + 2 seekers, sim_flags={args.flags}, "
                                   "rand_seed=0, random move actions in [-5,4] each step (scripts/benchmark.py args)",
                       "worlds_per_gpu": N, "total_worlds": total_worlds, "agents_per_world": A,
                       "sharding": "contiguous world ranges per rank, no collective on the step path"},
            "agent_steps_per_sec": total_worlds * A * args.steps / dt,
            # which part of the 240-step episode the timed window covers (every world is in lock-step): steps < 95
            # are the preparation phase (seekers frozen, no reward rays), every 240th step regenerates all levels
            "episode_steps_covered": {"first": args.warmup % 240, "count": args.steps,
                                      "level_regenerations": (args.warmup % 240 + args.steps) // 240,
                                      "note": "a window inside [0, 95) is preparation phase only"
                                      if args.warmup % 240 + args.steps <= 95 else "covers all phases of an episode"
                                      if args.steps >= 240 else "partial episode"},
            "roofline": roofline,
        }
        if not args.no_cpu_baseline and world_size == 1:
            out["cpu_baseline"] = cpu_baseline(args.cpu_seconds)
        elif not args.no_cpu_baseline:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    del sim
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
