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

Two ways to get N ranks (the reference is single-GPU, src/mgr.hpp:18 `gpuID`; the ranks are this build's extension):
  * `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N` — RANK / LOCAL_RANK / WORLD_SIZE come
    from the launcher;
  * plain `python bench.py --gpus N` — with WORLD_SIZE unset the process becomes a LAUNCHER: it starts N copies of
    itself (rank r pinned to device r), never touches a GPU itself, relays rank 0's JSON line and exits non-zero if
    any rank failed.  A rank whose device does not exist fails loudly ("device r not visible") instead of the job
    shrinking silently.
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


def csrc_fingerprint():
    """First 12 hex digits of the SHA-256 over the kernel sources: ties a profiles/*_traffic.json to the build it measured."""
    import hashlib
    d = os.path.join(ROOT, "marl-hideandseek_amd", "csrc")
    h = hashlib.sha256()
    for f in sorted(os.listdir(d)):
        if f.endswith((".h", ".hip")):
            h.update(f.encode())
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:12]


def cpu_baseline(seconds_budget=45.0, steps_of=1920):
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
        if time.time() - t0 > seconds_budget or steps >= steps_of:
            break
    dt = time.time() - t0
    return {"value": nworlds * steps / dt, "unit": "world-steps/s", "cores": cores, "kind": "port",
            "steps": steps, "steps_of": steps_of, "complete": steps >= steps_of,
            "sample": f"{nworlds} worlds x {steps} of {steps_of} steps (scripts/cpu_benchmark.py args: BASELINE configs[0]"
                      f"{'' if steps >= steps_of else ', cut at the time budget'}), own CPU restatement "
                      f"(oracle/), {cores} threads over worlds; not the Madrona CPU backend"}


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def launch_ranks(nranks, argv):
    """The launcher half of `bench.py --gpus N` (WORLD_SIZE unset): N child processes, one per GPU.  This process
    imports neither torch nor the simulator and makes no HIP call.  Returns the exit code."""
    import subprocess
    port = _free_port()
    procs = []
    for r in range(nranks):
        env = dict(os.environ)
        env.update({"RANK": str(r), "LOCAL_RANK": str(r), "WORLD_SIZE": str(nranks), "LOCAL_WORLD_SIZE": str(nranks),
                    "MASTER_ADDR": "127.0.0.1", "MASTER_PORT": str(port), "HS_BENCH_LAUNCHED": "1"})
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL, text=True))
    # A rank that dies (no such device, a HIP error) must not leave the others waiting in a barrier for ever: poll, and
    # once one has failed give the rest a few seconds, then end exactly the processes started here.
    failed, deadline = [], None
    while any(p.poll() is None for p in procs):
        for r, p in enumerate(procs):
            if p.poll() not in (None, 0) and r not in failed:
                failed.append(r)
                deadline = deadline or time.time() + 10.0
        if deadline and time.time() > deadline:
            for p in procs:
                if p.poll() is None:
                    p.kill()
        time.sleep(0.05)
    out0 = procs[0].stdout.read() if procs[0].stdout else ""
    failed = [r for r, p in enumerate(procs) if p.returncode != 0]
    if failed:
        sys.stderr.write("bench.py launcher: rank(s) %s of %d failed (exit codes %s); no result line\n"
                         % (failed, nranks, [procs[r].returncode for r in failed]))
        return 1
    lines = [ln for ln in out0.splitlines() if ln.startswith("{")]
    if not lines:
        sys.stderr.write("bench.py launcher: rank 0 printed no JSON line\n")
        return 1
    print(lines[-1], flush=True)
    return 0


def stub_rank(args, rank, local_rank, world_size):
    """HS_BENCH_STUB=1 (tests/test_bench_launcher.py, no GPU): the rank plumbing of main() — process group, barrier,
    max-over-ranks of the time, count of ranks that stepped, gather of per-rank times — around a sleep instead of the
    simulator.  Never set by the driver; the line says "stub": true."""
    import torch
    import torch.distributed as dist
    if world_size > 1:
        dist.init_process_group(backend="gloo")
        dist.barrier()
    t0 = time.perf_counter()
    time.sleep(0.01 * (rank + 1))
    dt_own = time.perf_counter() - t0
    dt, stepped, per_rank = dt_own, 1, [dt_own]
    if world_size > 1:
        dist.barrier()
        t = torch.tensor([dt_own], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([1], dtype=torch.int64)
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        stepped = int(c.item())
        g = [torch.zeros(1, dtype=torch.float64) for _ in range(world_size)]
        dist.all_gather(g, torch.tensor([dt_own], dtype=torch.float64))
        per_rank = [float(x.item()) for x in g]
    if rank == 0:
        print(json.dumps({"stub": True, "n_gpus": stepped, "world_size": world_size, "steps": args.steps,
                          "ms_per_step": dt * 1e3 / max(args.steps, 1),
                          "ms_per_step_per_rank": [x * 1e3 / max(args.steps, 1) for x in per_rank],
                          "env": {k: os.environ.get(k) for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT")},
                          "local_rank": local_rank, "worlds_per_gpu": args.worlds_per_gpu,
                          "world_offsets": [r * args.worlds_per_gpu for r in range(world_size)]}), flush=True)
    if world_size > 1:
        dist.barrier()
        dist.destroy_process_group()
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1920)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--worlds-per-gpu", type=int, default=WORLDS_PER_GPU,
                    help="16000 = BASELINE configs[1] (default); 16384 = one rank of configs[3] (131072 worlds over 8 GPUs)")
    ap.add_argument("--flags", type=int, default=0)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-preroll", action="store_true",
                    help="time the steps right after the warm-up even in a run shorter than an episode (a preparation-phase window)")
    ap.add_argument("--cpu-seconds", type=float, default=45.0,
                    help="time budget of the cpu_baseline leg (2000 worlds x 1920 steps take about 25 s on 16 threads)")
    ap.add_argument("--profile-every", type=int, default=0,
                    help="steps between two steps that carry HIP events (0: 7, or 4 for runs of at most 64 steps)")
    args = ap.parse_args()
    if args.gpus < 1:
        ap.error("--gpus must be >= 1")

    # `bench.py --gpus N` outside a launcher: become the launcher (before anything touches a GPU).
    # (HS_BENCH_FORCE_LAUNCHER=1 sends --gpus 1 through the launcher too: the 1-GPU test of this path)
    if (args.gpus > 1 or os.environ.get("HS_BENCH_FORCE_LAUNCHER") == "1") and "WORLD_SIZE" not in os.environ:
        return launch_ranks(args.gpus, sys.argv[1:])

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world_size = int(os.environ.get("WORLD_SIZE", "1"))
    if world_size != args.gpus:
        raise RuntimeError(f"--gpus {args.gpus} but the launcher started WORLD_SIZE={world_size} ranks")
    if os.environ.get("HS_BENCH_STUB") == "1":
        return stub_rank(args, rank, local_rank, world_size)

    import numpy as np
    import torch
    import gpu_hideseek

    # Rehearsal switch for a one-GPU box: HS_BENCH_REHEARSE=1 maps every rank to GPU 0 and uses gloo for the barrier /
    # max-reduction (RCCL refuses two ranks on one device).  The driver's runs never set it.
    rehearse = os.environ.get("HS_BENCH_REHEARSE") == "1"
    if rehearse:
        local_rank = 0
    ndev = torch.cuda.device_count()          # (counting devices does not initialise the GPU)
    if local_rank >= ndev:
        raise RuntimeError(f"rank {rank}: device {local_rank} not visible ({ndev} GPU(s) on this node); "
                           f"--gpus {args.gpus} needs {args.gpus} devices")
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs a GPU: the HIP path is the only execution path")
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
    # A run shorter than an episode sees only the part of it the window falls in, and the two parts differ: in the 95
    # preparation steps two of the four agents are frozen (k_physics 0.36 ms), afterwards all four push things around
    # (0.49 ms).  Right after the warm-up a short window — the driver's 20 steps — would be preparation phase only, i.e.
    # 20 % better than the run over whole episodes.  Short runs are therefore moved, by untimed steps, to where the window
    # holds the two phases in the episode's own proportion (95 : 145); runs of 240 steps or more start right away.
    preroll = 0
    if args.steps < 240 and not args.no_preroll:
        first = (95 - round(args.steps * 95 / 240)) % 240          # first timed episode step
        preroll = (first - args.warmup) % 240
        for _ in range(preroll):
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
    barrier()
    t0 = time.perf_counter()
    tprev = t0
    step_s = []                        # host wall time of every step (a step blocks until its kernels are done)
    for i in range(args.steps):
        if i % P:
            one_step()
        else:
            sim.set_profiling(True)
            one_step()
            k = sim.last_step_kernel_ms()
            sim.set_profiling(False)
            nsamp += 1
            kms["physics"] += k["physics"]
            if not skip_obs:
                kms["observe"] += k["observe"]
        tnow = time.perf_counter()
        step_s.append(tnow - tprev)
        tprev = tnow
    barrier()
    dt = time.perf_counter() - t0
    dt_own = dt
    stepped, per_rank_ms = 1, [dt_own / args.steps * 1e3]
    if dist is not None:
        cdev = "cpu" if rehearse else dev
        t = torch.tensor([dt], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
        c = torch.tensor([1], dtype=torch.int64, device=cdev)          # ranks that actually stepped
        dist.all_reduce(c, op=dist.ReduceOp.SUM)
        stepped = int(c.item())
        g = [torch.zeros(1, dtype=torch.float64, device=cdev) for _ in range(world_size)]
        dist.all_gather(g, torch.tensor([dt_own], dtype=torch.float64, device=cdev))
        per_rank_ms = [float(x.item()) / args.steps * 1e3 for x in g]

    # A candidate pair beyond the LDS capacities spills to the slow path (counted), none is ever dropped: say so per run.
    status = sim.device_status()
    assert status["dropped_candidate_pairs"] == 0, f"broadphase candidate pairs were dropped: {status}"

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
    # HBM-side traffic cannot be measured inside this process (PMC counters need rocprofv3): it is read from the newest
    # committed PMC run (tools/pmc.sh + tools/pmc_summary.py) and only when that run was made with THIS build of the
    # kernels (fingerprint of csrc/); otherwise null.  The source is always named.
    traffic, traffic_source = None, "none: no profiles/*_traffic.json"
    tfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_traffic.json"))
    if tfiles and N == WORLDS_PER_GPU and args.flags == 0:
        try:
            tj = json.load(open(os.path.join(ROOT, "profiles", tfiles[-1])))
            if tj.get("csrc_sha") == csrc_fingerprint():
                traffic = tj.get(dom, {}).get("hbm_bytes_per_step")
                traffic_source = f"profiles/{tfiles[-1]} (rocprofv3 PMC passes of a 40-step run of this build, csrc {tj.get('csrc_sha')}; not this run)"
            else:
                traffic_source = f"dropped: profiles/{tfiles[-1]} was measured on other kernel sources (csrc {tj.get('csrc_sha')} != {csrc_fingerprint()})"
        except Exception as e:
            traffic_source = f"unreadable profiles/{tfiles[-1]}: {e}"
    elif tfiles:
        traffic_source = "none: the committed PMC run is for 16000 worlds, sim_flags 0"
    # What actually bounds the kernels is VALU issue, not HBM: the instruction counts of the committed SQ-counter run of THIS
    # build (tools/sqpmc.sh -> profiles/*_sq_counters.txt, same fingerprint rule as the traffic) against the rate at which a
    # SIMD issues f32 VALU instructions with all its wave slots busy (tools/ubench/pk_rate.hip, measured on the part).
    valu_issue = None
    sfiles = sorted(f for f in os.listdir(os.path.join(ROOT, "profiles")) if f.endswith("_sq_counters.txt"))
    if sfiles and N == WORLDS_PER_GPU and args.flags == 0:
        try:
            txt = open(os.path.join(ROOT, "profiles", sfiles[-1])).read()
            if f"csrc_sha {csrc_fingerprint()}" in txt:
                valu_issue = {"source": f"profiles/{sfiles[-1]} (rocprofv3 SQ counters of a 24-step run of this build, window placed like a short bench run; not this run)",
                              "simd_count": 1024, "issue_ns_per_instruction_and_simd": 1.2,
                              "peak_source": "tools/ubench/pk_rate.hip: a SIMD with all wave slots busy issues one f32 VALU instruction per 1.2 ns"}
                for n in kms:
                    for line in txt.splitlines():
                        if line.startswith("sq1") and names[n] in line and "SQ_INSTS_VALU=" in line:
                            insts = float(line.split("SQ_INSTS_VALU=")[1].split()[0])
                            floor_ms = insts * 1.2e-9 / 1024 * 1e3
                            valu_issue[n] = {"valu_wave_instructions_per_launch": insts, "issue_bound_ms": floor_ms,
                                             "frac": floor_ms / (kms[n] / max(nsamp, 1))}
        except Exception as e:
            valu_issue = {"source": f"unreadable profiles/{sfiles[-1]}: {e}"}
    step_bytes, step_per_world = algorithmic_bytes(sim, A, "physics_only_step" if skip_obs else "step")
    step_ms = dt / args.steps * 1e3
    roofline = {"bound": "hbm", "kernel": per_stage[dom]["kernel"], "achieved": per_stage[dom]["achieved_GBps"],
                "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": per_stage[dom]["frac"], "traffic": traffic,
                "traffic_source": traffic_source,
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
                "valu_issue": valu_issue,
                "schedule": "k_physics then k_observe on one stream"}

    if rank == 0:
        assert stepped == world_size, f"{stepped} of {world_size} ranks stepped"
        total_worlds = N * stepped
        out = {
            "metric": "world-steps/sec (agent-steps/sec derived) at 16K worlds, 1/2/4/8 GPU",
            "value": total_worlds * args.steps / dt,
            "unit": "world-steps/s",
            "n_gpus": stepped, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "ms_per_step_per_rank": per_rank_ms,
            # host wall time per step on rank 0 (a step blocks until its kernels are done): the scatter of the run itself
            "ms_per_step_percentiles": {k: float(np.percentile(np.asarray(step_s) * 1e3, q))
                                        for k, q in (("min", 0), ("p10", 10), ("p50", 50), ("p90", 90), ("p99", 99), ("max", 100))},
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": f"{N} worlds/GPU x {args.steps} steps, 2 hiders + 2 seekers, sim_flags={args.flags}, "
                                   "rand_seed=0, random move actions in [-5,4] each step (scripts/benchmark.py args)",
                       "worlds_per_gpu": N, "total_worlds": total_worlds, "agents_per_world": A,
                       "sharding": "contiguous world ranges per rank, no collective on the step path"},
            "agent_steps_per_sec": total_worlds * A * args.steps / dt,
            # which part of the 240-step episode the timed window covers (every world is in lock-step): steps < 95
            # are the preparation phase (seekers frozen, no reward rays), every 240th step regenerates all levels
            "episode_steps_covered": {"first": (args.warmup + preroll) % 240, "count": args.steps,
                                      "untimed_steps_before": {"warmup": args.warmup, "preroll": preroll},
                                      "preparation_steps_in_window": sum(1 for i in range(args.steps) if (args.warmup + preroll + i) % 240 < 95),
                                      "level_regenerations": ((args.warmup + preroll) % 240 + args.steps) // 240,
                                      "note": "whole episodes" if args.steps >= 240 else
                                      ("a preparation-phase window (two of four agents frozen): about 20 % faster than whole episodes"
                                       if (args.warmup + preroll) % 240 + args.steps <= 95 else
                                       "shorter than an episode: moved by untimed preroll steps so that preparation and "
                                       "post-preparation steps are in the episode's own proportion (95 : 145); no level regeneration "
                                       "inside, and the post-preparation steps are the first ones after the seekers' release, when the "
                                       "contact load is still building up: about 7 % faster than whole episodes (32.4 M world-steps/s)")},
            "roofline": roofline,
        }
        out["device_status"] = status
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
    sys.exit(main() or 0)
