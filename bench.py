#!/usr/bin/env python3
"""bench.py - simulated seconds per wall-clock second of the SSP-SLAM path integrator on MI355X.

Metric (BASELINE.json): sim-sec / wall-sec of the simulator step loop (``n_steps*dt / wall(sim.run)``,
build excluded - reference ``experiments/run_pathint.py:160-165``) for ``PathIntegration`` 2-D,
``ssp_dim=1015``, ``pi_n_neurons=10000`` per VCO (508 VCO ensembles, 5.08 M LIF neurons), synthetic
band-limited random path, random-seed 0.

    python bench.py --gpus 1 --steps K --warmup W

One "step" = one block of ``--block`` simulator timesteps (default 1000 = one simulated second) with all
inputs already resident in HBM.  The timed region is exactly K blocks between barrier +
torch.cuda.synchronize() pairs; the max over ranks is taken; rank 0 prints one JSON line.

With N > 1 (launched by torch.distributed.run, one rank per GPU) the 508 VCO ensembles of the SAME
model are sharded over the ranks (strong scaling): each rank steps its VCOs, the decoded
oscillator states are all-gathered once per block (RCCL) and rank 0 applies the linear read-out.

Extra legs on rank 0 at N = 1 (both outside the timed region):
  * roofline: every full-block launch of the dominant kernel (k_ens_block: one launch steps all VCOs through
    a whole block of timesteps) of a 2-block pass is bracketed with HIP events on the simulator's stream;
    achieved = algorithmic bytes per launch (52 B per neuron-step x neuron-steps per launch) /
    average duration.  The kernel keeps the neuron parameters in registers for the whole block, so this figure
    exceeds the streaming-HBM roofline (frac > 1); `traffic` is the HBM traffic actually measured (rocprofv3
    PMC passes) and `valu` prices the kernel against what bounds it now, VALU issue slots.
  * cpu_baseline: the NumPy float64 oracle (oracle/stepper.py, a restatement of nengo's reference
    simulator) stepping the same built model for a bounded sample of timesteps on the host cores; the
    GPU trajectory is checked against it on that window (parity, cosine error).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed blocks")
    ap.add_argument("--warmup", type=int, default=2, help="untimed blocks")
    ap.add_argument("--block", type=int, default=1000, help="simulator timesteps per block")
    ap.add_argument("--ssp-dim", type=int, default=1015)
    ap.add_argument("--pi-n-neurons", type=int, default=10000)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--eval-points", type=int, default=4000,
                    help="decoder-solve eval points per VCO (nengo's default max(1500, 2n) = 20000 costs ~10x the build)")
    ap.add_argument("--cpu-steps", type=int, default=60, help="oracle timesteps for the cpu_baseline / parity leg (0 = skip)")
    ap.add_argument("--sim-block", type=int, default=0,
                    help="timesteps per time-batched block inside the simulator (one k_ens_block launch each); "
                         "0 = --block, so that a bench step is exactly one block")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="length of the separate roofline leg that is only run when the timed region had no timed launch of "
                         "the dominant kernel (--sim-block different from --block); 0 = 2 full simulator blocks")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path with several ranks sharing one GPU (RCCL needs one GPU per rank)")
    return ap.parse_args()


def main():
    args = parse()
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if world > 1:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")

    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    from sspslam_amd.simulator import Simulator
    from sspslam_amd.sharding import ShardedPathIntegration

    dt = 0.001
    if args.sim_block <= 0:
        args.sim_block = args.block
    if args.profile_steps <= 0:
        args.profile_steps = 2 * args.sim_block
    n_total = (args.steps + args.warmup) * args.block
    T_path = max(20.0, (n_total + args.profile_steps + 10) * dt)
    space = H.make_ssp_space(2, args.ssp_dim)
    path, vels = H.make_random_path(T_path, dt=dt, limit=0.1, seed=args.seed)
    pm = H.make_pathint_model(space, path, vels, args.pi_n_neurons, seed=args.seed)
    K = (space.ssp_dim + 1) // 2
    N = K * args.pi_n_neurons

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.time()
    if world == 1:
        model = build(pm.model, dt=dt, n_eval_points=args.eval_points)
        sim = Simulator(None, model=model, dtype=args.dtype, device=local_rank, block_steps=args.sim_block)
        runner = None
    else:
        # ranks that share a GPU (rehearsal only) build one after the other: concurrent rocSOLVER use from
        # several processes on one device was seen to fail; with one GPU per rank all ranks build at once
        shared = world > torch.cuda.device_count()
        runner = None
        for turn in range(world if shared else 1):
            if not shared or turn == rank:
                runner = ShardedPathIntegration(pm, rank, world, dt=dt, dtype=args.dtype, device=local_rank,
                                                n_eval_points=args.eval_points, block=args.block, block_steps=args.sim_block)
            if shared:
                dist.barrier()
        sim, model = runner.sim, runner.model
    build_s = time.time() - t0

    def run_block():
        if runner is None:
            # profile=True: a HIP event pair on the simulator's own stream around every launch of the dominant kernel
            # (two event records per block; the roofline figures below are these launches of the timed region itself)
            sim.run_steps(args.block, profile=True, collect=False)
        else:
            runner.run_block()

    if runner is None:
        sim.prepare(n_total)
    else:
        runner.prepare(n_total)
    for _ in range(args.warmup):
        run_block()
    if runner is not None:
        runner.flush()
    barrier()
    c_warm = sim.counters() if runner is None else None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        run_block()
    if runner is not None:
        runner.flush()                       # rank 0: the read-out of the last block is part of the job
    barrier()
    wall = time.perf_counter() - t0
    if world > 1:
        w = torch.tensor([wall], device="cuda" if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    sim_seconds = args.steps * args.block * dt
    value = sim_seconds / wall

    out = {
        "metric": "sim-sec/wall-sec, SSP-SLAM ssp_dim=1015 10k PI neurons, 1/2/4/8 MI355X",
        "value": round(value, 4), "unit": "sim-sec/wall-sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "config": {"workload": f"PathIntegration 2-D ssp_dim={space.ssp_dim} pi_n_neurons={args.pi_n_neurons}/VCO "
                               f"({K} VCOs, {N} LIF neurons), configs[1]",
                   "timesteps_per_step": args.block, "dt": dt, "eval_points_per_vco": args.eval_points,
                   "parallelism": "1 GPU" if world == 1 else f"VCO-sharded x{world}, all-gather per {args.block} steps ({args.dist_backend})",
                   "build_seconds": round(build_s, 1)},
    }

    if rank == 0 and world == 1:
        # ---- roofline: HIP events around every launch of the dominant kernel in the timed region ------
        c = sim.counters()
        n_timed = c["dominant_launches"] - c_warm["dominant_launches"]
        ms_timed = c["dominant_ms_total"] - c_warm["dominant_ms_total"]
        timed_region = "timed region"
        if n_timed == 0:         # (block size different from the bench step: the block launches were not full blocks)
            sim.run_steps(args.profile_steps, profile=True, collect=False)
            c2 = sim.counters()
            n_timed, ms_timed = c2["dominant_launches"] - c["dominant_launches"], c2["dominant_ms_total"] - c["dominant_ms_total"]
            timed_region = "separate leg after the timed region"
        if n_timed:
            avg_ms = ms_timed / n_timed
            achieved = c["dominant_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
            traffic, traffic_note = None, None
            pmc = os.path.join(ROOT, "profiles", "pmc_traffic.json")
            blocked = c["launches_per_step"] == 0       # whole-block kernel (k_ens_block) vs one k_ensarray per timestep
            if os.path.exists(pmc):          # HBM bytes per launch from separate rocprofv3 --pmc passes of this command
                with open(pmc) as f:
                    t = json.load(f)
                if t.get("units_per_launch") == c["dominant_units_per_launch"] and t.get("dtype") == args.dtype:
                    traffic = t["hbm_bytes_per_launch"]
                    traffic_note = ("measured HBM bytes per launch (rocprofv3 FETCH_SIZE x2 + WRITE_SIZE, separate passes, "
                                    "profiles/pmc_traffic.json): parameters and state are read once per block, not once per timestep")
            out["roofline"] = {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                               "frac": round(achieved / HBM_PEAK_GBS, 4), "traffic": traffic, "traffic_note": traffic_note,
                               "hbm_gbs_measured": round(traffic / (avg_ms * 1e-3) / 1e9, 1) if traffic else None,
                               "kernel": "k_ens_block" if blocked else "k_ensarray", "avg_launch_us": round(avg_ms * 1e3, 2),
                               "algorithmic_bytes_per_launch": c["dominant_bytes_per_launch"],
                               "bytes_per_neuron_step": c["dominant_bytes_per_launch"] / c["dominant_units_per_launch"],
                               "launches_timed": n_timed, "launches_timed_in": timed_region,
                               "launches_per_timestep": c["launches_per_step"]}
            if blocked:
                # VALU-issue roofline of the block kernel: issue slots per neuron-step counted from the ISA
                # (tools/isa_loop_count.py), 4 cycles per wave64 slot on one of 1024 SIMDs at 2.4 GHz
                slots = 29.0           # k_ens_block<float,3,5,20,512,true>: 580 slots per wave-timestep / 20 neurons per lane
                peak_units = 1024 * 2.4e9 * 64 / (4 * slots)
                out["roofline"]["note"] = ("frac > 1: temporal blocking - one launch advances every neuron by "
                                           f"{c['dominant_units_per_launch'] // (K * args.pi_n_neurons)} timesteps from registers, so the "
                                           "per-timestep HBM stream the roofline assumes is not paid; the kernel is VALU-issue bound")
                out["roofline"]["valu"] = {"issue_slots_per_neuron_step": slots, "peak_neuron_steps_per_s": float("%.4g" % peak_units),
                                           "achieved_neuron_steps_per_s": float("%.4g" % (c["dominant_units_per_launch"] / (avg_ms * 1e-3))),
                                           "frac": round(c["dominant_units_per_launch"] / (avg_ms * 1e-3) / peak_units, 3)}
        sim._collect()
        gpu_probe = sim.data[pm.probe]
        # ---- cpu_baseline + parity leg ----------------------------------------------------------------
        if args.cpu_steps > 0:
            from oracle import OracleSimulator
            try:
                from threadpoolctl import threadpool_info
                cores = max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
            except Exception:
                cores = 1
            ref = OracleSimulator(model)
            t0 = time.perf_counter()
            ref.run_steps(args.cpu_steps)
            cpu_wall = time.perf_counter() - t0
            cpu_value = args.cpu_steps * dt / cpu_wall
            want = ref.probe_data(0)
            k = args.cpu_steps
            lo = min(20, k // 2)
            ce = H.cosine_error(gpu_probe[lo:k], want[lo:k])
            out["cpu_baseline"] = {"value": round(cpu_value, 6), "unit": "sim-sec/wall-sec", "cores": int(cores),
                                   "kind": "port",
                                   "sample": f"first {k} timesteps of the same built model, NumPy float64 oracle "
                                             f"(nengo-equivalent merged-operator stepping), {os.cpu_count()} host cpus visible"}
            out["parity"] = {"window_timesteps": k, "max_cosine_error": float(ce.max()),
                             "max_abs_diff": float(np.abs(gpu_probe[:k] - want[:k]).max()), "bar": 1e-3}
            out["gpu_over_cpu"] = round(value / cpu_value, 1)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
