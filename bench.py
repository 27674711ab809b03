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
``value`` is that device-resident stepping rate; ``value_end_to_end`` times what the reference's timer wraps
(``with sim: sim.run(T)`` followed by ``sim.data[probe]``: input tabulation + upload + stepping + read-back).

With N > 1 (launched by torch.distributed.run, one rank per GPU) the 508 VCO ensembles of the SAME
model are sharded over the ranks (strong scaling): each rank steps its VCOs, the decoded
oscillator states are all-gathered (RCCL) and rank 0 applies the linear read-out.

Extra legs on rank 0 at N = 1 (all outside the timed region):
  * roofline: every launch of the dominant kernel in the timed region is bracketed with HIP events on the
    simulator's stream.  The whole-block kernel (k_ens_block) keeps neuron parameters and state in registers / LDS
    for a block of timesteps, so it is bound by vector-ALU issue, not HBM: ``bound = "valu"``, peak = the issue
    cost of the time loop's own instruction mix (profiles/k_ens_block_isa.json, counted from the gfx950 assembly by
    tools/isa_loop_count.py) at the per-instruction SIMD cycles measured on this GPU by tools/valu_issue_rate.hip
    (profiles/valu_issue_rate.json).  The HBM figures (SURVEY 8d's 52 B per neuron-step streaming basis, and the
    traffic measured by rocprofv3 PMC passes) are kept as secondary fields.
  * cpu_baseline: the NumPy float64 oracle (oracle/stepper.py, a restatement of nengo's reference
    simulator) stepping the same built model on the host cores, 100 warm-up + 200 timed timesteps; the
    GPU trajectory is checked against it on that window (parity, cosine error).
  * slam: BASELINE configs[2] - SLAMNetwork at ssp_dim 1015, 10 000 neurons per VCO, 10 150 memory neurons, 10
    landmarks (reference experiments/run_slam.py:180-235) - stepping rate, launches per timestep, per-kernel device
    time, parity on an oracle window and its own cpu_baseline.
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
N_SIMD = 1024              # 256 CUs x 4 SIMDs
CLOCK_HZ = 2.4e9           # peak shader clock (MI355X_MICROARCH.md); the chip runs this kernel at 2.25-2.4 GHz


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
    ap.add_argument("--cpu-steps", type=int, default=200, help="timed oracle timesteps of the cpu_baseline / parity leg (0 = skip)")
    ap.add_argument("--cpu-warmup", type=int, default=100, help="untimed oracle timesteps before them (BASELINE.md section 2)")
    ap.add_argument("--sim-block", type=int, default=0,
                    help="timesteps per time-batched block inside the simulator (one k_ens_block launch each); "
                         "0 = --block, so that a bench step is exactly one block")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="length of the separate roofline leg that is only run when the timed region had no timed launch of "
                         "the dominant kernel (--sim-block different from --block); 0 = 2 full simulator blocks")
    ap.add_argument("--slam-steps", type=int, default=512, help="timed timesteps of the SLAMNetwork leg (0 = skip the leg)")
    ap.add_argument("--slam-cpu-steps", type=int, default=20, help="timed oracle timesteps of the SLAM leg (after 10 warm-up)")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--plan", default="auto", choices=["auto", "block", "stream"],
                    help="N > 1: plan of each rank's VCO shard - the whole-block kernel (one workgroup per VCO), one streaming launch "
                         "per timestep (an ensemble over several workgroups), or whichever is faster at this shard size (timed on "
                         "one block before the timed region, the slowest rank's time decides)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path with several ranks sharing one GPU (RCCL needs one GPU per rank)")
    return ap.parse_args()


def host_threads():
    try:
        from threadpoolctl import threadpool_info
        return max([i.get("num_threads", 1) for i in threadpool_info()] + [1])
    except Exception:
        return 1


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(p):
        return None
    with open(p) as f:
        return json.load(f)


def valu_roofline(c, units_per_s):
    """VALU-issue roofline of the whole-block kernel from files under profiles/ (see the module docstring)."""
    isa_all, rates = load_json("k_ens_block_isa.json"), load_json("valu_issue_rate.json")
    key = "%d,%d,%d" % (c["block_tpb"], c["block_npt"], c["block_enc_lds"])
    if not isa_all or not rates or key not in isa_all.get("variants", {}):
        return {"note": f"no instruction count for variant {key} under profiles/ (tools/isa_loop_count.py)"}
    isa = isa_all["variants"][key]
    w = str(c["block_tpb"] // 256)                                # waves per SIMD
    cyc = {"packed": rates["v_pk_fma_f32"][w], "trans": rates["v_rcp_f32"][w], "plain_fma": rates["v_fma_f32"][w],
           "plain": rates["v_max_i32"][w]}
    cls = isa["by_class"]
    per_wave_step = sum(cls.get(k, 0) * v for k, v in cyc.items())     # SIMD cycles one wave's timestep needs
    peak = N_SIMD * CLOCK_HZ * 64 * isa["neurons_per_lane"] / per_wave_step
    return {"variant": key, "waves_per_simd": int(w), "valu_instructions_per_wave_timestep": isa["valu_instructions"],
            "by_class": {k: cls.get(k, 0) for k in cyc}, "simd_cycles_per_instruction": cyc,
            "simd_cycles_per_wave_timestep": round(per_wave_step, 1), "valu_per_neuron_step": isa["valu_per_neuron_step"],
            "peak_neuron_steps_per_s": float("%.4g" % peak), "achieved_neuron_steps_per_s": float("%.4g" % units_per_s),
            "frac": round(units_per_s / peak, 3), "clock_hz": CLOCK_HZ, "simds": N_SIMD,
            "sources": ["profiles/k_ens_block_isa.json", "profiles/valu_issue_rate.json"]}


def slam_leg(args, H, build, Simulator, OracleSimulator, dt):
    """BASELINE configs[2]: SLAMNetwork, d = 1015, 10 000 neurons per VCO, M = 10 150, c = 100, 10 landmarks."""
    t0 = time.time()
    space = H.make_ssp_space(2, 1015)
    n_run = args.slam_steps + 64 + 64 + 64
    path, vels = H.make_random_path(max(20.0, (n_run + 10) * dt), dt=dt, limit=0.1, seed=args.seed)
    sm = H.make_slam_model(space, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=10150, circonv_n_neurons=100,
                           view_rad=0.2, seed=args.seed)
    model = build(sm.model, dt=dt, n_eval_points=args.eval_points)
    out = {"workload": "SLAMNetwork 2-D ssp_dim=1015 pi_n_neurons=10000/VCO mem_n_neurons=10150 circonv_n_neurons=100 "
                       f"10 landmarks ({model.n_neurons} neurons), configs[2]", "dtype": "f32"}
    with Simulator(None, model=model, dtype="f32") as sim:
        out["build_seconds"] = round(time.time() - t0, 1)
        sim.prepare(n_run)
        k = max(0, args.slam_cpu_steps) + 10
        sim.run_steps(64, collect=False)                       # (also the parity window: the run starts at t = 0)
        t0 = time.perf_counter()
        sim.run_steps(args.slam_steps, collect=False)
        wall = time.perf_counter() - t0
        c = sim.counters()
        out.update(value=round(args.slam_steps * dt / wall, 4), unit="sim-sec/wall-sec", timesteps_timed=args.slam_steps,
                   us_per_timestep=round(1e6 * wall / args.slam_steps, 2), launches_per_timestep=c["launches_per_step"],
                   device_us_per_timestep=round(1e3 * c["last_run_ms"] / args.slam_steps, 2))
        out["plan"] = ("rounds: every operator in the earliest round its data hazards allow, one heterogeneous grid (k_round) per "
                       "round; the 16 timesteps of a step graph software-pipelined")
        sim.run_steps(64, profile=2, collect=False)            # device time of the pipelined sequence, launch by launch (eager, event pairs)
        kt = sim.kernel_times()
        out["kernels_us_per_timestep"] = {nm: {"launches_per_timestep": round(n / 64, 2), "us": round(1e3 * ms / 64, 2)}
                                          for nm, (n, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1])}
        sim._collect()
        got = sim.data[sm.probe]
        if args.slam_cpu_steps > 0:
            ref = OracleSimulator(model)
            ref.run_steps(10)
            t0 = time.perf_counter()
            ref.run_steps(args.slam_cpu_steps)
            cpu_wall = time.perf_counter() - t0
            want = ref.probe_data([i for i, p in enumerate(model.probes) if p["probe"] is sm.probe][0])
            lo = min(10, k // 2)
            ce = H.cosine_error(got[lo:k], want[lo:k])
            out["cpu_baseline"] = {"value": round(args.slam_cpu_steps * dt / cpu_wall, 6), "unit": "sim-sec/wall-sec",
                                   "cores": int(host_threads()), "kind": "port",
                                   "sample": f"10 warm-up + {args.slam_cpu_steps} timed timesteps of the same built model, NumPy float64 oracle"}
            out["parity"] = {"window_timesteps": k, "max_cosine_error": float(ce.max()),
                             "max_abs_diff": float(np.abs(got[:k] - want[:k]).max()), "bar": 1e-3}
            out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)
    # the same model under the per-operator plan (flag 2097152: one launch per big operator, k_program for the small ones):
    # its event pairs give the device time per kernel, which the round grid does not separate
    with Simulator(None, model=model, dtype="f32", flags=2097152) as sim:
        sim.prepare(64 + 128 + 64)
        sim.run_steps(64, collect=False)
        t0 = time.perf_counter()
        sim.run_steps(128, collect=False)
        wall = time.perf_counter() - t0
        sim.run_steps(64, profile=2, collect=False)
        kt = sim.kernel_times()
        out["per_operator_plan"] = {"us_per_timestep": round(1e6 * wall / 128, 2), "launches_per_timestep": sim.counters()["launches_per_step"],
                                    "kernels_us_per_timestep": {nm: {"launches_per_timestep": round(n / 64, 2), "us": round(1e3 * ms / 64, 2)}
                                                                for nm, (n, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1])}}
    return out


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
                                                n_eval_points=args.eval_points, block=args.block, block_steps=args.sim_block,
                                                flags=128 if args.plan == "stream" else 0)
            if shared:
                dist.barrier()
        plan_seconds = None
        if args.plan == "auto":
            plan_seconds = runner.choose_plan((0, 128), steps=args.block)      # collective; outside the timed region
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
        "value_is": "device-resident stepping rate: inputs tabulated and uploaded before the timed region, probe samples left in HBM",
        "config": {"workload": f"PathIntegration 2-D ssp_dim={space.ssp_dim} pi_n_neurons={args.pi_n_neurons}/VCO "
                               f"({K} VCOs, {N} LIF neurons), configs[1]",
                   "timesteps_per_step": args.block, "dt": dt, "eval_points_per_vco": args.eval_points,
                   "parallelism": "1 GPU" if world == 1 else f"VCO-sharded x{world}, all-gather per {args.block} steps ({args.dist_backend})",
                   "build_seconds": round(build_s, 1)},
    }

    if world > 1:
        out["config"]["shard_plan"] = {"chosen": "streaming launch per timestep (flag 128)" if runner._flags == 128 else "planner default (k_ens_block where a VCO fits a workgroup)",
                                       "vcos_per_rank": runner.hi - runner.lo,
                                       "seconds_per_block": {str(k): round(v, 6) for k, v in (plan_seconds or {}).items()}}
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
            units_per_s = c["dominant_units_per_launch"] / (avg_ms * 1e-3)
            streaming_gbs = c["dominant_bytes_per_launch"] / (avg_ms * 1e-3) / 1e9
            blocked = c["launches_per_step"] == 0       # whole-block kernel (k_ens_block) vs one k_ensarray per timestep
            traffic, hbm = None, {}
            pmc = load_json("pmc_traffic.json")         # HBM bytes per launch from separate rocprofv3 --pmc passes of this command
            kname = "k_ens_block" if blocked else "k_ensarray"
            if pmc and pmc.get("units_per_launch") == c["dominant_units_per_launch"] and pmc.get("dtype") == args.dtype \
                    and kname in pmc.get("kernel", ""):
                traffic = pmc["hbm_bytes_per_launch"]
                hbm = {"measured_bytes_per_launch": traffic, "measured_gbs": round(traffic / (avg_ms * 1e-3) / 1e9, 1),
                       "measured_frac_of_peak": round(traffic / (avg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4),
                       "measured_by": "separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this command (profiles/pmc_traffic.json, "
                                      f"kernel {pmc['kernel']}); not re-measured in this run"}
            hbm.update(algorithmic_bytes_per_launch_streaming_basis=c["dominant_bytes_per_launch"],
                       bytes_per_neuron_step_streaming_basis=c["dominant_bytes_per_launch"] / c["dominant_units_per_launch"],
                       streaming_basis_gbs=round(streaming_gbs, 1), peak_gbs=HBM_PEAK_GBS,
                       streaming_basis_over_peak=round(streaming_gbs / HBM_PEAK_GBS, 3))
            common = {"kernel": kname, "avg_launch_us": round(avg_ms * 1e3, 2), "launches_timed": n_timed,
                      "launches_timed_in": timed_region, "launches_per_timestep": c["launches_per_step"],
                      "units_per_launch": c["dominant_units_per_launch"], "traffic": traffic, "hbm": hbm}
            if blocked:
                v = valu_roofline(c, units_per_s)
                out["roofline"] = {"bound": "valu", "achieved": v.get("achieved_neuron_steps_per_s", float("%.4g" % units_per_s)),
                                   "peak": v.get("peak_neuron_steps_per_s"), "unit": "neuron-steps/s", "frac": v.get("frac"),
                                   **common, "valu": v,
                                   "note": "temporal blocking: one launch advances every neuron by "
                                           f"{c['dominant_units_per_launch'] // (K * args.pi_n_neurons)} timesteps from registers and LDS, so the "
                                           "per-timestep HBM stream of SURVEY 8d is not paid (hbm.streaming_basis_over_peak > 1) and the "
                                           "kernel is bound by vector-ALU issue"}
            else:
                out["roofline"] = {"bound": "hbm", "achieved": round(streaming_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(streaming_gbs / HBM_PEAK_GBS, 4), **common}
        sim._collect()
        gpu_probe = sim.data[pm.probe]
        # ---- end to end: what the reference's timer wraps (run_pathint.py:160-165) + the read-back its next lines do -----
        if not args.no_end_to_end:
            T = args.steps * args.block * dt
            e2e = {}
            for label, strip in (("harness_nodes", False), ("plain_closures", True)):
                sim.reset()
                saved = [tb["fn"] for tb in model.tables]
                if strip:      # the reference scripts pass plain `lambda t: table[int((t - dt) / dt)]` closures: one Python call per timestep
                    for tb in model.tables:
                        tb["fn"] = (lambda f: (lambda t: f(t)))(tb["fn"])
                t0 = time.perf_counter()
                sim.run(T)
                data = sim.data[pm.probe]
                e2e[label] = round(T / (time.perf_counter() - t0), 2)
                for tb, f in zip(model.tables, saved):
                    tb["fn"] = f
                assert data.shape[0] == args.steps * args.block
                sim.clear_probe_data()
            out["value_end_to_end"] = e2e["harness_nodes"]
            out["end_to_end"] = {"unit": "sim-sec/wall-sec", "simulated_seconds": T, "value": e2e["harness_nodes"],
                                 "value_plain_closures": e2e["plain_closures"],
                                 "includes": "sim.run(T) from reset - input-node tabulation, upload, stepping - and sim.data[probe] "
                                             "(read-back of every sample as float64)",
                                 "note": "`value` with the harness's input nodes (vectorised .table twin of the reference closure, same "
                                         "float64 index arithmetic); `value_plain_closures` with the closures wrapped so that every "
                                         "timestep is one Python call, as for the reference scripts' own lambdas"}
            sim.reset()
        # ---- cpu_baseline + parity leg ----------------------------------------------------------------
        if args.cpu_steps > 0:
            from oracle import OracleSimulator
            ref = OracleSimulator(model)
            ref.run_steps(args.cpu_warmup)
            t0 = time.perf_counter()
            ref.run_steps(args.cpu_steps)
            cpu_wall = time.perf_counter() - t0
            cpu_value = args.cpu_steps * dt / cpu_wall
            want = ref.probe_data(0)
            k = min(args.cpu_warmup + args.cpu_steps, gpu_probe.shape[0])
            lo = min(20, k // 2)
            ce = H.cosine_error(gpu_probe[lo:k], want[lo:k])
            out["cpu_baseline"] = {"value": round(cpu_value, 6), "unit": "sim-sec/wall-sec", "cores": int(host_threads()),
                                   "kind": "port",
                                   "sample": f"{args.cpu_warmup} warm-up + {args.cpu_steps} timed timesteps of the same built model, NumPy "
                                             f"float64 oracle (nengo-equivalent merged-operator stepping), {os.cpu_count()} host cpus visible"}
            out["parity"] = {"window_timesteps": k, "max_cosine_error": float(ce.max()),
                             "max_abs_diff": float(np.abs(gpu_probe[:k] - want[:k]).max()), "bar": 1e-3}
            out["gpu_over_cpu"] = round(value / cpu_value, 1)
        sim.close()
        if args.slam_steps > 0 and args.dtype == "f32":
            from oracle import OracleSimulator
            try:
                out["slam"] = slam_leg(args, H, build, Simulator, OracleSimulator, dt)
            except Exception as e:                     # the headline line must not depend on the secondary leg
                out["slam"] = {"error": repr(e)}
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
