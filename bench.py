#!/usr/bin/env python3
"""bench.py - simulated seconds per wall-clock second of the SSP-SLAM step loop on MI355X.

Metric (BASELINE.json): sim-sec / wall-sec of the simulator step loop (``n_steps*dt / wall(sim.run)``,
build excluded - reference ``experiments/run_pathint.py:160-165``, ``experiments/run_slam.py:230-235``).

    python bench.py --gpus N --steps K --warmup W [--workload pathint|slam]

``--workload pathint`` (default; BASELINE configs[1], the configuration the metric is quoted on): ``PathIntegration`` 2-D,
``ssp_dim=1015``, ``pi_n_neurons=10000`` per VCO (508 VCO ensembles, 5.08 M LIF neurons), synthetic band-limited random
path, seed 0.  ``--workload slam`` (configs[2]): ``SLAMNetwork`` at the same size with 10 150 memory neurons, 10 landmarks.

One "step" = one block of ``--block`` simulator timesteps (default 1000 = one simulated second; 250 for the SLAM
workload) with all inputs already resident in HBM.  The timed region is exactly K blocks between barrier +
torch.cuda.synchronize() pairs - enqueued by one run_steps call on one GPU, as sim.run(T) enqueues them (no host
synchronisation between blocks); the max over ranks is taken; rank 0 prints one JSON line.

``--gpus N`` with N > 1 and no WORLD_SIZE in the environment: this process starts
``python -m torch.distributed.run --nproc-per-node N bench.py ...`` itself - before anything touches the GPU - and relays
rank 0's line (the driver's own torch.distributed.run launch sets WORLD_SIZE and is used as it is).  With N ranks the SAME
model is sharded (strong scaling): the path integrator's VCO ensembles over the ranks with a batched all-gather of the
decoded oscillator states (RCCL) and the linear read-out on rank 0; the SLAMNetwork's ensemble arrays by ensemble and its
dense populations by neuron with ONE all-reduce per timestep, the whole run enqueued on one stream (ssn_phase_async).

Extra legs on rank 0 at N = 1 (all outside the timed region):
  * roofline: every launch of the dominant kernel in the timed region is bracketed with HIP events on the simulator's
    stream.  The whole-block kernel (k_ens_block) keeps neuron parameters and state in registers / LDS for a block of
    timesteps, so it is bound by vector-ALU issue, not HBM: ``bound = "valu"``.  Peak = the time loop's own instruction
    mix (profiles/k_ens_block_isa.json, every mnemonic counted from the gfx950 assembly by tools/isa_loop_count.py) priced
    with issue rates measured IN THIS RUN by ssn_probe_issue_rate (csrc/ssn_probe.hip): nanoseconds of SIMD time per wave64
    instruction from the wall time of a launch of independent instructions, at 1 - 4 resident waves per SIMD - no clock
    constant, no per-wave median.  ``frac`` is quoted against the BEST column per instruction kind; the figure at the
    kernel's own occupancy is next to it.  The HBM figures (SURVEY 8d's streaming basis and the traffic measured by
    rocprofv3 PMC passes) are kept as secondary fields.
  * cpu_baseline: the NumPy float64 oracle (oracle/stepper.py, a restatement of nengo's reference simulator) stepping the
    same built model on the host cores; the GPU trajectory is checked against it on that window (parity, cosine error).
  * slam (pathint workload only): the configs[2] leg - stepping rate, launches per timestep, HBM roofline from the measured
    traffic, parity against the oracle on a window with a landmark in view (PES and Voja live) incl. learned decoders and
    map recall, and its own cpu_baseline.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0      # MI355X HBM3E spec peak (MI355X_MICROARCH.md); ~6300 measured achievable
N_SIMD = 1024              # 256 CUs x 4 SIMDs
METRIC = "sim-sec/wall-sec, SSP-SLAM ssp_dim=1015 10k PI neurons, 1/2/4/8 MI355X"


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10, help="timed blocks")
    ap.add_argument("--warmup", type=int, default=2, help="untimed blocks")
    ap.add_argument("--workload", default="pathint", choices=["pathint", "slam"],
                    help="pathint: BASELINE configs[1] (the headline); slam: configs[2], sharded over the ranks at N > 1")
    ap.add_argument("--block", type=int, default=0, help="simulator timesteps per bench step (0 = 1000 for pathint, 250 for slam)")
    ap.add_argument("--ssp-dim", type=int, default=1015)
    ap.add_argument("--pi-n-neurons", type=int, default=10000)
    ap.add_argument("--mem-n-neurons", type=int, default=0, help="slam: memory / recall / error / ovc population size (0 = 10 * ssp_dim)")
    ap.add_argument("--circonv-n-neurons", type=int, default=100)
    ap.add_argument("--dtype", default="f32", choices=["f32", "f64"])
    ap.add_argument("--eval-points", type=eval_points_arg, default=4000,
                    help="decoder-solve eval points per ensemble, or `default` for nengo's own rule max(clip(500 d, 750, 2500), 2 n) "
                         "= 20000 per 10000-neuron VCO (reference pathintegration.py:162-166 leaves it to nengo).  The stepping cost "
                         "is the same either way; 4000 is a build-time shortcut (~10x fewer rows in every decoder solve), and the "
                         "line says which was used (config.eval_points_per_vco)")
    ap.add_argument("--no-model-cache", action="store_true",
                    help="build every model from scratch (default: built models are kept under $SSN_CACHE_DIR or ~/.cache/sspslam_amd, "
                         "sspslam_amd/modelcache.py; config.build_cache says whether this run hit)")
    ap.add_argument("--cpu-steps", type=int, default=200, help="timed oracle timesteps of the cpu_baseline / parity leg (0 = skip)")
    ap.add_argument("--cpu-warmup", type=int, default=100, help="untimed oracle timesteps before them (BASELINE.md section 2)")
    ap.add_argument("--sim-block", type=int, default=0,
                    help="timesteps per time-batched block inside the simulator (one k_ens_block launch each); "
                         "0 = --block, so that a bench step is exactly one block")
    ap.add_argument("--profile-steps", type=int, default=0,
                    help="length of the separate roofline leg that is only run when the timed region had no timed launch of "
                         "the dominant kernel (--sim-block different from --block); 0 = 2 full simulator blocks")
    ap.add_argument("--slam-steps", type=int, default=512, help="pathint workload: timed timesteps of the SLAMNetwork leg (0 = skip the leg)")
    ap.add_argument("--slam-cpu-steps", type=int, default=150,
                    help="oracle timesteps of the SLAM parity window (a landmark is in view from the first timestep: PES and Voja live); 0 = skip")
    ap.add_argument("--no-end-to-end", action="store_true")
    ap.add_argument("--no-issue-probe", action="store_true", help="price the VALU roofline from profiles/valu_issue_rate.json instead of measuring in the run")
    ap.add_argument("--plan", default="auto", choices=["auto", "block", "stream"],
                    help="N > 1, pathint: plan of each rank's VCO shard - the whole-block kernel (one workgroup per VCO), one streaming launch "
                         "per timestep (an ensemble over several workgroups), or whichever is faster at this shard size (timed on "
                         "one block before the timed region, the slowest rank's time decides)")
    ap.add_argument("--seed", type=int, default=0)
    ap.add_argument("--rehearse-dist", action="store_true",
                    help="one rank, but through the N > 1 code: a process group of world size 1 on --dist-backend, the sharded runner, "
                         "its collectives (what a one-GPU box can exercise of the RCCL path)")
    ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"],
                    help="gloo: rehearsal of the multi-rank path with several ranks sharing one GPU (RCCL needs one GPU per rank)")
    return ap.parse_args()


def eval_points_arg(v):
    """--eval-points: an integer, or `default` (None -> the builder applies nengo's rule per ensemble)."""
    return None if str(v).lower() in ("default", "nengo", "none") else int(v)


def eval_points_label(args):
    return args.eval_points if args.eval_points is not None else f"nengo default: max(clip(500 d, 750, 2500), 2 n) = {max(1500, 2 * args.pi_n_neurons)} per VCO"


def launch_ranks(args):
    """--gpus N > 1 without a torch.distributed environment: start the N ranks as a child (before this process has made
    any GPU call - it never does), relay rank 0's JSON line, exit with the child's status."""
    import socket
    import subprocess
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    p = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, text=True)
    line = None
    for ln in p.stdout.splitlines():
        if ln.startswith("{") and '"metric"' in ln:
            line = ln
        elif ln.strip():
            print(ln, file=sys.stderr)
    if line is not None:
        print(line, flush=True)
    sys.exit(p.returncode if p.returncode else (0 if line is not None else 1))


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


# ---- VALU-issue roofline of the whole-block kernel ------------------------------------------------------------------------
TRANS = ("v_rcp", "v_log", "v_exp", "v_rsq", "v_sqrt", "v_sin", "v_cos")


def issue_kind(op):
    """VALU mnemonic -> the instruction kind ssn_probe_issue_rate measures for it (None: not a vector-ALU instruction)."""
    if not op.startswith("v_"):
        return None
    if op.startswith(TRANS):
        return "trans"
    if op.startswith("v_pk_fma"):
        return "v_pk_fma_f32"
    if op.startswith("v_pk_add"):
        return "v_pk_add_f32"
    if op.startswith("v_pk_"):
        return "v_pk_mul_f32"
    if op.endswith("_dpp") or op.endswith("_sdwa"):
        return "dpp"
    if op.startswith(("v_fma_f32", "v_fmac_f32", "v_mac_f32")):
        return "v_fma_f32"
    if op.startswith(("v_add_f32", "v_sub_f32", "v_subrev_f32", "v_mul_f32")):
        return "v_add_f32"
    if op.startswith("v_mov_b32"):
        return "v_mov_b32"
    if op.startswith(("v_readlane", "v_readfirstlane", "v_writelane")):
        return "lane"
    if op.startswith("v_cndmask"):
        return "v_cndmask_b32"
    return "other"


def measure_issue_rates(device, iters=3000):
    """{kind: {waves per SIMD: ns of SIMD time per wave64 instruction}}, measured now on this GPU (csrc/ssn_probe.hip)."""
    import ctypes as C
    from sspslam_amd import _lib
    lib = _lib.load()
    out, clocks = {}, {}
    for name, kind in _lib.PROBE_KINDS.items():
        out[name] = {}
        for w in (1, 2, 3, 4):
            ns, mhz = C.c_double(), C.c_double()
            rc = lib.ssn_probe_issue_rate(device, kind, w, iters, C.byref(ns), C.byref(mhz))
            if rc != 0:
                raise RuntimeError("ssn_probe_issue_rate: " + _lib.last_error())
            out[name][str(w)] = round(ns.value, 4)
            clocks.setdefault(str(w), []).append(mhz.value)
    return out, {w: [round(min(v)), round(max(v))] for w, v in clocks.items()}


def valu_roofline(c, units_per_s, rates, rates_from, clocks):
    """VALU-issue roofline of the whole-block kernel: its loop's instruction histogram (profiles/k_ens_block_isa.json) priced
    with measured nanoseconds of SIMD time per wave64 instruction."""
    isa_all = load_json("k_ens_block_isa.json")
    key = "%d,%d,%d" % (c["block_tpb"], c["block_npt"], c["block_enc_lds"])
    if not isa_all or not rates or key not in isa_all.get("variants", {}) or "ops" not in isa_all["variants"][key]:
        return {"note": f"no instruction histogram for variant {key} under profiles/ (tools/isa_loop_count.py)"}
    isa = isa_all["variants"][key]
    counts = {}
    for op, n in isa["ops"].items():
        k = issue_kind(op)
        if k is not None:
            counts[k] = counts.get(k, 0) + n
    w_own = str(c["block_tpb"] // 256)                               # waves per SIMD the kernel runs with
    ns_own = sum(n * rates[k][w_own] for k, n in counts.items())     # SIMD ns per wave-timestep at the kernel's own occupancy
    best = {k: min(rates[k].values()) for k in counts}
    ns_best = sum(n * best[k] for k, n in counts.items())            # ... at the best column of every instruction kind
    per_wave = 64 * isa["neurons_per_lane"]
    peak_own, peak_best = N_SIMD * per_wave / (ns_own * 1e-9), N_SIMD * per_wave / (ns_best * 1e-9)
    # the same histogram priced with the issue costs of MI355X_MICROARCH.md instead of rates measured in this run: a packed
    # (VOP3P) instruction 4 cycles of SIMD time, any other VALU instruction 2, a transcendental 8, at the 2.4 GHz peak clock
    # ("report the roofline both ways")
    n_packed = sum(n for k, n in counts.items() if k.startswith("v_pk_"))
    n_trans = counts.get("trans", 0)
    n_plain = sum(counts.values()) - n_packed - n_trans
    ns_guide = (4 * n_packed + 2 * n_plain + 8 * n_trans) / 2.4
    peak_guide = N_SIMD * per_wave / (ns_guide * 1e-9)
    ops = isa.get("ops", {})
    not_valu = {"s_nop": ops.get("s_nop", 0), "s_waitcnt": ops.get("s_waitcnt", 0),
                "lds": sum(n for o, n in ops.items() if o.startswith("ds_")), "s_barrier": ops.get("s_barrier", 0),
                "other_scalar": sum(n for o, n in ops.items() if o.startswith("s_") and o not in ("s_nop", "s_waitcnt", "s_barrier"))}
    return {"variant": key, "waves_per_simd": int(w_own), "valu_instructions_per_wave_timestep": sum(counts.values()),
            "guide_costs": {"cycles_per_instruction": {"packed": 4, "plain": 2, "transcendental": 8}, "clock_ghz": 2.4,
                            "instructions": {"packed": n_packed, "plain": n_plain, "transcendental": n_trans},
                            "simd_ns_per_wave_timestep": round(ns_guide, 1), "peak_neuron_steps_per_s": float("%.4g" % peak_guide),
                            "frac": round(units_per_s / peak_guide, 3)},
            "not_counted_by_the_roofline_per_wave_timestep": not_valu,
            "valu_per_neuron_step": round(sum(counts.values()) / isa["neurons_per_lane"], 2),
            "instructions_by_kind": counts,
            "simd_ns_per_wave_instruction": {k: rates[k] for k in counts},
            "simd_ns_per_wave_timestep": {"kernel_occupancy": round(ns_own, 1), "best_column": round(ns_best, 1)},
            "peak_neuron_steps_per_s": {"best_column": float("%.4g" % peak_best), "kernel_occupancy": float("%.4g" % peak_own)},
            "achieved_neuron_steps_per_s": float("%.4g" % units_per_s),
            "frac": round(units_per_s / peak_best, 3), "frac_at_kernel_occupancy": round(units_per_s / peak_own, 3),
            "issue_rates_from": rates_from, "shader_clock_mhz_during_probe_by_waves": clocks, "simds": N_SIMD,
            "sources": ["profiles/k_ens_block_isa.json (tools/isa_loop_count.py)", rates_from]}


def slam_roofline(us_per_timestep, n_neurons):
    """HBM roofline of the SLAMNetwork timestep: bytes per timestep measured by rocprofv3 PMC passes over every k_round
    dispatch (profiles/slam_config3_traffic.json) over the time of a timestep in this run."""
    t = load_json("slam_config3_traffic.json")
    if not t or t.get("n_neurons") != n_neurons:
        return {"bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None,
                "note": f"no measured traffic under profiles/ for a model of {n_neurons} neurons (slam_config3_traffic.json is configs[2])"}
    gbs = t["hbm_bytes_per_timestep"] / (us_per_timestep * 1e-6) / 1e9
    return {"bound": "hbm", "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
            "traffic": t["hbm_bytes_per_timestep"], "kernel": "k_round (all rounds of a timestep)",
            "algorithmic_bytes_per_timestep_dense_basis": t.get("algorithmic_bytes_per_timestep_dense_basis"),
            "measured_by": t.get("measured_by"), "us_per_timestep": us_per_timestep,
            "note": "traffic is BELOW SURVEY 8d's dense count because decoder products, PES and Voja only touch the rows of "
                    "spiking / active neurons; achieved = measured bytes / time of a timestep in this run"}


def emit(args, out):
    """The bench line: stdout (the descriptor saved by init_dist where a rank's stdout was redirected)."""
    f = getattr(args, "_json_out", None) or sys.stdout
    print(json.dumps(out), file=f, flush=True)


def init_dist(args):
    import torch
    import torch.distributed as dist
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    sharded = world > 1 or args.rehearse_dist
    if sharded:
        # RCCL prints a version banner on stdout when its first communicator comes up: a rank's stdout is pointed at stderr for the
        # whole run and the ONE JSON line goes to the saved descriptor (emit)
        sys.stdout.flush()
        args._json_out = os.fdopen(os.dup(1), "w")
        os.dup2(2, 1)
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29531")
        os.environ.setdefault("RANK", "0")
        os.environ.setdefault("WORLD_SIZE", "1")
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the hot path has no CPU fallback)")
    local_rank = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(local_rank)
    if sharded:
        if args.dist_backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group("gloo")
    return torch, dist, world, rank, local_rank


def slam_parity_and_baseline(args, H, sm, model, sim, OracleSimulator, dt, out):
    """Oracle window from t = 0 (landmark 0 is in view: the landmark inputs, PES and Voja are live) against the samples the
    simulator took on the same timesteps; learned decoders / encoders / map recall (i) at the end of the window."""
    import numpy as np
    import sspslam_amd.frontend as fe
    k = args.slam_cpu_steps
    am = sm.slam.assomemory
    wb, eb = model.params[am.conn_out].learned_buffer, model.params[am.memory].encoder_buffer
    got = sim.data[sm.probe][:k]
    W_gpu, E_gpu = sim.read_buffer(wb), sim.read_buffer(eb)
    ref = OracleSimulator(model)
    warm = min(10, k // 2)
    ref.run_steps(warm)
    t0 = time.perf_counter()
    ref.run_steps(k - warm)
    cpu_wall = time.perf_counter() - t0
    want = ref.probe_data([i for i, p in enumerate(model.probes) if p["probe"] is sm.probe][0])
    W_ref, E_ref, E0 = ref.buf[wb], ref.buf[eb], model.buffers[eb]
    lo = min(20, k // 2)
    ce = H.cosine_error(got[lo:k], want[lo:k])
    out["cpu_baseline"] = {"value": round((k - warm) * dt / cpu_wall, 6), "unit": "sim-sec/wall-sec", "cores": int(host_threads()),
                           "kind": "port", "sample": f"{warm} warm-up + {k - warm} timed timesteps of the same built model, NumPy float64 oracle"}
    rec_g, _ = H.map_recall(sm.ssp_space, sm.lm_space, model.params[am.memory], fe.LIF(), W_gpu)
    rec_r, _ = H.map_recall(sm.ssp_space, sm.lm_space, model.params[am.memory], fe.LIF(), W_ref)
    nr = np.linalg.norm(rec_r, axis=1)
    seen = nr > 1e-3 * max(nr.max(), 1e-300)
    out["parity"] = {"window_timesteps": k, "max_cosine_error": float(ce.max()), "max_abs_diff": float(np.abs(got[:k] - want[:k]).max()), "bar": 1e-3,
                     "learning_live_in_window": bool(np.abs(W_ref).max() > 0 and np.abs(E_ref - E0).max() > 0),
                     "pes_decoders_rel_frobenius_error": float(np.linalg.norm(W_gpu - W_ref) / max(np.linalg.norm(W_ref), 1e-300)),
                     "voja_encoder_shift_rel_frobenius_error": float(np.linalg.norm(E_gpu - E_ref) / max(np.linalg.norm(E_ref - E0), 1e-300)),
                     "map_recall_max_cosine_error": float(H.cosine_error(rec_g[seen], rec_r[seen]).max()) if seen.any() else None,
                     "landmarks_recalled": int(seen.sum())}
    out["gpu_over_cpu"] = round(out["value"] / out["cpu_baseline"]["value"], 1)


def slam_leg(args, H, build, Simulator, OracleSimulator, dt, timed_steps, warm_steps=64, device=0):
    """BASELINE configs[2] on ONE GPU: SLAMNetwork, d = 1015, 10 000 neurons per VCO, M = 10 150, c = 100, 10 landmarks."""
    t0 = time.time()
    k = max(0, args.slam_cpu_steps)
    n_run = max(k, warm_steps) + timed_steps + 64
    sm = H.make_config3_model(seed=args.seed, T=max(20.0, (n_run + 10) * dt), dt=dt, pi_n_neurons=args.pi_n_neurons,
                              mem_n_neurons=args.mem_n_neurons or 10 * args.ssp_dim, circonv_n_neurons=args.circonv_n_neurons,
                              ssp_dim=args.ssp_dim)
    model = build(sm.model, dt=dt, n_eval_points=args.eval_points)
    build_cache = model.stats.get("cache")
    out = {"workload": f"SLAMNetwork 2-D ssp_dim={args.ssp_dim} pi_n_neurons={args.pi_n_neurons}/VCO mem_n_neurons={args.mem_n_neurons or 10 * args.ssp_dim} "
                       f"circonv_n_neurons={args.circonv_n_neurons} 10 landmarks ({model.n_neurons} neurons), configs[2]; landmark 0 placed "
                       "0.07 from the start of the path (in view from t = 0)", "dtype": "f32"}
    with Simulator(None, model=model, dtype="f32", device=device) as sim:
        out["build_seconds"] = round(time.time() - t0, 1)
        out["build_cache"] = build_cache
        sim.prepare(n_run)
        sim.run_steps(max(k, warm_steps), collect=False)          # (the parity window: the run starts at t = 0)
        t0 = time.perf_counter()
        sim.run_steps(timed_steps, collect=False)
        wall = time.perf_counter() - t0
        c = sim.counters()
        out.update(value=round(timed_steps * dt / wall, 4), unit="sim-sec/wall-sec", timesteps_timed=timed_steps,
                   us_per_timestep=round(1e6 * wall / timed_steps, 2), launches_per_timestep=c["launches_per_step"],
                   device_us_per_timestep=round(1e3 * c["last_run_ms"] / timed_steps, 2))
        out["roofline"] = slam_roofline(out["us_per_timestep"], model.n_neurons)
        out["plan"] = ("rounds: every operator in the earliest round its data hazards allow, one heterogeneous grid (k_round) per "
                       "round; the 16 timesteps of a step graph software-pipelined")
        sim.run_steps(64, profile=2, collect=False)            # device time of the pipelined sequence, launch by launch (eager, event pairs)
        kt = sim.kernel_times()
        out["kernels_us_per_timestep"] = {nm: {"launches_per_timestep": round(n / 64, 2), "us": round(1e3 * ms / 64, 2)}
                                          for nm, (n, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1])}
    if k > 0:
        # parity: a fresh simulator stepped for exactly the oracle's window (learned state is compared at its end)
        with Simulator(None, model=model, dtype="f32", device=device) as sim:
            sim.run_steps(k)
            slam_parity_and_baseline(args, H, sm, model, sim, OracleSimulator, dt, out)
    return out


def slam_main(args):
    """--workload slam: configs[2] as the headline object, sharded over the ranks at N > 1 (ShardedSLAM)."""
    torch, dist, world, rank, local_rank = init_dist(args)
    from sspslam_amd import harness as H
    from sspslam_amd.modelcache import cached_build
    from sspslam_amd.simulator import Simulator
    from sspslam_amd.sharding import ShardedSLAM
    build = (lambda net, **kw: cached_build(net, cache=False if args.no_model_cache else None, **kw))
    dt = 0.001
    block = args.block or 250
    n_total = (args.steps + args.warmup) * block

    sharded = world > 1 or args.rehearse_dist

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    out = {"metric": METRIC, "unit": "sim-sec/wall-sec", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
           "higher_is_better": True, "scaling": "strong", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
           "value_is": "device-resident stepping rate: inputs tabulated and uploaded before the timed region, probe samples left in HBM"}
    if not sharded:
        from oracle import OracleSimulator
        leg = slam_leg(args, H, build, Simulator, OracleSimulator, dt, timed_steps=args.steps * block, warm_steps=args.warmup * block,
                       device=local_rank)
        wall = leg["timesteps_timed"] * dt / leg["value"]
        out.update(value=leg["value"], ms_per_step=round(1e3 * wall / args.steps, 4),
                   config={"workload": leg["workload"], "timesteps_per_step": block, "dt": dt, "eval_points_per_vco": eval_points_label(args),
                           "parallelism": "1 GPU", "build_seconds": leg["build_seconds"], "build_cache": leg.get("build_cache")},
                   roofline=leg["roofline"], slam=leg)
        for key in ("cpu_baseline", "parity", "gpu_over_cpu"):
            if key in leg:
                out[key] = leg[key]
        emit(args, out)
        return
    t0 = time.time()
    sm = H.make_config3_model(seed=args.seed, T=max(20.0, (n_total + 10) * dt), dt=dt, pi_n_neurons=args.pi_n_neurons,
                              mem_n_neurons=args.mem_n_neurons or 10 * args.ssp_dim, circonv_n_neurons=args.circonv_n_neurons,
                              ssp_dim=args.ssp_dim)
    shared = world > torch.cuda.device_count()
    r = None
    for turn in range(world if shared else 1):          # ranks sharing a GPU (rehearsal) build one after the other (rocSOLVER)
        if not shared or turn == rank:
            r = ShardedSLAM(sm, rank, world, dt=dt, dtype="f32", device=local_rank, n_eval_points=args.eval_points)
        if shared:
            dist.barrier()
    build_s = time.time() - t0
    r.prepare(n_total)
    for _ in range(args.warmup):
        r.run_steps(block)
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        r.run_steps(block)
    barrier()
    wall = time.perf_counter() - t0
    w = torch.tensor([wall], device="cuda" if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    wall = float(w.item())
    n_ex = sum(hi - lo for lo, hi in r.model.exchange)
    out.update(value=round(args.steps * block * dt / wall, 4), ms_per_step=round(1e3 * wall / args.steps, 4),
               config={"workload": f"SLAMNetwork 2-D ssp_dim={args.ssp_dim} pi_n_neurons={args.pi_n_neurons}/VCO "
                                   f"mem_n_neurons={args.mem_n_neurons or 10 * args.ssp_dim} circonv_n_neurons={args.circonv_n_neurons} 10 landmarks, configs[2]",
                       "timesteps_per_step": block, "dt": dt, "eval_points_per_vco": eval_points_label(args), "build_seconds": round(build_s, 1),
                       "parallelism": f"neuron-sharded x{world}: ensemble arrays by ensemble, dense populations by neuron, one all-reduce of "
                                      f"{n_ex} values per timestep ({args.dist_backend}), "
                                      + ("whole run enqueued on one stream (ssn_phase_async)" if r._stream_ordered() else "host loop (gloo rehearsal)")},
               us_per_timestep=round(1e6 * wall / (args.steps * block), 2), launches_per_timestep=r.sim.counters()["launches_per_step"])
    r.close()
    if rank == 0:
        emit(args, out)
    dist.destroy_process_group()


def pathint_main(args):
    import numpy as np
    torch, dist, world, rank, local_rank = init_dist(args)
    from sspslam_amd import harness as H
    from sspslam_amd.modelcache import cached_build
    from sspslam_amd.simulator import Simulator
    from sspslam_amd.sharding import ShardedPathIntegration
    build = (lambda net, **kw: cached_build(net, cache=False if args.no_model_cache else None, **kw))

    dt = 0.001
    args.block = args.block or 1000
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

    sharded = world > 1 or args.rehearse_dist

    def barrier():
        if sharded:
            dist.barrier()
        torch.cuda.synchronize()

    t0 = time.time()
    if not sharded:
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

    if runner is None:
        sim.prepare(n_total)
    else:
        runner.prepare(n_total)
    def run_blocks(k):
        if runner is None:
            # one call enqueues all k blocks (as sim.run(T) does); a call per block would put a host synchronisation and a cold
            # launch queue behind every block: 3.09 vs 3.03 ms per block (tools/experiments/sync_per_block.py).  profile=True: a
            # HIP event pair on the simulator's own stream around every launch of the dominant kernel
            if k > 0:
                sim.run_steps(k * args.block, profile=True, collect=False)
        else:
            if k > 0:
                runner.run_steps(k * args.block)       # (device path: one simulator call per exchange interval, not per block)

    run_blocks(args.warmup)
    if runner is not None:
        runner.flush()
    barrier()
    c_warm = sim.counters() if runner is None else None
    t0 = time.perf_counter()
    run_blocks(args.steps)
    if runner is not None:
        runner.flush()                       # rank 0: the read-out of the last block is part of the job
    barrier()
    wall = time.perf_counter() - t0
    if sharded:
        w = torch.tensor([wall], device="cuda" if args.dist_backend == "nccl" else "cpu", dtype=torch.float64)
        dist.all_reduce(w, op=dist.ReduceOp.MAX)
        wall = float(w.item())
    sim_seconds = args.steps * args.block * dt
    value = sim_seconds / wall

    out = {
        "metric": METRIC,
        "value": round(value, 4), "unit": "sim-sec/wall-sec", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "ms_per_step": round(1e3 * wall / args.steps, 4), "higher_is_better": True,
        "scaling": "strong", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
        "value_is": "device-resident stepping rate: inputs tabulated and uploaded before the timed region, probe samples left in HBM",
        "config": {"workload": f"PathIntegration 2-D ssp_dim={space.ssp_dim} pi_n_neurons={args.pi_n_neurons}/VCO "
                               f"({K} VCOs, {N} LIF neurons), configs[1]",
                   "timesteps_per_step": args.block, "dt": dt, "eval_points_per_vco": eval_points_label(args),
                   "parallelism": "1 GPU" if not sharded else f"VCO-sharded x{world}, all-gather per {args.block * (runner.gather_every if runner._device_exchange() else 1)} steps ({args.dist_backend})",
                   "build_seconds": round(build_s, 1), "build_cache": (model.stats.get("cache") if model is not None else None)},
    }

    if sharded:
        out["config"]["shard_plan"] = {"chosen": "streaming launch per timestep (flag 128)" if runner._flags == 128 else "planner default (k_ens_block where a VCO fits a workgroup)",
                                       "vcos_per_rank": runner.hi - runner.lo,
                                       "seconds_per_block": {str(k): round(v, 6) for k, v in (plan_seconds or {}).items()}}
    if rank == 0 and not sharded:
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
                rates, rates_from, clocks = None, None, None
                if not args.no_issue_probe:
                    try:
                        rates, clocks = measure_issue_rates(local_rank)
                        rates_from = "ssn_probe_issue_rate in this run (csrc/ssn_probe.hip): wall time of a launch of independent instructions"
                    except Exception as e:               # noqa: BLE001 - fall back to the checked-in measurement, and say so
                        rates_from = f"in-run probe failed ({e!r}); "
                if rates is None:
                    vir = load_json("valu_issue_rate.json") or {}
                    ns = vir.get("ns", {})
                    name_of = {"trans": "v_rcp_f32", "dpp": "v_add_f32_dpp", "lane": "v_readlane_b32", "other": "v_max_i32"}
                    rates = {k: ns[name_of.get(k, k)] for k in ("v_pk_fma_f32", "v_pk_mul_f32", "v_pk_add_f32", "trans", "v_fma_f32", "v_add_f32",
                                                               "dpp", "v_mov_b32", "lane", "v_cndmask_b32", "other") if name_of.get(k, k) in ns} or None
                    rates_from = (rates_from or "") + "profiles/valu_issue_rate.json (tools/valu_issue_rate.hip, kernel-time basis)"
                v = valu_roofline(c, units_per_s, rates, rates_from, clocks)
                peak = (v.get("peak_neuron_steps_per_s") or {}).get("best_column")
                out["roofline"] = {"bound": "valu", "achieved": v.get("achieved_neuron_steps_per_s", float("%.4g" % units_per_s)),
                                   "peak": peak, "unit": "neuron-steps/s", "frac": v.get("frac"),
                                   "frac_at_guide_issue_costs": (v.get("guide_costs") or {}).get("frac"),
                                   **common, "valu": v,
                                   "note": "temporal blocking: one launch advances every neuron by "
                                           f"{c['dominant_units_per_launch'] // (K * args.pi_n_neurons)} timesteps from registers and LDS, so the "
                                           "per-timestep HBM stream of SURVEY 8d is not paid (hbm.streaming_basis_over_peak > 1) and the "
                                           "kernel is bound by vector-ALU issue; peak = the loop's instruction histogram priced at the best "
                                           "measured issue rate of every instruction kind (frac_at_kernel_occupancy: at the kernel's own "
                                           "waves per SIMD)"}
            else:
                out["roofline"] = {"bound": "hbm", "achieved": round(streaming_gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                                   "frac": round(streaming_gbs / HBM_PEAK_GBS, 4), **common}
        sim._collect()
        gpu_probe_view = sim.data[pm.probe]
        gpu_probe = np.array(gpu_probe_view[:max(1, args.cpu_warmup + args.cpu_steps)])      # (the parity window; the rest is released)
        # ---- end to end: what the reference's timer wraps (run_pathint.py:160-165) + the read-back its next lines do -----
        if not args.no_end_to_end:
            T = args.steps * args.block * dt
            e2e = {}
            del gpu_probe_view
            data = None
            e2e_runs = {"harness_nodes": [], "plain_closures": []}
            for label, strip in (("warm_up", False),) + (("harness_nodes", False), ("plain_closures", True)) * 3:
                # (the first pass is untimed: it pays the first-use allocations of the read-back path - pinned staging, one
                #  array per probe - as the W warm-up blocks of the timed region pay the step loop's; every leg is then run three
                #  times - a whole run is ~70 ms of wall time with Python threads in it, 4 % apart from run to run - and the
                #  MEDIAN is reported, all three listed beside it)
                sim.reset()
                saved = [tb["fn"] for tb in model.tables]
                if strip:      # the reference scripts pass plain `lambda t: table[int((t - dt) / dt)]` closures: one Python call per timestep
                    for tb in model.tables:
                        tb["fn"] = (lambda f: (lambda t: f(t)))(tb["fn"])
                t0 = time.perf_counter()
                sim.run(T)
                data = sim.data[pm.probe]
                e2e[label] = round(T / (time.perf_counter() - t0), 2)
                if label in e2e_runs:
                    e2e_runs[label].append(e2e[label])
                for tb, f in zip(model.tables, saved):
                    tb["fn"] = f
                assert data.shape[0] == args.steps * args.block
                # the result is released OUTSIDE the timed region (the pass before this change timed `data = ...` while the
                # previous pass's 162 MB array was still bound to the name: ~9 ms of munmap inside the timer)
                sim.clear_probe_data()
                data = None
            for label in e2e_runs:
                e2e[label] = sorted(e2e_runs[label])[len(e2e_runs[label]) // 2]
            out["value_end_to_end"] = e2e["harness_nodes"]
            out["end_to_end"] = {"unit": "sim-sec/wall-sec", "simulated_seconds": T, "value": e2e["harness_nodes"],
                                 "value_plain_closures": e2e["plain_closures"],
                                 "runs": e2e_runs, "statistic": "median of three runs per leg",
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
                out["slam"] = slam_leg(args, H, build, Simulator, OracleSimulator, dt, timed_steps=args.slam_steps, device=local_rank)
            except Exception as e:                     # the headline line must not depend on the secondary leg
                out["slam"] = {"error": repr(e)}
    if rank == 0:
        emit(args, out)
    if sharded:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        launch_ranks(args)                   # does not return
    if args.workload == "slam":
        slam_main(args)
    else:
        pathint_main(args)


if __name__ == "__main__":
    main()
