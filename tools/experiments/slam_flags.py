"""Config-3 SLAM timestep under alternative plans: one model build, one simulator per flag set.
usage: python tools/experiments/slam_flags.py [flags[,ENV=value...] ...]      (default: 0 256)
  e.g.  0  0,SSN_ROUND_INTERLEAVE=0  536870912      (default plan | contiguous block order | Stockham FFT)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
flag_sets = sys.argv[1:] or ["0", "256"]
SPG = [int(x) for x in os.environ.get("SSN_SPG", "0").split(",")]
SWEEPS = [x for x in os.environ.get("SSN_SWEEPS", "").split(",")]
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=10150, circonv_n_neurons=100, view_rad=0.2)
bm = build(sm.model, n_eval_points=4000)
for spec, spg, sw in [(f, g, w) for f in flag_sets for g in SPG for w in SWEEPS]:
    parts = spec.split(",")
    fl = int(parts[0])
    envs = dict(kv.split("=", 1) for kv in parts[1:])
    for k, v in envs.items():
        os.environ[k] = v
    os.environ.pop("SSN_ENS_SWEEPS", None)
    if sw:
        os.environ["SSN_ENS_SWEEPS"] = sw
    sim = Simulator(None, model=bm, dtype="f32", flags=fl, steps_per_graph=spg)
    for k in envs:
        os.environ.pop(k, None)
    sim.prepare(1500)
    sim.run_steps(128, collect=False)
    t0 = time.perf_counter()
    sim.run_steps(512, collect=False)
    wall = time.perf_counter() - t0
    c = sim.counters()
    print(f"flags {spec} spg {spg} sweeps {sw or '-'}: {1e6 * wall / 512:.1f} us/step, {c['launches_per_step']} launches", flush=True)
    if os.environ.get("SSN_KT"):
        sim.run_steps(64, profile=2, collect=False)
        kt = sim.kernel_times()
        print("   ", {k: (n // 64, round(1e3 * ms / 64, 1)) for k, (n, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1])}, flush=True)
    sim.close()
