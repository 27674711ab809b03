"""One-off parity check of the 3-D SLAMNetwork (10^6-row clean-up grid) at a larger dimension than the test-suite's
d = 33: f64 vs the oracle, f32 factored / table-pass clean-up vs the oracle's clean-up rows.  usage: gpu_check_slam3d.py [ssp_dim=201] [steps=40]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

d = int(sys.argv[1]) if len(sys.argv) > 1 else 201
N = int(sys.argv[2]) if len(sys.argv) > 2 else 40
t0 = time.time()
space = H.make_ssp_space(3, ssp_dim=d, rng=np.random.default_rng(3))
path, vels = H.make_random_path(10.0, limit=0.5, seed=2, domain_dim=3)
sm = H.make_slam_model(space, path, vels, n_landmarks=20, pi_n_neurons=100, mem_n_neurons=400, circonv_n_neurons=20, view_rad=0.6,
                       weights_sample_every=0.02)
with sm.model:
    p_clean = nengo.Probe(sm.slam.gridcells)
model = build(sm.model)
cl = next(o for o in model.ops if o["kind"] == "cleanup")
print("d = %d: clean-up %d x %d, factors %d x %d x %d; build %.1fs" % (space.ssp_dim, cl["rows"], cl["cols"], cl["grid_rows"], cl["grid_cols"], cl["grid_k2"], time.time() - t0), flush=True)
t0 = time.time()
ref = OracleSimulator(model); ref.run_steps(N)
print("oracle %d steps %.1fs" % (N, time.time() - t0), flush=True)
want, want_clean = ref.probe_data(0), ref.probe_data(2)
with Simulator(None, model=model, dtype="f64") as sim:
    sim.run_steps(N)
    print("f64: output max|diff| %.3e, clean-up max|diff| %.3e" % (np.abs(sim.data[sm.probe] - want).max(), np.abs(sim.data[p_clean] - want_clean).max()), flush=True)
for flags, split in ((0, None), (0, "3"), (524288, None)):
    if split: os.environ["SSN_GRID_SPLIT"] = split
    with Simulator(None, model=model, dtype="f32", flags=flags) as sim:
        os.environ.pop("SSN_GRID_SPLIT", None)
        sim.run_steps(N)
        cc = H.cosine_error(sim.data[p_clean][5:], want_clean[5:])
        print("f32 flags %d split %s: output cosine error %.3e; clean-up rows equal on %.0f %% of the steps, worst cosine error %.3e; launches/step %d" % (
            flags, split, H.cosine_error(sim.data[sm.probe][20:], want[20:]).max(), 100 * (cc < 1e-6).mean(), cc.max(), sim.counters()["launches_per_step"]), flush=True)
