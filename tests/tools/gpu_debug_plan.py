"""Print the generic plan's items and dependency edges (SSN_DEBUG_PLAN) for a small SLAMNetwork, and time flags 0 vs 256."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("SSN_DEBUG_PLAN", "1")
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
s = H.make_ssp_space(2, 55)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=5, pi_n_neurons=100, mem_n_neurons=550, circonv_n_neurons=50, view_rad=0.6)
bm = build(sm.model)
spg = int(sys.argv[1]) if len(sys.argv) > 1 else 0
for fl in (256, 0):
    sim = Simulator(None, model=bm, dtype="f32", flags=fl, steps_per_graph=spg)
    os.environ.pop("SSN_DEBUG_PLAN", None)
    sim.prepare(4000)
    sim.run_steps(800, collect=False)
    t0 = time.time(); sim.run_steps(1600, collect=False); el = time.time() - t0
    print("flags %d: %.1f us/step" % (fl, el / 1600 * 1e6), flush=True)
    sim.close()
