"""SSN_DEBUG_PLAN dump for a config-3-shaped SLAMNetwork (few neurons per VCO: same programs, quick build)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ.setdefault("SSN_DEBUG_PLAN", "1")
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=40, mem_n_neurons=10150, circonv_n_neurons=100, view_rad=0.6)
bm = build(sm.model, n_eval_points=200)
sim = Simulator(None, model=bm, dtype="f32")
sim.close()
