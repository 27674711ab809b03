"""SLAMNetwork step-loop timing on the HIP backend, several library flag settings in one process (A/B on one box).

usage: bench_slam.py ssp_dim pi_n mem_n circonv_n steps n_eval flags[,flags...]
"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

d, pi_n, M, c, steps, m_eval = [int(a) for a in sys.argv[1:7]]
flag_sets = [int(f) for f in sys.argv[7].split(",")] if len(sys.argv) > 7 else [0]
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=pi_n, mem_n_neurons=M, circonv_n_neurons=c, view_rad=0.6)
t0 = time.time()
bm = build(sm.model, n_eval_points=m_eval)
print("build %.1fs" % (time.time() - t0), bm.stats, flush=True)
for rep in range(2):
    for fl in flag_sets:
        sim = Simulator(None, model=bm, dtype="f32", flags=fl)
        sim.prepare(2 * steps)
        sim.run_steps(steps, collect=False)
        t0 = time.time()
        sim.run_steps(steps, collect=False)
        el = time.time() - t0
        print("flags %3d: %.1f us/step (%.3f sim-s/wall-s), launches/step %d" %
              (fl, el / steps * 1e6, steps * 1e-3 / el, sim.counters()["launches_per_step"]), flush=True)
        sim.close()
