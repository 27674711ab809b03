"""Config 2: twenty 1000-step blocks as twenty run_steps calls (a host synchronisation behind each) vs one call."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(70.0, dt=0.001, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, 10000)
bm = build(pm.model, n_eval_points=4000)
sim = Simulator(None, model=bm, dtype="f32", block_steps=1000)
sim.prepare(66000)
for _ in range(5):
    sim.run_steps(1000, collect=False)
for rep in range(2):
    t0 = time.perf_counter()
    for _ in range(20):
        sim.run_steps(1000, collect=False)
    a = time.perf_counter() - t0
    t0 = time.perf_counter()
    sim.run_steps(20000, collect=False)
    b = time.perf_counter() - t0
    print("20 calls of 1000 steps: %.3f ms per block (%.1f sim-s/wall-s); one call of 20000: %.3f ms per block (%.1f)" % (a / 20 * 1e3, 20 / a, b / 20 * 1e3, 20 / b), flush=True)
