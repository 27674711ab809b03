"""Config 2 over a long window: the f32 whole-block kernel vs the f32 per-timestep kernel (flag 128) vs the f64 parity mode
(equal to the NumPy oracle to rounding on the windows the oracle is run for).  usage: long_run_block.py [seconds]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

T = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
dt = 0.001
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(max(20.0, T + 1.0), dt=dt, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, 10000)
bm = build(pm.model, n_eval_points=4000)
n = int(T / dt)
true = s.encode(path[:n]) if hasattr(s, "encode") else None
out = {}
for name, dtype, flags in (("block", "f32", 0), ("step", "f32", 128), ("f64", "f64", 0)):
    with Simulator(None, model=bm, dtype=dtype, flags=flags) as sim:
        t0 = time.perf_counter()
        sim.run_steps(n)
        y = np.array(sim.data[pm.probe])
        el = time.perf_counter() - t0
    out[name] = y
    line = "%-5s: %.1f sim-s/wall-s (incl. probe read-back)" % (name, T / el)
    if true is not None:
        sim_true = np.sum(y * true, axis=1) / (np.linalg.norm(y, axis=1) * np.linalg.norm(true, axis=1) + 1e-300)
        q = n // 4
        line += "; similarity to the true SSP per quarter: " + " ".join("%.4f" % sim_true[i * q:(i + 1) * q].mean() for i in range(4)) + ", min after 0.2 s %.4f" % sim_true[200:].min()
    print(line, flush=True)

def cos_err(a, b):
    return 1.0 - np.sum(a * b, axis=1) / (np.linalg.norm(a, axis=1) * np.linalg.norm(b, axis=1) + 1e-300)

for a, b in (("block", "step"), ("block", "f64"), ("step", "f64")):
    e = cos_err(out[a], out[b])[20:]
    print("%s vs %s cosine error: first 1 s max %.2e, whole run max %.2e, mean %.2e" % (a, b, e[:980].max(), e.max(), e.mean()), flush=True)
