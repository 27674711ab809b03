"""Long-window sanity of the f32 block kernel: config-2-shaped path integrator for T seconds with the whole-block
kernel and with the per-timestep kernel; both must track the true SSP (cosine to encode(path)) equally well.
usage: gpu_long_run.py ssp_dim n_per_vco T"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator

d, n, T = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(max(T, 10.0), limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
bm = build(pm.model, n_eval_points=4000 if n >= 2000 else None)
steps = int(T / 0.001)
real = s.encode(path[:steps])
outs = {}
for name, flags, dtype in (("block", 0, "f32"), ("step", 128, "f32"), ("f64", 0, "f64")):
    sim = Simulator(None, model=bm, dtype=dtype, flags=flags)
    t0 = time.time(); sim.run_steps(steps); el = time.time() - t0
    out = sim.data[pm.probe]
    sims = np.sum(out * real, axis=1) / np.maximum(np.linalg.norm(out, axis=1), 1e-12)
    outs[name] = (out, sims)
    q = steps // 4
    print("%-5s: %.1f sim-s/wall-s (incl. probe read-back); similarity to the true SSP per quarter: %s, min after 0.2 s %.4f" %
          (name, T / el, " ".join("%.4f" % sims[i * q:(i + 1) * q].mean() for i in range(4)), sims[200:].min()), flush=True)
    sim.close()
ce = H.cosine_error(outs["block"][0][20:], outs["step"][0][20:])
print("block vs step kernel cosine error: first 1 s max %.2e, whole run max %.2e, mean %.2e" % (ce[:980].max(), ce.max(), ce.mean()))
ce = H.cosine_error(outs["block"][0][20:], outs["f64"][0][20:])
print("f32 block kernel vs f64 parity mode (equal to the NumPy oracle to rounding) cosine error: first 1 s max %.2e, whole run max %.2e, mean %.2e"
      % (ce[:980].max(), ce.max(), ce.mean()))
