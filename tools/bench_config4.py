"""BASELINE config 4 on ONE GPU: PathIntegration d = 4033 (n_rotates 24, n_scales 28 -> 2017 VCOs), n = 50 000 neurons
per VCO = 1.0e8 LIF neurons (an 8-GPU configuration: 6 GB of parameters + state in f32).  No oracle at this size: the
run is checked against the true SSP of the path.  usage: bench_config4.py [n_per_vco] [steps] [n_eval] [ssp_dim]
(ssp_dim given: HexagonalSSPSpace(ssp_dim=...) as the reference scripts construct it - 4033 yields d = 3751 = 25 x 25 scales
x rotations, reference sspspace.py:683-686 - instead of the n_rotates = 24, n_scales = 28 space that really has 4033 dimensions)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 600
m_eval = int(sys.argv[3]) if len(sys.argv) > 3 else 2000
s = H.make_ssp_space(2, int(sys.argv[4])) if len(sys.argv) > 4 else H.make_ssp_space(2, n_scales=28, n_rotates=24)
print("ssp_dim", s.ssp_dim, flush=True)
path, vels = H.make_random_path(10.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
t0 = time.time()
bm = build(pm.model, n_eval_points=m_eval)
print("build %.1fs" % (time.time() - t0), bm.stats, flush=True)
sim = Simulator(None, model=bm, dtype="f32")
sim.prepare(2 * steps)
sim.run_steps(steps // 2, collect=False)
t0 = time.time(); sim.run_steps(steps, collect=False); el = time.time() - t0
c = sim.counters()
sim.run_steps(64, profile=True, collect=False)
c2 = sim.counters()
out = sim.data[pm.probe]
real = s.encode(path[:out.shape[0]])
sims = np.sum(out * real, axis=1) / np.maximum(np.linalg.norm(out, axis=1), 1e-12)
print("%.3f sim-s/wall-s (%.1f us/step), launches/step %d, device MB %.0f; similarity to the true SSP after 0.2 s: min %.4f mean %.4f" %
      (steps * 1e-3 / el, el / steps * 1e6, c["launches_per_step"], c["device_bytes"] / 1e6, sims[200:].min(), sims[200:].mean()), flush=True)
if c2["dominant_launches"]:
    ms = c2["dominant_ms_total"] / c2["dominant_launches"]
    print("dominant kernel avg %.1f us over %d launches: %.0f GB/s on the 52 B/neuron-step basis" %
          (ms * 1e3, c2["dominant_launches"], c2["dominant_bytes_per_launch"] / ms / 1e6), flush=True)
