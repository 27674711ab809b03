"""BASELINE config 5 on ONE GPU: SLAMNetwork over a three-dimensional domain, ssp_dim=2047 (-> 1801: 15 scales x
15 rotations of the 3-D simplex), 20 landmarks.  The reference hard-codes 100 sample points per axis for the clean-up
(slam.py:209), i.e. a 10^6 x 1801 table (7.2 GB in f32) that is read in full every timestep; the phase matrix uses
seeded rotations here (the reference's are unseeded, sspspace.py:725).  No oracle at this size: the run is checked
against the true SSP of the path.

usage: bench_config5.py [ssp_dim] [pi_n] [steps] [n_eval] [build-only]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build

import threading


def _heartbeat(t_start=time.time()):
    while True:
        time.sleep(60)
        print("  ... %d s" % (time.time() - t_start), flush=True)


threading.Thread(target=_heartbeat, daemon=True).start()
ssp_dim = int(sys.argv[1]) if len(sys.argv) > 1 else 2047
pi_n = int(sys.argv[2]) if len(sys.argv) > 2 else 800
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 300
m_eval = int(sys.argv[4]) if len(sys.argv) > 4 else 2000
build_only = len(sys.argv) > 5
t0 = time.time()
s = H.make_ssp_space(3, ssp_dim, rng=np.random.default_rng(0))
d = s.ssp_dim
print("ssp_dim %d, VCOs %d" % (d, (d + 1) // 2), flush=True)
path, vels = H.make_random_path(10.0, limit=0.1, seed=0, domain_dim=3)
sm = H.make_slam_model(s, path, vels, n_landmarks=20, pi_n_neurons=pi_n, mem_n_neurons=10 * d, circonv_n_neurons=100,
                       view_rad=0.6)
print("network %.1fs (clean-up table %s, %.2f GB f64)" % (time.time() - t0, sm.slam.sample_ssps.shape,
                                                          sm.slam.sample_ssps.nbytes / 1e9), flush=True)
t0 = time.time()
bm = build(sm.model, n_eval_points=m_eval)
print("build %.1fs" % (time.time() - t0), bm.stats, flush=True)
if build_only:
    sys.exit(0)
from sspslam_amd.simulator import Simulator
for flags in [int(f) for f in os.environ.get("SSN_FLAGS", "0").split(",")]:      # plan switches to compare in one process
    t0 = time.time()
    sim = Simulator(None, model=bm, dtype="f32", flags=flags)
    sim.prepare(2 * steps + 64)
    print("flags %d: upload + tabulate %.1fs" % (flags, time.time() - t0), flush=True)
    sim.run_steps(steps, collect=False)
    t0 = time.time(); sim.run_steps(steps, collect=False); el = time.time() - t0
    c = sim.counters()
    out = sim.data[sm.probe]
    real = sm.real_ssp[:out.shape[0]]
    sims = np.sum(out * real, axis=1) / np.maximum(np.linalg.norm(out, axis=1), 1e-12)
    table_gb = sm.slam.sample_ssps.size * 4 / 1e9
    print("flags %d: %.3f sim-s/wall-s (%.1f us/step), launches/step %d, device GB %.1f; clean-up table %.2f GB (a pass over it every step would need %.0f GB/s); "
          "similarity to the true SSP after 0.2 s: min %.4f mean %.4f" %
          (flags, steps * 1e-3 / el, el / steps * 1e6, c["launches_per_step"], c["device_bytes"] / 1e9, table_gb,
           table_gb / (el / steps), sims[200:].min(), sims[200:].mean()), flush=True)
    if os.environ.get("SSN_KT"):
        sim.run_steps(32, profile=2, collect=False)
        print("   ", {k: (n // 32, round(1e3 * ms / 32, 1)) for k, (n, ms) in sorted(sim.kernel_times().items(), key=lambda kv: -kv[1][1])}, flush=True)
    sim.close()
