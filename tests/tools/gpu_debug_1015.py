"""Where does the f64 HIP run leave the oracle at ssp_dim = 1015 (mid-size SLAM)?  Per-step max |diff| of the path
integrator's output, for the default plan and for the plain one-launch-per-operator plan."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

space = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(space, path, vels, n_landmarks=10, pi_n_neurons=100, mem_n_neurons=1200, circonv_n_neurons=20,
                       view_rad=0.6, weights_sample_every=0.05)
with sm.model:
    p_clean = nengo.Probe(sm.slam.gridcells)
    p_osc = nengo.Probe(sm.slam.pathintegrator.oscillators.output)
model = build(sm.model)
N = 60
ref = OracleSimulator(model)
ref.run_steps(N)
want, want_osc = ref.probe_data(0), ref.probe_data(3)
for flags in (0, 4096, 1048576, 131072, 4096 | 1048576, 4096 | 131072, 1048576 | 131072, 4096 | 1048576 | 131072, 8192, 65536, 1024):
    with Simulator(None, model=model, dtype="f64", flags=flags) as sim:
        sim.run_steps(N)
        got, osc = sim.data[sm.probe], sim.data[p_osc]
        lps = sim.counters()["launches_per_step"]
    do = np.abs(osc - want_osc).max(1)
    first = int(np.argmax(do > 1e-12)) if (do > 1e-12).any() else -1
    bad = np.unique(np.nonzero(np.abs(osc[max(first, 0)] - want_osc[max(first, 0)]) > 1e-12)[0] // 3) if first >= 0 else []
    print("flags %8d (%2d launches/step): first bad step %3d, max osc diff %.2e, VCOs off at that step: %s" %
          (flags, lps, first, do.max(), str(list(bad))[:200]), flush=True)
