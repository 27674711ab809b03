"""One-off parity checks of PathIntegration shapes the test-suite does not hold: very large ensembles (per-timestep
k_ensarray over many 1024-neuron chunks, as BASELINE config 4 uses it) and config 4's dimension d = 4033 with few neurons."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
for name, space, n, m_eval in (("d=7, n=50000", H.make_ssp_space(2, 7), 50000, 1500),
                               ("d=4033, n=40", H.make_ssp_space(2, n_scales=28, n_rotates=24), 40, None),
                               ("d=55, n=12000", H.make_ssp_space(2, 55), 12000, 1500)):
    t0 = time.time()
    pm = H.make_pathint_model(space, path, vels, n)
    bm = build(pm.model, n_eval_points=m_eval)
    ref = OracleSimulator(bm); ref.run_steps(150)
    want = ref.probe_data(0)
    print("%s (ssp_dim %d): build + oracle %.1fs" % (name, space.ssp_dim, time.time() - t0), flush=True)
    for dtype, flags in (("f64", 0), ("f64", 128), ("f32", 0), ("f32", 128)):
        with Simulator(None, model=bm, dtype=dtype, flags=flags) as sim:
            sim.run_steps(150)
            got = sim.data[pm.probe]
            print("   %s flags %3d: launches/step %d, max|diff| %.3e, cosine error %.3e" % (
                dtype, flags, sim.counters()["launches_per_step"], np.abs(got - want).max(), H.cosine_error(got[20:], want[20:]).max()), flush=True)
