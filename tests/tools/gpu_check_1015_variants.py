"""One-off parity checks at ssp_dim = 1015 with few neurons per population: SLAMViewNetwork and SLAMNetwork with a
grid-cell population, f64 vs the oracle (max |diff| of every signal probe) and f32 cosine error."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
N = 80
for name in ("slamview", "slam+gridcells"):
    t0 = time.time()
    if name == "slamview":
        sm = H.make_slamview_model(s, path, vels, n_landmarks=10, pi_n_neurons=100, mem_n_neurons=1200, view_rad=0.6, weights_sample_every=0.05)
        probes = [sm.probe, sm.recall_probe]
    else:
        import inspect
        kw = dict(n_landmarks=10, pi_n_neurons=100, mem_n_neurons=1200, circonv_n_neurons=20, view_rad=0.6, weights_sample_every=0.05)
        if "gc_n_neurons" in inspect.signature(H.make_slam_model).parameters:
            kw["gc_n_neurons"] = 1500
        else:
            print("make_slam_model has no gc_n_neurons option: plain SLAM instead", flush=True)
        sm = H.make_slam_model(s, path, vels, **kw)
        probes = [sm.probe]
    model = build(sm.model)
    ref = OracleSimulator(model); ref.run_steps(N)
    print("%s: build + oracle %.1fs" % (name, time.time() - t0), flush=True)
    with Simulator(None, model=model, dtype="f64") as sim:
        sim.run_steps(N)
        for j, p in enumerate(probes):
            print("  f64 probe %d max|diff| %.3e" % (j, np.abs(sim.data[p] - ref.probe_data(j)).max()), flush=True)
    with Simulator(None, model=model, dtype="f32") as sim:
        sim.run_steps(N)
        print("  f32 cosine error %.3e" % H.cosine_error(sim.data[probes[0]][20:], ref.probe_data(0)[20:]).max(), flush=True)
