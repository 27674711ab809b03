"""SLAM config 3 (d = 1015, 5.53 M neurons): the f32 fast mode against the f64 parity mode (which equals the NumPy oracle
to rounding on every window the oracle can follow) over a long window, and both against the true SSP of the path.
usage: python tools/experiments/slam_long_run.py [steps]"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=10150, circonv_n_neurons=100, view_rad=0.2)
bm = build(sm.model, n_eval_points=4000)
import sspslam_amd.frontend as fe
am = sm.slam.assomemory
out, W = {}, {}
for dtype in ("f64", "f32"):
    with Simulator(None, model=bm, dtype=dtype) as sim:
        t0 = time.perf_counter()
        sim.run_steps(steps)
        out[dtype] = sim.data[sm.probe]
        W[dtype] = sim.read_buffer(bm.params[am.conn_out].learned_buffer)
        print("%s: %d timesteps in %.2f s (%.1f us per timestep incl. input tabulation and read-back)" % (dtype, steps, time.perf_counter() - t0, 1e6 * (time.perf_counter() - t0) / steps), flush=True)
ce = H.cosine_error(out["f32"][20:], out["f64"][20:])
print("f32 vs f64 over %d timesteps: max cosine error %.3e (bar 1e-3), mean %.3e; per quarter max: %s" %
      (steps, ce.max(), ce.mean(), ", ".join("%.2e" % q.max() for q in np.array_split(ce, 4))))
real = sm.real_ssp[:steps]
for dtype in ("f64", "f32"):
    o = out[dtype]
    sim_true = np.sum(o * real, axis=1) / np.maximum(np.linalg.norm(o, axis=1) * np.linalg.norm(real, axis=1), 1e-12)
    print("%s: similarity of the decoded SSP to the true SSP of the path, per quarter (mean): %s" %
          (dtype, ", ".join("%.4f" % q.mean() for q in np.array_split(sim_true[50:], 4))))

# map recall (definition (i), run_slam.py:263-268): landmark SPs -> built memory activities -> learned PES decoders
rec = {d: H.map_recall(sm.ssp_space, sm.lm_space, bm.params[am.memory], fe.LIF(), W[d]) for d in W}
ce_map = H.cosine_error(rec["f32"][0], rec["f64"][0])
norms = np.linalg.norm(rec["f64"][0], axis=1)
print("map recall after %d timesteps, %d landmarks: recalled-vector norms %.3f .. %.3f; f32 vs f64 cosine error per landmark max %.3e (bar 1e-3); "
      "decoded positions differ by at most %.3e" % (steps, len(norms), norms.min(), norms.max(), ce_map[norms > 1e-6].max() if (norms > 1e-6).any() else 0.0,
                                                    np.abs(rec["f32"][1] - rec["f64"][1]).max()))
print("learned decoders: max |W| %.3e, max |W_f32 - W_f64| %.3e" % (np.abs(W["f64"]).max(), np.abs(W["f32"] - W["f64"]).max()))
