"""SLAMNetwork on the HIP backend vs the oracle.  usage: gpu_check_slam.py ssp_dim pi_n mem_n circonv_n steps oracle_steps [n_eval]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from collections import Counter
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

d, pi_n, M, c, steps, osteps = [int(a) for a in sys.argv[1:7]]
m_eval = int(sys.argv[7]) if len(sys.argv) > 7 else None
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
t0 = time.time()
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=pi_n, mem_n_neurons=M, circonv_n_neurons=c, view_rad=0.6)
print("construct %.1fs" % (time.time() - t0), flush=True)
t0 = time.time(); bm = build(sm.model, n_eval_points=m_eval); print("build %.1fs" % (time.time() - t0), bm.stats, flush=True)
print(Counter(o["kind"] for o in bm.ops), "stages", Counter(o["stage"] for o in bm.ops), flush=True)
if osteps > 0:
    ref = OracleSimulator(bm); t0 = time.time(); ref.run_steps(osteps); t_or = time.time() - t0
    want = ref.probe_data(0)
    print("oracle %d steps %.2fs (%.5f sim-s/wall-s)" % (osteps, t_or, osteps * 0.001 / t_or), flush=True)
for dtype in (("f64", "f32") if osteps > 0 else ("f32",)):
    t0 = time.time(); sim = Simulator(None, model=bm, dtype=dtype); print("  create %.2fs" % (time.time() - t0), flush=True)
    t0 = time.time(); sim.prepare(steps); print("  prepare %.2fs" % (time.time() - t0), flush=True)
    t0 = time.time(); sim.run_steps(steps, collect=False); el = time.time() - t0
    sim._collect()
    got = sim.data[sm.probe]
    k = min(osteps, steps)
    lo = min(20, k // 2)
    ce = H.cosine_error(got[lo:k], want[lo:k]) if k else np.zeros(1)
    if not k:
        want = got
    cc = sim.counters()
    print(dtype, "max|diff| %.3e cos err max %.3e | wall %.3fs -> %.3f sim-s/wall-s (%.1f us/step), launches/step %d, dev MB %.0f" %
          (np.abs(got[:k] - want[:k]).max(), ce.max(), el, steps * 0.001 / el, el / steps * 1e6, cc["launches_per_step"], cc["device_bytes"] / 1e6), flush=True)
    sim.close()
