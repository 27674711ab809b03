"""GPU bring-up check: PathIntegration on the HIP backend vs the oracle (f64 tight, f32 loose).
usage: gpu_check_pi.py ssp_dim n_per_vco steps [n_eval_points] [oracle_steps]"""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

d = int(sys.argv[1]) if len(sys.argv) > 1 else 55
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
m_eval = int(sys.argv[4]) if len(sys.argv) > 4 and int(sys.argv[4]) > 0 else None
osteps = int(sys.argv[5]) if len(sys.argv) > 5 else steps
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
t0 = time.time(); bm = build(pm.model, n_eval_points=m_eval); print("build %.2fs" % (time.time() - t0), bm.stats, flush=True)
ref = OracleSimulator(bm); t0 = time.time(); ref.run_steps(osteps); t_or = time.time() - t0
want = ref.probe_data(0)
print("oracle %d steps %.2fs (%.4f sim-s/wall-s)" % (osteps, t_or, osteps * 0.001 / t_or), flush=True)
for dtype in ("f64", "f32"):
    t0 = time.time(); sim = Simulator(None, model=bm, dtype=dtype); print("  create %.2fs" % (time.time() - t0), flush=True)
    sim.prepare(steps)
    t0 = time.time(); sim.run_steps(steps, collect=False); el = time.time() - t0
    sim._collect()
    got = sim.data[pm.probe]
    k = min(osteps, steps)
    ce = H.cosine_error(got[min(20, k // 2):k], want[min(20, k // 2):k])
    c = sim.counters()
    print(dtype, "shape", got.shape, "max|diff| %.3e" % np.abs(got[:k] - want[:k]).max(), "cos err max %.3e mean %.3e" % (ce.max(), ce.mean()),
          "| wall %.3fs device %.1f ms -> %.2f sim-s/wall-s, launches/step %d, dev MB %.0f" % (el, c["last_run_ms"], steps * 0.001 / el, c["launches_per_step"], c["device_bytes"] / 1e6), flush=True)
    sim.run_steps(200, profile=True)
    c = sim.counters()
    if c["dominant_launches"]:
        ms = c["dominant_ms_total"] / c["dominant_launches"]
        print("   dominant kernel avg %.2f us, %.1f GB/s algorithmic; profiled run %.1f us/step" % (ms * 1e3, c["dominant_bytes_per_launch"] / ms / 1e6, c["last_run_ms"] / 200 * 1e3), flush=True)
    sim.close()
