"""GPU bring-up check: PI cfg1-like model on the HIP backend vs the oracle (f64 tight, f32 loose)."""
import sys, time, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator

d = int(sys.argv[1]) if len(sys.argv) > 1 else 55
n = int(sys.argv[2]) if len(sys.argv) > 2 else 500
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
s = H.make_ssp_space(2, d)
T = max(1.0, steps * 0.001)
path, vels = H.make_random_path(T, limit=2.0, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
t0 = time.time(); bm = build(pm.model); print("build %.2fs" % (time.time() - t0), bm.stats, flush=True)
ref = OracleSimulator(bm); t0 = time.time(); ref.run_steps(steps); t_or = time.time() - t0
want = ref.probe_data(0)
print("oracle %.2fs (%.3f sim-s/wall-s)" % (t_or, steps * 0.001 / t_or), flush=True)
for dtype in ("f64", "f32"):
    sim = Simulator(None, model=bm, dtype=dtype)
    sim.prepare(steps)
    t0 = time.time(); sim.run_steps(steps); el = time.time() - t0
    got = sim.data[pm.probe]
    ce = H.cosine_error(got[20:], want[20:])
    c = sim.counters()
    print(dtype, "shape", got.shape, "max|diff| %.3e" % np.abs(got - want).max(), "cos err max %.3e mean %.3e" % (ce.max(), ce.mean()),
          "| wall %.3fs device %.1f ms -> %.1f sim-s/wall-s, launches/step %d" % (el, c["last_run_ms"], steps * 0.001 / el, c["launches_per_step"]), flush=True)
    # second run + profile leg
    sim.run_steps(200, profile=True)
    c = sim.counters()
    if c["dominant_launches"]:
        ms = c["dominant_ms_total"] / c["dominant_launches"]
        print("   dominant kernel avg %.2f us, %.1f GB/s algorithmic" % (ms * 1e3, c["dominant_bytes_per_launch"] / ms / 1e6))
    sim.close()
