import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from helpers import small_pathint
import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator
print("lib", os.environ.get("SSN_HIP_LIB"))
for d, n, keep_w in ((19, 70, False), (19, 70, True), (19, 600, True), (19, 3000, True), (1015, 70, True)):
    pm = small_pathint(ssp_dim=d, n=n, T=10.0, limit=0.2)
    if keep_w:
        with pm.model:
            pw = nengo.Probe(pm.pathintegrator.oscillators.output, synapse=None)
    model = build(pm.model, n_eval_points=min(1500, max(300, 2 * n)))
    ref = OracleSimulator(model); ref.run_steps(300)
    want = ref.probe_data(0)
    with Simulator(None, model=model, dtype="f32", block_steps=128) as sim:
        sim.run_steps(300)
        got = sim.data[pm.probe]
        c = sim.counters()
        ce = H.cosine_error(got[20:], want[20:])
        extra = ""
        if keep_w:
            gw, ww = sim.data[pw], ref.probe_data(1)
            extra = " osc-output max abs diff %.3e (max %.3e)" % (np.abs(gw - ww).max(), np.abs(ww).max())
        print("d %4d n %5d dout5 %d variant (%d,%d,%d) threads %d launches/step %d: max ce %.3e%s" % (
            d, n, keep_w, c["block_tpb"], c["block_npt"], c["block_enc_lds"], c["block_threads"], c["launches_per_step"], ce.max(), extra), flush=True)
