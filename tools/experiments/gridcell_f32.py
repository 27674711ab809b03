"""f32 cosine error of the two grid-cell test models under the round plan and the per-operator plan."""
import os, sys
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator
import test_gpu_parity as tp
for net, probes in tp._gridcell_models():
    model = build(net)
    ref = OracleSimulator(model); ref.run_steps(300)
    idx = {id(p["probe"]): i for i, p in enumerate(model.probes)}
    want = ref.probe_data(idx[id(probes[0])])
    for fl in (0, 8388608, 2097152, 0, 8388608, 2097152):
        with Simulator(None, model=model, dtype="f32", flags=fl) as sim:
            sim.run_steps(300)
            ce = H.cosine_error(sim.data[probes[0]][20:], want[20:])
            print("flags", fl, "max ce %.3e at %d; launches %d" % (ce.max(), ce.argmax(), sim.counters()["launches_per_step"]), flush=True)
