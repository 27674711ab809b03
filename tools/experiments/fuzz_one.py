"""One seed of the random-network generator in f32 against the oracle, probe by probe."""
import sys, os
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import random_network
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator
seed = int(sys.argv[1])
net, probes = random_network(seed)
model = build(net)
for o in model.ops:
    print({k: (v if not hasattr(v, "shape") else v.shape) for k, v in o.items() if k in ("kind", "stage", "dst", "src", "len", "rows", "cols", "mode", "n", "K", "lr")})
ref = OracleSimulator(model); ref.run_steps(150)
for dtype in ("f64", "f32"):
    with Simulator(None, model=model, dtype=dtype) as sim:
        sim.run_steps(150)
        for p in probes:
            q = [i for i, mp in enumerate(model.probes) if mp["probe"] is p][0]
            want, got = ref.probe_data(q), sim.data[p]
            big = np.linalg.norm(want, axis=1) > 0.05
            ce = (1 - np.sum(want[big] * got[big], axis=1) / (np.linalg.norm(want[big], axis=1) * np.linalg.norm(got[big], axis=1))).max() if big.sum() else -1
            print(dtype, "probe", q, p.obj, "synapse", p.synapse, "shape", want.shape, "max abs diff %.3e" % np.abs(want - got).max(), "max |want| %.3f" % np.abs(want).max(), "cos err %.3e" % ce, "first diff step", int(np.argmax(np.abs(want - got).max(axis=1) > 1e-4)))
