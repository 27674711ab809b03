import os, sys
sys.path.insert(0, "/root/repo")
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
sm = H.make_config3_model(seed=0, T=20.0, dt=0.001)
bm = build(sm.model, n_eval_points=4000)
pes = [o for o in bm.ops if o["kind"] == "pes"][0]
print({k: v for k, v in pes.items() if not hasattr(v, "shape")})
sim = Simulator(None, model=bm, dtype="f32")
for n in (100, 300, 600):
    sim.run_steps(n, collect=False)
    act = sim.read_signal(pes["act"], pes["cols"])
    err = sim.read_signal(pes["err"], pes["rows"])
    nz = act != 0
    lines = nz.reshape(-1)[: (nz.size // 16) * 16].reshape(-1, 16).any(axis=1)
    print("after %d steps: act nonzero %.3f, 64-B lines with a nonzero %.3f, err nonzero %.3f, |act|>1e-6: %.3f" % (sim.n_steps, nz.mean(), lines.mean(), (err != 0).mean(), (np.abs(act) > 1e-6).mean()))
sim.close()
