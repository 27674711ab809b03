import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
os.environ["SSN_DEBUG_PLAN"] = "1"
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=10150, circonv_n_neurons=100, view_rad=0.2)
bm = build(sm.model, n_eval_points=4000)
import collections
print("ops by stage/kind:", collections.Counter((o["stage"], o["kind"]) for o in bm.ops))
for o in bm.ops:
    if o["stage"] == 1:
        print({k: (v if not hasattr(v, "shape") else v.shape) for k, v in o.items() if k in ("kind", "level", "dst", "src", "len", "rows", "cols", "mode", "alpha", "a", "K", "n", "dout", "dft", "x", "j", "out", "err", "act", "spk", "key")})
sim = Simulator(None, model=bm, dtype="f32", steps_per_graph=int(os.environ.get("SSN_SPG", "0")))
sim.prepare(400)
sim.run_steps(100, collect=False)
sim.run_steps(100, profile=2, collect=False)
print(sim.kernel_times())
sim.close()
