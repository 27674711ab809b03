"""Where does the build of SLAM config 3 (62 s on the GPU box) go?  cProfile of harness.make_config3_model + builder.build."""
import cProfile, pstats, sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from sspslam_amd import harness as H
from sspslam_amd.builder import build
t0 = time.time()
pr = cProfile.Profile()
pr.enable()
sm = H.make_config3_model()
t1 = time.time()
model = build(sm.model, n_eval_points=4000)
pr.disable()
print("make_config3_model %.1f s, build %.1f s, %d neurons" % (t1 - t0, time.time() - t1, model.n_neurons))
pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
