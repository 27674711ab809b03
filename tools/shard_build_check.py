import sys, time; import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.sharding import ShardedPathIntegration
rank, world = int(sys.argv[1]), int(sys.argv[2])
space = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(space, path, vels, 10000, seed=0)
t0=time.time()
r = ShardedPathIntegration(pm, rank, world, dtype="f32", n_eval_points=4000, block=200)
print("built shard", rank, world, "in %.1fs"%(time.time()-t0), r.lo, r.hi, flush=True)
r.prepare(400); r.run_steps(400); 
print("ran; probe", None if r.probe_data() is None else r.probe_data().shape)
