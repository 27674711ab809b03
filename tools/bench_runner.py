"""Host overhead of the sharded runner: ShardedPathIntegration(world=1) vs the bare simulator, config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.sharding import ShardedPathIntegration
space = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(space, path, vels, int(sys.argv[1]) if len(sys.argv) > 1 else 10000, seed=0)
dev = len(sys.argv) > 2 and sys.argv[2] == "device"        # all-device block exchange (what RCCL ranks use)
r = ShardedPathIntegration(pm, 0, 1, dtype="f32", n_eval_points=1000, block=1000, device_exchange=dev)
r.prepare(12000)
r.run_block(); r.flush()
for label in ("async",):
    t0 = time.perf_counter()
    for _ in range(8):
        t1 = time.perf_counter(); r.run_block(); print("  block %.1f ms" % ((time.perf_counter() - t1) * 1e3), flush=True)
    r.flush()
    el = time.perf_counter() - t0
    print("runner: %.2f ms per 1000-step block -> %.1f sim-s/wall-s" % (el / 8 * 1e3, 8.0 / el))
t0 = time.perf_counter(); r.sim.run_steps(2000, collect=False); el = time.perf_counter() - t0
print("bare core sim: %.2f ms per 1000 steps" % (el / 2 * 1e3))
