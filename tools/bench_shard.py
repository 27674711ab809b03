"""Per-rank step time of the VCO-sharded path integrator, measured on one GPU: builds shard `rank` of `world`
of the config-2 model and times its core (no exchange, no read-out).  usage: bench_shard.py world [flags...]
(flags: ssn_model_desc.flags, e.g. 0 = whole-block kernel, 128 = one streaming launch per timestep)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import sspslam_amd.frontend as nengo
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
from sspslam_amd.sharding import shard_range

world = int(sys.argv[1]) if len(sys.argv) > 1 else 8
flag_list = [int(f) for f in sys.argv[2:]] or [0, 128]
space = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(space, path, vels, 10000, seed=0)
lo, hi, per = shard_range(pm.pathintegrator.n_oscs, 0, world)
with pm.model:
    p = nengo.Probe(pm.pathintegrator.oscillators.output[3 * lo:3 * hi], synapse=None)
pm.model.probes.remove(p)
t0 = time.time()
model = build(pm.model, n_eval_points=1000, vco_shard=(0, world), probes=[p], prune=True)
print("shard 0/%d: VCOs [%d,%d) built in %.1fs" % (world, lo, hi, time.time() - t0), flush=True)
for flags in flag_list:
    tag = flags
    sim = Simulator(None, model=model, dtype="f32", flags=flags)
    sim.prepare(6000)
    sim.run_steps(1000, collect=False)
    t0 = time.perf_counter(); sim.run_steps(4000, collect=False); el = time.perf_counter() - t0
    c = sim.counters()
    print("flags %4d: %.2f us/step (device %.2f), launches/step %d -> %.1f sim-s/wall-s per rank" %
          (tag, el / 4000 * 1e6, c["last_run_ms"] / 4000 * 1e3, c["launches_per_step"], 4.0 / el), flush=True)
    sim.close()
