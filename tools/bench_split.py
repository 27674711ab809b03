"""Split ensembles of the whole-block kernel on ONE GPU: the per-rank step time of a 4- / 8-GPU shard of BASELINE config 2
(127 / 64 VCOs of 10 000 neurons) with 1, 2 and 4 member workgroups per VCO (flag 1073741824, SSN_BLOCK_SPLIT).
usage: python tools/bench_split.py [world ...]        (default 4 8)"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

SPLIT = 1073741824
space = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
for world in [int(x) for x in sys.argv[1:]] or [4, 8]:
    pm = H.make_pathint_model(space, path, vels, 10000, seed=0)
    from sspslam_amd import frontend as nengo
    pi = pm.pathintegrator
    K = pi.n_oscs
    per = -(-K // world)
    with pm.model:
        probe = nengo.Probe(pi.oscillators.output[0:3 * per], synapse=None)
    pm.model.probes.remove(probe)
    probe.unused = np.all(np.asarray(pi.to_SSP)[:, 0:3 * per] == 0.0, axis=0)        # (as ShardedPathIntegration: dead frequency rows)
    model = build(pm.model, n_eval_points=2000, vco_shard=(0, world), probes=[probe], prune=True)
    base = None
    for P, variant in ((1, None), (2, None), (2, "512,10,0"), (4, None), (4, "512,6,0"), (4, "512,10,0")):
        if per * P > 256:
            continue
        os.environ.pop("SSN_BLOCK_VARIANT", None)
        if variant:
            os.environ["SSN_BLOCK_VARIANT"] = variant
        os.environ["SSN_BLOCK_SPLIT"] = str(P)
        sim = Simulator(None, model=model, dtype="f32", flags=SPLIT if P > 1 else 0, block_steps=1000)
        os.environ.pop("SSN_BLOCK_SPLIT", None)
        os.environ.pop("SSN_BLOCK_VARIANT", None)
        sim.prepare(4000)
        sim.run_steps(1000, collect=False)
        sim.run_steps(2000, profile=True, collect=False)
        c = sim.counters()
        us = c["dominant_ms_total"] / max(1, c["dominant_launches"])
        sim._collect()
        got = sim.data[probe]
        if base is None:
            base = got
        print("shard of %d GPUs (%d VCOs x 10 000): %d member(s) per VCO, variant (%d,%d,%d): %.3f us per timestep (kernel), "
              "max |diff| to the unsplit run %.2e" % (world, per, c["block_members"], c["block_tpb"], c["block_npt"], c["block_enc_lds"], us,
                                                     np.abs(got - base).max()), flush=True)
        sim.close()
