"""One-off sweep: random networks built neuron-sharded for one rank, run through ShardedSLAM (cycles + single timesteps) against the
unsharded oracle.  usage: fuzz_sharded.py first last"""
import sys, os, types
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
import sspslam_amd.frontend as nengo
from helpers import random_network
from sspslam_amd.builder import build
from sspslam_amd.sharding import ShardedSLAM
from oracle import OracleSimulator
lo, hi = int(sys.argv[1]), int(sys.argv[2])
ok = refused = 0; bad = []
for seed in range(lo, hi):
    net, probes = random_network(seed, shardable=True)
    try:
        build(net, neuron_shard=(0, 1), replicate=[])
    except nengo.BuildError:
        refused += 1; continue
    try:
        full = build(net); ref = OracleSimulator(full); steps = 16 * 3 + 7; ref.run_steps(steps)
        sm = types.SimpleNamespace(model=net, probe=probes[0], slam=None)
        r = ShardedSLAM(sm, 0, 1, dtype="f64", replicate=[]); r.prepare(steps); r.run_steps(steps)
        w = 0.0
        for p in probes:
            q = [i for i, mp in enumerate(full.probes) if mp["probe"] is p][0]
            w = max(w, float(np.abs(r.sim.data[p] - ref.probe_data(q)).max()))
        r.close()
        if w < 1e-9: ok += 1
        else: bad.append((seed, w)); print("MISMATCH", seed, w, flush=True)
    except Exception as e:
        bad.append((seed, repr(e)[:200])); print("ERROR", seed, repr(e)[:300], flush=True)
print("done", lo, hi, "ok", ok, "refused", refused, "bad", bad)
