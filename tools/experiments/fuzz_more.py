"""One-off sweep of the random-network generator (tests/helpers.random_network) over many seeds in f64 against the oracle.
usage: fuzz_more.py first last [big]"""
import sys, os, traceback
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import random_network
from sspslam_amd.builder import build
from sspslam_amd.simulator import Simulator
from oracle import OracleSimulator
lo, hi = int(sys.argv[1]), int(sys.argv[2])
big = len(sys.argv) > 3 and sys.argv[3] == "big"
fused = chains = 0
bad = []
for seed in range(lo, hi):
    try:
        net, probes = random_network(seed, big=big, learned_probes=True)
        model = build(net, n_eval_points=800) if big else build(net)
        ref = OracleSimulator(model); ref.run_steps(120)
        for dtype, kw in (("f64", {}), ("f64", dict(steps_per_graph=1))):
            with Simulator(None, model=model, dtype=dtype, **kw) as sim:
                c = sim.counters(); fused += c["fused_populations"]; chains += c["serial_chains"]
                sim.run_steps(50); sim.run_steps(70)
                for p in probes:
                    q = [i for i, mp in enumerate(model.probes) if mp["probe"] is p][0]
                    d = float(np.abs(sim.data[p] - ref.probe_data(q)).max())
                    if not d < 1e-9:
                        bad.append((seed, q, kw, d)); print("MISMATCH", seed, q, kw, d, flush=True)
    except Exception as e:
        bad.append((seed, repr(e)[:200])); print("ERROR", seed, repr(e)[:300], flush=True)
    if seed % 20 == 0: print("seed", seed, "bad so far", len(bad), flush=True)
print("done", lo, hi, "bad", bad, "fused populations", fused, "serial chains", chains)
