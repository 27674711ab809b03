"""Whole-block kernel at BASELINE configs[1]: neurons in sampling order vs dealt to (wave, round) slots by firing sector
(SSN_BLOCK_SORT, Sim::reorder_block_neurons) - kernel time per 1000-timestep block, share of silent slots, parity against the
f64 mode over one simulated second.  Run once per library build to compare kernels (SSN_HIP_LIB=.../libssn_hip_noskip.so:
built with F32_EXTRA=-DSSN_BLOCK_SKIP=0, the round-3 time loop).
usage: bench_block_sort.py [blocks] [n_per_vco] [ssp_dim]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

blocks = int(sys.argv[1]) if len(sys.argv) > 1 else 12
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
d = int(sys.argv[3]) if len(sys.argv) > 3 else 1015
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
bm = build(pm.model, n_eval_points=4000)
print("lib", os.environ.get("SSN_HIP_LIB", "default"), "build", bm.stats.get("cache"), flush=True)
ref = None
with Simulator(None, model=bm, dtype="f64") as sim:
    sim.run_steps(1000)
    ref = np.array(sim.data[pm.probe])
for sort in ("0", "1", "0", "1"):
    os.environ["SSN_BLOCK_SORT"] = sort
    with Simulator(None, model=bm, dtype="f32", block_steps=1000) as sim:
        sim.prepare((blocks + 3) * 1000)
        sim.run_steps(1000)
        got = np.array(sim.data[pm.probe])
        sim.run_steps(2000, collect=False)
        c0 = sim.counters()
        t0 = time.perf_counter()
        sim.run_steps(blocks * 1000, profile=True, collect=False)
        wall = time.perf_counter() - t0
        c = sim.counters()
    ms = (c["dominant_ms_total"] - c0["dominant_ms_total"]) / max(1, c["dominant_launches"] - c0["dominant_launches"])
    slots, silent = c["block_slots"] - c0["block_slots"], c["block_slots_silent"] - c0["block_slots_silent"]
    ce = H.cosine_error(got[20:], ref[20:])
    print("sort %s: k_ens_block %.3f ms per 1000 timesteps (%d launches), %.1f sim-s/wall-s; silent slots %.1f %% of %d; "
          "f32 vs f64 over 1 s: max cosine error %.2e" % (sort, ms, c["dominant_launches"] - c0["dominant_launches"],
                                                        blocks / wall, 100.0 * silent / max(1, slots), slots, ce.max()), flush=True)
