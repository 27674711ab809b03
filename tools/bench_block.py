"""Whole-block VCO kernel (k_ens_block) variants vs the per-timestep kernel, one process (A/B on one box).

usage: bench_block.py ssp_dim n_per_vco steps n_eval variant[;variant...]
  variant = "tpb,npt,lds" (SSN_BLOCK_VARIANT), "auto" (library's choice) or "step" (flags=128: k_ensarray per timestep)
"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

d, n, steps, m_eval = [int(a) for a in sys.argv[1:5]]
variants = sys.argv[5].split(";") if len(sys.argv) > 5 else ["auto", "step"]
block = int(os.environ.get("BLOCK", "256"))
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
t0 = time.time()
bm = build(pm.model, n_eval_points=m_eval or None)
print("build %.1fs" % (time.time() - t0), bm.stats, flush=True)
base = None
for v in variants:
    os.environ.pop("SSN_BLOCK_VARIANT", None)
    flags = 0
    if v == "step":
        flags = 128
    elif v != "auto":
        os.environ["SSN_BLOCK_VARIANT"] = v
    try:
        sim = Simulator(None, model=bm, dtype="f32", flags=flags, block_steps=block)
    except Exception as e:
        print("%-12s: %s" % (v, e), flush=True)
        continue
    sim.prepare(3 * steps)
    sim.run_steps(steps, collect=False)
    t0 = time.time()
    sim.run_steps(steps, collect=False)
    el = time.time() - t0
    dev_ms = sim.counters()["last_run_ms"]
    sim.run_steps(steps, profile=True, collect=False)
    c = sim.counters()
    sim._collect()
    got = sim.data[pm.probe]
    if base is None:
        base = got
    k = min(200, steps)
    ce = H.cosine_error(got[20:k], base[20:k]).max()
    dom = c["dominant_ms_total"] / c["dominant_launches"] * 1e3 if c["dominant_launches"] else float("nan")
    print("%-12s: %.2f us/step wall (%.1f sim-s/wall-s), device %.2f us/step, dominant kernel %.1f us x %d, launches/step %d, "
          "cos err vs first (200 steps) %.2e" % (v, el / steps * 1e6, steps * 1e-3 / el, dev_ms / steps * 1e3, dom,
                                                 c["dominant_launches"], c["launches_per_step"], ce), flush=True)
    sim.close()
