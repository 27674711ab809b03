"""Where the wall time of a whole `sim.run(T)` + `sim.data[probe]` goes at config 2 (PCIe-inclusive rate):
input tabulation + upload (prepare), the device run, probe read-back.  usage: bench_end_to_end.py [T]"""
import os, sys, time
os.environ["SSN_TRACE_RUN"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

T = float(sys.argv[1]) if len(sys.argv) > 1 else 10.0
s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(5 * max(T, 10.0) + 3.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, 10000)
bm = build(pm.model, n_eval_points=4000)
sim = Simulator(None, model=bm, dtype="f32")
steps = int(T / 0.001)
sim.prepare(1024); sim.run_steps(1024, collect=False); sim._collect()          # warm
for rep in range(2):
    t0 = time.perf_counter(); sim.prepare(steps); t1 = time.perf_counter()
    sim.run_steps(steps, collect=False); t2 = time.perf_counter()
    sim._collect(); out = sim.data[pm.probe]; t3 = time.perf_counter()
    print("T = %.0f s: prepare (tabulate + upload) %.1f ms, run %.1f ms (%.1f sim-s/wall-s), probe read-back (%d x %d -> float64 host) %.1f ms; "
          "end to end %.1f sim-s/wall-s" % (T, (t1 - t0) * 1e3, (t2 - t1) * 1e3, T / (t2 - t1), steps, out.shape[1], (t3 - t2) * 1e3, T / (t3 - t0)), flush=True)
    sim.clear_probe_data()
t0 = time.perf_counter()
sim.run_steps(steps)                       # unprepared: tabulation of chunk k+1 overlaps the device run of chunk k
out = sim.data[pm.probe]
t1 = time.perf_counter()
print("T = %.0f s: plain run_steps + data (pipelined tabulation): %.1f ms -> %.1f sim-s/wall-s end to end" % (T, (t1 - t0) * 1e3, T / (t1 - t0)), flush=True)

# the same with a per-phase timeline of the pipelined path
import threading
sim.clear_probe_data()
orig_run, orig_coll = sim._lib.ssn_run_steps, sim._collect_bulk
marks = []
def timed_collect():
    a = time.perf_counter(); orig_coll(); marks.append(("collect", a, time.perf_counter()))
sim._collect_bulk = timed_collect
t0 = time.perf_counter()
sim.run_steps(steps)
t1 = time.perf_counter()
out = sim.data[pm.probe]
t2 = time.perf_counter()
print("pipelined run_steps %.1f ms + data[probe] %.1f ms; helper-thread fetches: %s" %
      ((t1 - t0) * 1e3, (t2 - t1) * 1e3, ", ".join("%.1f-%.1f ms" % ((a - t0) * 1e3, (b - t0) * 1e3) for _, a, b in marks)), flush=True)


# the library-side timeline of pipelined runs (Simulator.trace), with the harness's vectorised input nodes and with plain closures
for label, strip in (("harness nodes (.table twins)", False), ("plain per-timestep closures", True)):
    sim.reset()
    sim.clear_probe_data()
    saved = [tb["fn"] for tb in bm.tables]
    if strip:
        for tb in bm.tables:
            tb["fn"] = (lambda f: (lambda t: f(t)))(tb["fn"])
    for rep in range(2):
        sim.reset(); sim.clear_probe_data()
        sim.trace.clear()
        t0 = time.perf_counter()
        sim.run(T)
        out = sim.data[pm.probe]
        t1 = time.perf_counter()
    for tb, f in zip(bm.tables, saved):
        tb["fn"] = f
    print("%s: %.1f ms end to end -> %.1f sim-s/wall-s" % (label, (t1 - t0) * 1e3, T / (t1 - t0)))
    for name, a, b, detail in sim.trace:
        print("   %7.2f .. %7.2f ms  (%6.2f)  %s%s" % ((a - t0) * 1e3, (b - t0) * 1e3, (b - a) * 1e3, name, "" if detail is None else " [%s]" % detail))
