"""Where a timestep of the whole-block VCO kernel (k_ens_block) goes, section by section.

Needs the diagnostic build of the library (s_memtime stamps between the sections of the f32 time loop, ssn_block.hpp):

    make -C semantic-spiking-neural-slam-2023_amd/csrc OUT=../libssn_hip_stamps.so BUILD=build_stamps F32_EXTRA=-DSSN_BLOCK_STAMPS
    SSN_HIP_LIB=$PWD/semantic-spiking-neural-slam-2023_amd/libssn_hip_stamps.so python tools/block_stamps.py [ssp_dim n_per_vco steps]

Prints, per kernel variant, the shader cycles one wave spends per timestep in each section (mean over all waves and
timesteps of the launch) and the launch time; the stamps themselves cost ~5 x (s_memtime + s_waitcnt) per timestep, so
the launch time of the stamped build is printed next to the plain build's (SSN_HIP_LIB unset, same process impossible:
run tools/bench_block.py for that figure).
"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from sspslam_amd import _lib
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

d = int(sys.argv[1]) if len(sys.argv) > 1 else 1015
n = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 1000
variants = sys.argv[4].split(";") if len(sys.argv) > 4 else ["512,20,3", "768,14,3"]
lib = _lib.load()
try:
    fn = lib.ssn_debug_block_stamps
except AttributeError:
    sys.exit("this library has no block stamps: build it with F32_EXTRA=-DSSN_BLOCK_STAMPS and point SSN_HIP_LIB at it")
fn.restype, fn.argtypes = C.c_int, [C.POINTER(C.c_uint64), C.c_int]

s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
pm = H.make_pathint_model(s, path, vels, n)
t0 = time.time()
bm = build(pm.model, n_eval_points=min(4000, max(300, n // 2)))
print("build %.1fs, %d neurons" % (time.time() - t0, bm.n_neurons), flush=True)
names = ["input assembly", "neuron groups (encode + LIF + decode)", "wave reduction + LDS publish", "wait at the barrier",
         "totals + filter + hand-off"]
for v in variants:
    os.environ["SSN_BLOCK_VARIANT"] = v
    sim = Simulator(None, model=bm, dtype="f32", block_steps=steps)
    os.environ.pop("SSN_BLOCK_VARIANT", None)
    sim.prepare(3 * steps)
    sim.run_steps(steps, collect=False)
    out = (C.c_uint64 * 8)()
    fn(out, 1)                                   # drop the warm-up launch's stamps
    sim.run_steps(steps, profile=True, collect=False)
    c0 = sim.counters()
    fn(out, 1)
    st = [int(x) for x in out]
    c = sim.counters()
    wave_steps = st[5]                            # sum over waves of the timesteps each ran
    clock = 100.0 * st[6] / st[7] if st[7] else float("nan")
    us = c["dominant_ms_total"] / max(1, c["dominant_launches"]) * 1e3
    tot = sum(st[:5])
    print("variant %s (%d threads, %d waves per SIMD): launch %.1f us for %d timesteps (stamped build), shader clock %.0f MHz, "
          "%.0f cycles per wave-timestep between the stamps" % (v, c["block_threads"], c["block_threads"] // 256, us, steps, clock, tot / wave_steps))
    for i, nm in enumerate(names):
        print("    %-40s %8.1f cycles per wave-timestep  %5.1f %%" % (nm, st[i] / wave_steps, 100.0 * st[i] / tot))
    sys.stdout.flush()
    sim.close()
