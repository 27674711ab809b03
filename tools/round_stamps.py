"""What every round of a SLAM config-3 timestep waits for: per k_round launch of the steady state, per body kind, the number of
blocks, when the first one started and the last one ended (relative to the launch's first block), and the block-time sum.

Needs the diagnostic build of the library (every block of k_round leaves its launch id, body kind and s_memrealtime stamps):

    make -C semantic-spiking-neural-slam-2023_amd/csrc OUT=../libssn_hip_rstamps.so BUILD=build_rstamps F32_EXTRA=-DSSN_ROUND_STAMPS
    SSN_HIP_LIB=$PWD/semantic-spiking-neural-slam-2023_amd/libssn_hip_rstamps.so python tools/round_stamps.py [first_launch n_launches]

(the stamps cost two s_memrealtime + one atomic per block: the launch times of this build are a few percent above the plain one's)
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from sspslam_amd import _lib
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator

KINDS = ["glue", "gate", "argmax", "matvec_r1", "matvec_r4", "spmv", "neurons", "dft", "pes", "voja", "ens_3_4", "ens_3_5",
         "ens_1_1", "ens_small", "grid_lhs", "grid_dot", "matvec+neurons", "serial chain"]
lib = _lib.load()
try:
    fn = lib.ssn_debug_round_stamps
except AttributeError:
    sys.exit("this library has no round stamps: build it with F32_EXTRA=-DSSN_ROUND_STAMPS and point SSN_HIP_LIB at it")
fn.restype, fn.argtypes = C.c_longlong, [C.c_void_p, C.c_longlong, C.c_int]

s = H.make_ssp_space(2, 1015)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=10150, circonv_n_neurons=100, view_rad=0.2)
bm = build(sm.model, n_eval_points=4000)
G = int(os.environ.get("SSN_SPG", "64"))
sim = Simulator(None, model=bm, dtype="f32", steps_per_graph=G)
sim.prepare(4 * G + 64)
sim.run_steps(2 * G, collect=False)
fn(None, 0, 1)                                   # reset
sim.run_steps(G, collect=False)                  # one graph replay
cap = 1 << 20
buf = np.zeros((cap, 3), dtype=np.uint64)
n = fn(buf.ctypes.data_as(C.c_void_p), cap, 1)
sim.close()
buf = buf[:n]
lid = (buf[:, 0] >> np.uint64(32)).astype(np.int64)
kind = ((buf[:, 0] >> np.uint64(24)) & np.uint64(255)).astype(np.int64)
t0 = buf[:, 1].astype(np.int64)
t1 = buf[:, 2].astype(np.int64)
ids = sorted(set(lid.tolist()), key=lambda i: t0[lid == i].min())
print("%d blocks stamped, %d launches in one replay of %d timesteps; 100 MHz ticks -> us" % (n, len(ids), G))
first = int(sys.argv[1]) if len(sys.argv) > 1 else len(ids) // 2
count = int(sys.argv[2]) if len(sys.argv) > 2 else 12
prev_end = None
tot = 0.0
for pos in range(first, min(len(ids), first + count)):
    m = lid == ids[pos]
    a, b = t0[m].min(), t1[m].max()
    gap = (a - prev_end) / 100.0 if prev_end is not None else 0.0
    prev_end = b
    tot += (b - a) / 100.0 + max(gap, 0.0)
    print("launch %3d (id %d): %5d blocks, %6.2f us from first block start to last block end, %5.2f us after the launch before" %
          (pos, ids[pos], int(m.sum()), (b - a) / 100.0, gap))
    for k in sorted(set(kind[m].tolist())):
        mk = m & (kind == k)
        d = (t1[mk] - t0[mk]) / 100.0
        print("      %-15s %5d blocks: first start +%6.2f, last end +%6.2f us; block time mean %6.2f max %6.2f, sum %8.1f us" %
              (KINDS[k] if k < len(KINDS) else str(k), int(mk.sum()), (t0[mk].min() - a) / 100.0, (t1[mk].max() - a) / 100.0,
               d.mean(), d.max(), d.sum()))
print("sum over the %d launches shown: %.1f us" % (count, tot))
# the whole replay by body kind: workgroup-slot time (blocks x block time) per timestep - what the bodies take of the chip's
# 256 CUs x 7 workgroups = 1792 slots, whether they stream or wait
print("\nwhole replay (%d timesteps), per timestep:" % G)
span = (t1.max() - t0.min()) / 100.0
print("  wall %.1f us per timestep; slot time available %.1f ms per timestep (1792 slots)" % (span / G, 1792 * span / G / 1e3))
rows = []
for k in sorted(set(kind.tolist())):
    mk = kind == k
    d = (t1[mk] - t0[mk]) / 100.0
    rows.append((d.sum() / G / 1e3, KINDS[k] if k < len(KINDS) else str(k), int(mk.sum()) / G, d.mean(), np.median(d), np.percentile(d, 10)))
tot_ms = sum(r[0] for r in rows)
for ms, name, nb, mean, med, p10 in sorted(rows, reverse=True):
    print("  %-15s %7.2f ms slot time (%4.1f %%)  %7.1f blocks  block time mean %6.2f  median %6.2f  10th pct %6.2f us" % (name, ms, 100 * ms / tot_ms, nb, mean, med, p10))
print("  total %.2f ms" % tot_ms)
