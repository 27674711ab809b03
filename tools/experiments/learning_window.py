"""PES / Voja kernel times at SLAM config 3 in a window with a landmark in view and in one without
(VERDICT r1 item 8).  Per-operator plan (flags 2097152) so that every kernel has its own launch and event pair.
usage: python tools/experiments/learning_window.py > profiles/round2_learning_window.txt"""
import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np
from sspslam_amd import harness as H
from sspslam_amd.modelcache import cached_build as build
from sspslam_amd.simulator import Simulator
dt, view_rad, M, d = 0.001, 0.2, 10150, 1015
s = H.make_ssp_space(2, d)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=10000, mem_n_neurons=M, circonv_n_neurons=100, view_rad=view_rad)
dist = np.linalg.norm(sm.obj_locs[None, :, :] - path[:, None, :], axis=2)
in_view = (dist < view_rad).any(axis=1)
def first_run(mask, length, start=200):
    run = 0
    for t in range(start, len(mask)):
        run = run + 1 if mask[t] else 0
        if run >= length:
            return t - length + 1
    return None
W = 256
t_in, t_out = first_run(in_view, W), first_run(~in_view, W)
print(f"in view on {in_view.mean():.2%} of the path; window with a landmark in view from step {t_in}, without from step {t_out}")
bm = build(sm.model, n_eval_points=4000)
for name, t0 in (("landmark in view", t_in), ("no landmark in view", t_out)):
    if t0 is None:
        continue
    with Simulator(None, model=bm, dtype="f32", flags=2097152) as sim:
        sim.prepare(t0 + W + 8)
        sim.run_steps(t0, collect=False)
        sim.run_steps(W, profile=2, collect=False)
        kt = sim.kernel_times()
        spikes = sim.read_signal(*bm.sig[("ens_spk", sm.slam.assomemory.memory)]) if ("ens_spk", sm.slam.assomemory.memory) in bm.sig else None
    print(f"--- {name}: steps [{t0}, {t0 + W})")
    for k, (n, ms) in sorted(kt.items(), key=lambda kv: -kv[1][1]):
        us = 1e3 * ms / n
        extra = ""
        if k == "k_pes":
            b = 2 * M * 1016 * 4
            extra = f"   {b / 1e6:.1f} MB read + written per launch -> {b / us / 1e3:.0f} GB/s"
        if k == "k_voja" and spikes is not None:
            rows = int((np.asarray(spikes) != 0).sum())
            b = 2 * rows * 1016 * 4
            extra = f"   {rows} spiking rows at the end of the window: {b / 1e6:.2f} MB -> {b / us / 1e3:.0f} GB/s (event time includes the ~6 us launch floor)"
        print(f"  {k:18s} {n // W:2d} launches/step  {us:8.2f} us per launch{extra}")
