#!/usr/bin/env python3
"""SSP-SLAM on the MI355X backend with the command line of the reference's ``experiments/run_slam.py`` (the options
that concern the model; ``--backend mi355x`` / ``mi355x-f64``).

    python examples/run_slam.py --ssp-dim 1015 --pi-n-neurons 10000 --mem-n-neurons 10150 --n-landmarks 10 --T 20 --save

Prints the trajectory accuracy and the recalled landmark map (``run_slam.py:263-268``: activities of the memory
population for each landmark's semantic pointer times the final PES decoders, decoded to positions) and, with
``--save``, writes the result file with the reference's field names.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sspslam_amd.frontend as nengo          # noqa: E402
from sspslam_amd import harness as H           # noqa: E402
from sspslam_amd.simulator import Simulator    # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--backend", default="mi355x", type=str, help="mi355x (float32) or mi355x-f64 (parity mode)")
    p.add_argument("--domain-dim", default=2, type=int)
    p.add_argument("--path-data", default=None, type=str)
    p.add_argument("--data-dt", default=0.001, type=float)
    p.add_argument("--limit", default=0.1, type=float)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--T", default=200, type=float)
    p.add_argument("--n-landmarks", default=50, type=int)
    p.add_argument("--view-rad", default=0.2, type=float)
    p.add_argument("--update-thres", default=0.2, type=float)
    p.add_argument("--shift-rate", default=0.2, type=float)
    p.add_argument("--pi-n-neurons", default=800, type=int)
    p.add_argument("--mem-n-neurons", default=970, type=int)
    p.add_argument("--circonv-n-neurons", default=100, type=int)
    p.add_argument("--ssp-dim", default=97, type=int)
    p.add_argument("--n-scales", default=0, type=int)
    p.add_argument("--n-rotates", default=3, type=int)
    p.add_argument("--length-scale", default=0.2, type=float)
    p.add_argument("--save", action="store_true")
    p.add_argument("--save-dir", default="data")
    p.add_argument("--save-name-extra", default="")
    p.add_argument("--n-eval-points", default=0, type=int, help="decoder-solve evaluation points; 0 = nengo's default")
    return p.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    dt = 0.001
    if args.path_data is None:
        T = args.T
        path, vels = H.make_random_path(T, dt=dt, limit=args.limit, seed=args.seed, domain_dim=args.domain_dim)
    else:
        path, vels = H.load_path(args.path_data, data_dt=args.data_dt, dt=dt)
        T = path.shape[0] * dt
    space = H.make_ssp_space(path.shape[1], ssp_dim=args.ssp_dim, n_scales=args.n_scales, n_rotates=args.n_rotates,
                             length_scale=args.length_scale)
    sm = H.make_slam_model(space, path, vels, n_landmarks=args.n_landmarks, pi_n_neurons=args.pi_n_neurons,
                           mem_n_neurons=args.mem_n_neurons, circonv_n_neurons=args.circonv_n_neurons, view_rad=args.view_rad,
                           update_thres=args.update_thres, shift_rate=args.shift_rate, seed=args.seed, dt=dt,
                           weights_sample_every=T)
    dtype = "f64" if args.backend.endswith("f64") else "f32"
    t0 = time.time()
    sim = Simulator(sm.model, dt=dt, dtype=dtype, n_eval_points=args.n_eval_points or None)
    build_time = time.time() - t0
    start = time.time()
    with sim:
        sim.run(T)
        elapsed = time.time() - start
        out, ts = sim.data[sm.probe], sim.trange()
        decoders = sim.data[sm.weights_probe][-1]
        built_memory = sim.data[sm.slam.assomemory.memory]
    n = out.shape[0]
    est, sims, err = H.pathint_metrics(space, out, sm.real_ssp[:n], path[:n])
    lm_ssps, lm_locs = H.map_recall(space, sm.lm_space, built_memory, nengo.LIF(), decoders)
    lm_err = np.linalg.norm(lm_locs - sm.obj_locs, axis=1)
    print("d = %d, T = %.1f s: build %.1f s, run %.2f s (%.2f sim-s/wall-s); similarity to the true SSP mean %.4f, final position "
          "error %.4f; recalled landmark positions: median error %.3f" % (space.ssp_dim, T, build_time, elapsed, T / elapsed,
                                                                         sims[min(200, n - 1):].mean(), err[-1], float(np.median(lm_err))))
    if args.save:
        os.makedirs(args.save_dir, exist_ok=True)
        name = "slam_backend_%s%s_sspdim_%d_pinneurons_%d_memnneurons_%d_ccnneurons_%d_T_%d_limit_%s_seed_%d.npz" % (
            args.backend, args.save_name_extra, space.ssp_dim, args.pi_n_neurons, args.mem_n_neurons, args.circonv_n_neurons,
            int(T), args.limit, args.seed)
        H.save_slam_results(os.path.join(args.save_dir, name), space, ts, path, sm.real_ssp, sm.obj_locs, args.view_rad, out,
                            lm_ssps, lm_locs, elapsed, args=args)
        print("saved", os.path.join(args.save_dir, name))
    return out, lm_locs


if __name__ == "__main__":
    main()
