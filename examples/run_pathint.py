#!/usr/bin/env python3
"""Path integration on the MI355X backend with the command line of the reference's
``experiments/run_pathint.py`` (same option names, defaults and result file; ``--backend mi355x`` /
``mi355x-f64`` in place of ``cpu`` / ``ocl``).

    python examples/run_pathint.py --ssp-dim 1015 --pi-n-neurons 10000 --T 20 --save

Writes ``<save-dir>/pi_backend_<backend>..._sspdim_<d>_pinneurons_<n>_T_<T>_limit_<limit>_seed_<seed>.npz`` with the
field names ``plot_trials_2d.py`` of the reference reads (ts, path, real_ssp, pi_sim_out, pi_sims, pi_path, pi_error,
elapsed_time, ...).  Only plotting (``--plot``) and the Loihi back ends are not provided.
"""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import sspslam_amd.frontend as nengo          # noqa: E402  (the nengo-shaped object model; fe.install_as_nengo() for `import nengo`)
from sspslam_amd import harness as H           # noqa: E402
from sspslam_amd.simulator import Simulator    # noqa: E402
from sspslam_amd.sspspace import HexagonalSSPSpace, RandomSSPSpace   # noqa: E402


def parse(argv=None):
    p = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
    p.add_argument("--backend", default="mi355x", type=str, help="mi355x (float32) or mi355x-f64 (parity mode)")
    p.add_argument("--path-data", default=None, type=str, help="The path and name to path data.")
    p.add_argument("--data-dt", default=0.001, type=float)
    p.add_argument("--domain-dim", default=2, type=int, help="Dim of path to generate")
    p.add_argument("--limit", default=0.1, type=float)
    p.add_argument("--seed", default=0, type=int)
    p.add_argument("--T", default=20, type=float, help="The total simulation time in seconds.")
    p.add_argument("--pi-n-neurons", default=800, type=int, help="Number of neurons per VCO population in the PI net.")
    p.add_argument("--ssp-dim", default=97, type=int)
    p.add_argument("--n-scales", default=0, type=int)
    p.add_argument("--n-rotates", default=3, type=int)
    p.add_argument("--length-scale", default=0.2, type=float)
    p.add_argument("--save", action="store_true")
    p.add_argument("--use-rand", action="store_true")
    p.add_argument("--neuron-type", default="lif", help="lif, lifrate, relu")
    p.add_argument("--save-dir", default="data")
    p.add_argument("--save-name-extra", default="")
    p.add_argument("--n-eval-points", default=0, type=int,
                   help="decoder-solve evaluation points per ensemble; 0 = nengo's default max(1500, 2 n)")
    return p.parse_args(argv)


def main(argv=None):
    args = parse(argv)
    dt, tau, radius = 0.001, 0.05, 1.0
    if args.path_data is None:
        T, domain_dim = args.T, args.domain_dim
        path, vels = H.make_random_path(T, dt=dt, limit=args.limit, seed=args.seed, domain_dim=domain_dim, radius=radius)
    else:
        path, vels = H.load_path(args.path_data, data_dt=args.data_dt, dt=dt, radius=radius)
        T, domain_dim = path.shape[0] * dt, path.shape[1]
    bounds = radius * np.tile([-1.0, 1.0], (domain_dim, 1))
    if args.use_rand:
        space = RandomSSPSpace(domain_dim, ssp_dim=args.ssp_dim, domain_bounds=bounds, length_scale=args.length_scale,
                               rng=np.random.default_rng(args.seed))
    elif args.n_scales > 0:
        space = HexagonalSSPSpace(domain_dim, n_scales=args.n_scales, n_rotates=args.n_rotates, domain_bounds=bounds,
                                  length_scale=args.length_scale)
    else:
        space = HexagonalSSPSpace(domain_dim, ssp_dim=args.ssp_dim, domain_bounds=bounds, length_scale=args.length_scale)
    neuron_type = {"lif": nengo.LIF, "lifrate": nengo.LIFRate, "relu": nengo.RectifiedLinear}[args.neuron_type]()
    pm = H.make_pathint_model(space, path, vels, args.pi_n_neurons, tau=tau, neuron_type=neuron_type, seed=args.seed, dt=dt)
    dtype = "f64" if args.backend.endswith("f64") else "f32"
    t0 = time.time()
    sim = Simulator(pm.model, dt=dt, dtype=dtype, n_eval_points=args.n_eval_points or None)
    build_time = time.time() - t0
    start, start2 = time.thread_time(), time.time()
    with sim:
        sim.run(T)
        elapsed_thread_time, elapsed_time = time.thread_time() - start, time.time() - start2
        out, ts = sim.data[pm.probe], sim.trange()
    n = out.shape[0]
    est, sims, err = H.pathint_metrics(space, out, pm.real_ssp[:n], path[:n])
    print("d = %d, %d VCOs x %d neurons, T = %.1f s: build %.1f s, run %.2f s (%.1f sim-s/wall-s); similarity to the true SSP "
          "mean %.4f, final position error %.4f" % (space.ssp_dim, (space.ssp_dim + 1) // 2, args.pi_n_neurons, T, build_time,
                                                   elapsed_time, T / elapsed_time, sims[min(200, n - 1):].mean(), err[-1]))
    if args.save:
        os.makedirs(args.save_dir, exist_ok=True)
        name = "pi_backend_%s%s_sspdim_%d_pinneurons_%d_T_%d_limit_%s_seed_%d.npz" % (
            args.backend, args.save_name_extra, space.ssp_dim, args.pi_n_neurons, int(T), args.limit, args.seed)
        H.save_pathint_results(os.path.join(args.save_dir, name), space, ts, path, pm.real_ssp, out, elapsed_time, args=args,
                               elapsed_thread_time=elapsed_thread_time)
        print("saved", os.path.join(args.save_dir, name))
    return out


if __name__ == "__main__":
    main()
