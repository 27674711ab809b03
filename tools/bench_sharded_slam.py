"""Neuron-sharded SLAMNetwork, one process per rank (launch with torch.distributed.run): step time per timestep and the
share of the per-timestep exchange.  With one GPU per rank and --dist-backend nccl the exchange is
ssn_exchange_pack -> RCCL all-reduce -> ssn_exchange_unpack on device buffers; with gloo (rehearsal: several ranks
sharing one GPU) it goes through the host.

  python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 tools/bench_sharded_slam.py --dist-backend gloo
"""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import torch.distributed as dist

ap = argparse.ArgumentParser()
ap.add_argument("--ssp-dim", type=int, default=1015)
ap.add_argument("--pi-n-neurons", type=int, default=10000)
ap.add_argument("--mem-n-neurons", type=int, default=10150)
ap.add_argument("--circonv-n-neurons", type=int, default=100)
ap.add_argument("--steps", type=int, default=256)
ap.add_argument("--eval-points", type=int, default=4000)
ap.add_argument("--dist-backend", default="nccl", choices=["nccl", "gloo"])
ap.add_argument("--host-loop", action="store_true", help="round 2's loop: a blocking ssn_run_phase and a blocking exchange per timestep (A/B)")
a = ap.parse_args()
rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
local = int(os.environ.get("LOCAL_RANK", "0")) % max(1, torch.cuda.device_count())
os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
torch.cuda.set_device(local)
if world > 1:
    dist.init_process_group(a.dist_backend, **({"device_id": torch.device("cuda", local)} if a.dist_backend == "nccl" else {}))
from sspslam_amd import harness as H
from sspslam_amd.sharding import ShardedSLAM
space = H.make_ssp_space(2, a.ssp_dim)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(space, path, vels, n_landmarks=10, pi_n_neurons=a.pi_n_neurons, mem_n_neurons=a.mem_n_neurons,
                       circonv_n_neurons=a.circonv_n_neurons, view_rad=0.2)
shared = world > torch.cuda.device_count()
t0 = time.time()
r = None
for turn in range(world if shared else 1):          # ranks sharing a GPU build one after the other (rocSOLVER)
    if not shared or turn == rank:
        r = ShardedSLAM(sm, rank, world, dtype="f32", device=local, n_eval_points=a.eval_points, host_loop=a.host_loop)
    if shared and world > 1:
        dist.barrier()
build_s = time.time() - t0
r.prepare(2 * a.steps + 8)
r.run_steps(32)
if world > 1:
    dist.barrier()
torch.cuda.synchronize()
t0 = time.perf_counter()
r.run_steps(a.steps)
torch.cuda.synchronize()
wall = time.perf_counter() - t0
# the exchange alone: pack -> all-reduce -> unpack, as run_steps issues it
t1 = time.perf_counter()
for _ in range(64):
    r._exchange()
torch.cuda.synchronize()
ex = (time.perf_counter() - t1) / 64
if world > 1:
    w = torch.tensor([wall, ex], dtype=torch.float64)
    w = w.cuda() if a.dist_backend == "nccl" else w
    dist.all_reduce(w, op=dist.ReduceOp.MAX)
    wall, ex = float(w[0]), float(w[1])
if rank == 0:
    n_ex = sum(hi - lo for lo, hi in r.model.exchange)
    print("sharded SLAM x%d (%s, %s): %.1f us per timestep (%.2f sim-s/wall-s), exchange of %d values alone %.1f us; build %.0f s; "
          "launches per timestep %d" % (world, a.dist_backend, "stream-ordered run" if r._stream_ordered() else "host loop", 1e6 * wall / a.steps, a.steps * 1e-3 / wall, n_ex, 1e6 * ex, build_s,
                                        r.sim.counters()["launches_per_step"]), flush=True)
r.close()
if world > 1:
    dist.barrier()
    dist.destroy_process_group()
