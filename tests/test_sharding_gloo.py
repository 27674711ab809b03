"""N > 1 path on CPU: two processes (gloo) each step their VCO shard on the oracle, all-gather per block,
rank 0 replays the read-out - the result must equal the unsharded model bit for bit."""
import os
import sys
import tempfile

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as dist
from helpers import OracleBackedSimulator, small_pathint
from sspslam_amd.sharding import ShardedPathIntegration
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
pm = small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2)
r = ShardedPathIntegration(pm, rank, world, sim_factory=lambda m: OracleBackedSimulator(m), block=70)
assert (r.lo, r.hi) == ((0, 10), (10, 20), (20, 28))[rank] if world == 3 else True
r.run_steps(200)          # 70 + 70 + 60: ragged last block
if rank == 0:
    np.save({out!r}, r.probe_data())
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_equals_unsharded(world):
    import subprocess
    from sspslam_amd.builder import build
    from oracle import OracleSimulator
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import small_pathint
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "probe.npy")
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(WORKER.format(root=ROOT, out=out))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT=str(29650 + world), OMP_NUM_THREADS="2",
                   OPENBLAS_NUM_THREADS="2")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                            "--master-addr", "127.0.0.1", "--master-port", str(29600 + world), script],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        got = np.load(out)
    ref = OracleSimulator(build(small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2).model))
    ref.run_steps(200)
    want = ref.probe_data(0)
    assert got.shape == want.shape == (200, 55)
    np.testing.assert_array_equal(got, want)


def test_shard_ranges_cover_all_vcos():
    from sspslam_amd.sharding import shard_range
    for K in (28, 508, 2017, 3):
        for world in (1, 2, 4, 8):
            spans = [shard_range(K, r, world)[:2] for r in range(world)]
            assert spans[0][0] == 0 and spans[-1][1] == K
            assert all(a[1] == b[0] for a, b in zip(spans, spans[1:]))
            assert all(0 <= hi - lo <= shard_range(K, 0, world)[2] for lo, hi in spans)


def test_assemble_gathered_with_a_padded_last_shard():
    """The device exchange pads every rank's block to the common shard width; 28 VCOs over 3 ranks are shards of
    10, 10 and 8, and the padding of the last one must fall off the end of the assembled block."""
    import torch
    from sspslam_amd.sharding import ShardedPathIntegration, shard_range
    K, world, n = 28, 3, 5
    full = torch.arange(n * 3 * K, dtype=torch.float64).reshape(n, 3 * K)
    per = shard_range(K, 0, world)[2]
    out = torch.zeros((world, n, 3 * per), dtype=torch.float64)
    for r in range(world):
        lo, hi, _ = shard_range(K, r, world)
        out[r, :, :3 * (hi - lo)] = full[:, 3 * lo:3 * hi]
    got = ShardedPathIntegration.assemble_gathered(out, K)
    assert got.shape == (n, 3 * K) and torch.equal(got, full)
