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


@pytest.mark.parametrize("world", [2, 3, 8])          # (8 ranks over 28 VCOs: shards of 4, the last rank has none)
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


SLAM_WORKER = r"""
import os, sys
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as dist
from helpers import OracleBackedSimulator
from sspslam_amd import harness as H
from sspslam_amd.sharding import ShardedSLAM
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
s = H.make_ssp_space(2, 55)
path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons={M}, circonv_n_neurons=30, view_rad=0.6)
r = ShardedSLAM(sm, rank, world, sim_factory=lambda m: OracleBackedSimulator(m))
r.run_steps({steps})
am = sm.slam.assomemory
W, E = r.learned_decoders(am.conn_out), r.learned_encoders(am.memory)
if rank == 0:
    np.savez({out!r}, probe=r.probe_data(), W=W, E=E, exchange=np.array(r.model.exchange))
dist.barrier()
dist.destroy_process_group()
"""


@pytest.mark.parametrize("world,M", [(2, 120), (3, 125)])
def test_neuron_sharded_slam_equals_unsharded(world, M):
    """SLAMNetwork with every neuron population split over the ranks and ONE all-reduce per timestep (the
    feedback loop of reference slam.py:259,306-307 closes each timestep): trajectory, PES-learned decoders and
    Voja-moved encoders equal the unsharded oracle run (float64 sums in another order: 1e-9).  125 memory neurons over 3
    ranks: shares of 42, 42, 41 padded to 42."""
    import subprocess
    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    from oracle import OracleSimulator
    steps = 300
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "slam.npz")
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(SLAM_WORKER.format(root=ROOT, out=out, M=M, steps=steps))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={world}",
                            "--master-addr", "127.0.0.1", "--master-port", str(29620 + world), script],
                           env=env, capture_output=True, text=True, timeout=900)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        got = np.load(out)
        probe, W, E, exchange = got["probe"], got["W"], got["E"], got["exchange"]
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=M, circonv_n_neurons=30, view_rad=0.6)
    model = build(sm.model)
    ref = OracleSimulator(model)
    ref.run_steps(steps)
    am = sm.slam.assomemory
    W_ref = ref.buf[model.params[am.conn_out].learned_buffer]
    E_ref = ref.buf[model.params[am.memory].encoder_buffer]
    assert np.abs(W_ref).max() > 1e-6 and np.abs(E_ref - model.params[am.memory].scaled_encoders).max() > 1e-3
    np.testing.assert_allclose(probe, ref.probe_data(0), atol=1e-9, rtol=0)
    np.testing.assert_allclose(W, W_ref, atol=1e-12, rtol=1e-9)
    np.testing.assert_allclose(E, E_ref, atol=1e-10, rtol=1e-9)
    assert 3 <= len(exchange) <= 12            # a handful of ranges, one all-reduce per timestep


def test_neuron_sharding_refuses_partial_sums_into_neurons():
    """ovc_ens decodes into the product neurons of the first circular convolution within the timestep (reference
    slam.py:262-266, synapse=None): sharded, it would feed them a partial sum - the builder says which ensemble to replicate."""
    import sspslam_amd.frontend as fe
    from sspslam_amd import harness as H
    from sspslam_amd.builder import build
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30, view_rad=0.6)
    with pytest.raises(fe.BuildError, match="replicate"):
        build(sm.model, neuron_shard=(0, 2))
    ms = [build(H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30,
                                  view_rad=0.6).model, neuron_shard=(r, 2), replicate=None) for r in ()]
    assert ms == []
    a = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30, view_rad=0.6)
    b = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30, view_rad=0.6)
    ma = build(a.model, neuron_shard=(0, 2), replicate=[a.slam.ovc_ens])
    mb = build(b.model, neuron_shard=(1, 2), replicate=[b.slam.ovc_ens])
    assert ma.exchange == mb.exchange and ma.sig_size == mb.sig_size            # the ranks agree on what they exchange
    assert {o["phase"] for o in ma.ops} == {0, 1}
    assert all(o["kind"] in ("lowpass", "pes", "voja") for o in ma.ops if o["phase"] == 1)


PLAN_WORKER = r"""
import os, sys, time
sys.path.insert(0, {root!r}); sys.path.insert(0, os.path.join({root!r}, "tests"))
import numpy as np
import torch.distributed as dist
from helpers import OracleBackedSimulator, small_pathint
from sspslam_amd.sharding import ShardedPathIntegration
rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
dist.init_process_group("gloo", rank=rank, world_size=world)
built, closed = [], []

class Timed(OracleBackedSimulator):
    # a flags-aware factory product: plan 128 is made the slower one on rank 1 only - the collective decision must follow the
    # slowest rank, and both ranks must end on the same plan
    def __init__(self, model, flags):
        super().__init__(model)
        self.flags = flags
        built.append(flags)
    def run_steps(self, n, collect=True, profile=False):
        if self.flags == 128 and rank == 1:
            time.sleep(0.3)
        super().run_steps(n, collect=collect, profile=profile)
    def reset(self):
        from oracle import OracleSimulator
        self.o = OracleSimulator(self.model)
        self.n_steps = 0
    def close(self):
        closed.append(self.flags)

pm = small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2)
r = ShardedPathIntegration(pm, rank, world, sim_factory=lambda m, flags=0: Timed(m, flags), block=64)
seconds = r.choose_plan((0, 128), steps=16)
assert set(seconds) == {{0, 128}} and seconds[128] >= 0.3 > seconds[0], seconds          # the all-reduced maximum, on every rank
assert r._flags == 0 and r.sim.flags == 0
# shard simulators built: flags 0, then 128, then 0 again (the winner); the read-out (rank 0) always with flags 0; never two
# candidates resident at once: each was closed before the next was built
shard_built = [f for f in built]
if rank == 0:
    assert shard_built.count(128) == 1 and shard_built.count(0) == 3, shard_built        # shard 0, read-out 0, shard 128, shard 0
else:
    assert shard_built == [0, 128, 0], shard_built
assert closed[:2] == [0, 128], closed
r.run_steps(128)
if rank == 0:
    np.save({out!r}, r.probe_data())
# the documented injection contract sim_factory(model) still works, and refuses plan switches it cannot honour
r2 = ShardedPathIntegration(small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2), rank, world, sim_factory=lambda m: OracleBackedSimulator(m), block=64)
try:
    r2._make_sim(r2.model, 128)
    raise SystemExit("a factory without flags accepted a plan switch")
except Exception as e:
    assert "plan switches" in str(e), e
dist.barrier()
dist.destroy_process_group()
"""


def test_choose_plan_is_collective_and_keeps_one_candidate_resident():
    """ShardedPathIntegration.choose_plan (bench.py's default at N > 1): two gloo ranks on a flags-aware
    oracle-backed factory - both candidates are timed, the slowest rank's time decides, every rank picks the same plan, the
    read-out simulator ignores the shard's plan switches, and the chosen plan steps to the unsharded oracle's trajectory."""
    import subprocess
    from sspslam_amd.builder import build
    from oracle import OracleSimulator
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from helpers import small_pathint
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "probe.npy")
        script = os.path.join(tmp, "worker.py")
        with open(script, "w") as f:
            f.write(PLAN_WORKER.format(root=ROOT, out=out))
        env = dict(os.environ, MASTER_ADDR="127.0.0.1", OMP_NUM_THREADS="2", OPENBLAS_NUM_THREADS="2")
        p = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2",
                            "--master-addr", "127.0.0.1", "--master-port", "29611", script],
                           env=env, capture_output=True, text=True, timeout=600)
        assert p.returncode == 0, p.stdout[-2000:] + p.stderr[-4000:]
        got = np.load(out)
    ref = OracleSimulator(build(small_pathint(ssp_dim=55, n=30, T=10.0, limit=0.2).model))
    ref.run_steps(128)
    np.testing.assert_array_equal(got, ref.probe_data(0))


def test_sharded_slam_runner_issues_whole_cycles_then_single_timesteps():
    """ShardedSLAM.run_steps against a stand-in with the HIP simulator's phase surface: whole cycles of the plan pipelined over
    the exchange first (cycle_steps + 1 segments, the exchange between consecutive ones - never behind the last), the remaining
    timesteps as 0, x, 2, x, ..., 1; `cycles=False` keeps the per-timestep phases; the host loop and the stream-ordered loop
    issue the same sequence."""
    from sspslam_amd.sharding import ShardedSLAM

    class Sim:
        def __init__(self, c):
            self.c, self.calls, self.n_steps, self._prepared_until = c, [], 0, 10 ** 9
        def cycle_steps(self):
            return self.c
        def run_phase(self, p):
            self.calls.append(p)
        def phase_async(self, p, buf, stream):
            self.calls.append(p)
        def phase_sync(self, stream):
            self.calls.append("sync")
        def exchange_host(self, f):
            self.calls.append("x")

    class Runner(ShardedSLAM):
        def __init__(self, c, cycles, stream_ordered):
            self.sim, self.world, self.rank, self.cycles, self.host_loop = Sim(c), 2, 0, cycles, not stream_ordered
            self.n_steps, self._buf, self._stream, self._so = 0, None, None, stream_ordered
        def _agree(self, err):
            assert err is None
        def _stream_ordered(self):
            return False                      # (the stream-ordered branch needs torch streams: its sequence is checked on the GPU)
        def _exchange(self):
            self.sim.calls.append("x")

    r = Runner(4, True, False)
    r.run_steps(11)                           # 2 cycles of 4 + 3 single timesteps
    cyc = [3, "x", 3, "x", 3, "x", 3, "x", 3]
    assert r.sim.calls == cyc + cyc + [0, "x", 2, "x", 2, "x", 1] and r.n_steps == 11
    r = Runner(4, True, False)
    r.run_steps(8)
    assert r.sim.calls == cyc + cyc
    r = Runner(4, False, False)
    r.run_steps(3)
    assert r.sim.calls == [0, "x", 2, "x", 2, "x", 1]
    r = Runner(0, True, False)               # a library without a cycle plan for this model
    r.run_steps(2)
    assert r.sim.calls == [0, "x", 2, "x", 1]


@pytest.mark.parametrize("seed,world", [(1, 2), (3, 3), (5, 2), (8, 3), (14, 2), (21, 3), (36, 2), (44, 3), (11, 2), (15, 3), (29, 2)])
def test_neuron_sharded_random_networks_equal_unsharded_or_are_refused(seed, world):
    """`build(neuron_shard=(rank, world))` on seeded random networks (helpers.random_network without neuron probes): either
    every rank's model builds and - stepped in lockstep with the partial sums of the exchange ranges added between the phases -
    reproduces the unsharded trajectory on every probe, or the build is refused with the reason (a partial sum that would be read
    within the timestep, a replicated and a partial term in one signal): never a wrong answer."""
    import threading
    import sspslam_amd.frontend as nengo
    from helpers import random_network
    from sspslam_amd.builder import build
    from oracle import OracleSimulator
    net, probes = random_network(seed, shardable=True)
    try:
        shards = [build(net, neuron_shard=(r, world), replicate=[]) for r in range(world)]
    except nengo.BuildError as e:
        assert seed in (11, 15, 29) and "neuron sharding" in str(e)
        return
    assert seed not in (11, 15, 29)
    full = build(net)
    ref = OracleSimulator(full)
    ref.run_steps(80)
    sims = [OracleSimulator(m) for m in shards]
    bar, box, errs = threading.Barrier(world), [None] * world, []

    def make_hook(r):
        def hook(sig, ranges):
            box[r] = np.concatenate([sig[lo:hi] for lo, hi in ranges]) if ranges else np.zeros(0)
            bar.wait()
            tot = sum(box[1:], box[0].copy())
            bar.wait()
            off = 0
            for lo, hi in ranges:
                sig[lo:hi] = tot[off:off + hi - lo]
                off += hi - lo
        return hook

    def run(r):
        try:
            sims[r].exchange_hook = make_hook(r)
            sims[r].run_steps(80)
        except Exception as e:               # noqa: BLE001
            errs.append(repr(e))
            bar.abort()

    th = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errs, errs
    for p in probes:
        q = [i for i, mp in enumerate(full.probes) if mp["probe"] is p][0]
        for r in range(world):
            qs = [i for i, mp in enumerate(shards[r].probes) if mp["probe"] is p][0]
            np.testing.assert_allclose(sims[r].probe_data(qs), ref.probe_data(q), atol=1e-9, rtol=0, err_msg=f"seed {seed} rank {r}")
