"""Multi-GPU execution of the path integrator: VCO ensembles sharded over ranks.

The VCO ensembles of ``PathIntegration`` are mutually independent (each one's recurrence touches only
itself, reference ``networks/pathintegration.py:173-185``); the only cross-VCO coupling is the linear
read-out ``to_SSP @ oscillators.output`` (``:191``).  So, one process per GPU:

* rank r builds and steps only VCOs ``[r*ceil(K/W), (r+1)*ceil(K/W))`` - same seeds as the unsharded
  model, so the union of the shards IS that model; everything the shard does not need (the read-out
  chain) is pruned from its operator list;
* every ``block`` timesteps the ranks all-gather their decoded oscillator outputs
  (``block x 3*K/W`` values per rank; RCCL over xGMI under torch.distributed's "nccl" backend, gloo
  on CPU) - exact, because nothing downstream feeds back into the oscillators within a
  PathIntegration-only model - and rank 0 replays the gathered block through the read-out
  (``to_SSP`` -> Lowpass(5 ms) -> output node -> probe filter), a second small simulator whose input
  node is tabulated from the gathered data.

The simulator factory is injectable so the orchestration is exercised on CPU (gloo, world_size 2)
against the oracle in tests; by default it is the HIP ``Simulator``.
"""
import queue
import threading

import numpy as np

from . import frontend as nengo
from .modelcache import cached_build as build      # (a rank's shard is keyed by its (rank, world) like any other build argument)


def shard_range(K, rank, world):
    per = -(-K // world)
    return min(K, rank * per), min(K, (rank + 1) * per), per


class _BlockFeed:
    """Node function of the read-out's input: serves rows of the last gathered block."""

    def __init__(self, width, dt):
        self.width, self.dt = width, dt
        self.rows = np.zeros((0, width))
        self.first = 0          # 0-based step number of rows[0]

    def __call__(self, t):
        j = int(round(t / self.dt)) - 1 - self.first
        return self.rows[j] if 0 <= j < self.rows.shape[0] else np.zeros(self.width)

    def table(self, steps):
        j = np.asarray(steps) - 1 - self.first
        ok = (j >= 0) & (j < self.rows.shape[0])
        return self.rows, np.where(ok, j, -1).astype(np.int32)


class ShardedPathIntegration:
    def __init__(self, pm, rank, world, dt=0.001, dtype="f32", device=0, n_eval_points=None, block=1000,
                 sim_factory=None, dist=None, gather_device=None, async_readout=True, block_steps=0,
                 device_exchange=None, defer_readout=None, gather_every=4, flags=0):
        """``pm``: object from ``harness.make_pathint_model`` (model, pathintegrator, probe).  ``flags``: plan switches
        of the shard's simulator (``ssn_model_desc.flags``; see ``choose_plan``)."""
        if dist is None:
            import torch.distributed as dist
        self.dist, self.rank, self.world, self.dt, self.block = dist, rank, world, dt, int(block)
        pi = pm.pathintegrator
        K = pi.n_oscs
        self.K = K
        self.lo, self.hi, self.per = shard_range(K, rank, world)
        if sim_factory is None:
            from .simulator import Simulator

            def sim_factory(model, flags=0):
                return Simulator(None, model=model, dtype=dtype, device=device, block_steps=block_steps, flags=flags)
        self._sim_factory, self._flags = sim_factory, flags
        self.gather_device = gather_device
        self.dtype = dtype
        self.device_exchange = device_exchange     # None: automatic (RCCL backend + HIP simulator); True/False: forced
        # Device path: rank 0's GPU is (nearly) full while its VCO shard is being stepped - at two ranks 254 of the
        # 256 CUs hold a k_ens_block workgroup for the whole block - so read-out kernels launched meanwhile would
        # crawl on the few free CUs.  Gathered blocks (6 MB each at d = 1015) are therefore kept in HBM and replayed
        # through the read-out `defer_readout` blocks at a time, or when flush() / probe_data() asks for them.
        # Default: defer (16 blocks at a time) when this rank's VCOs occupy most of the GPU (> 192 of 256 CUs),
        # otherwise hand each block to the worker thread at once - the read-out then overlaps the next block
        # on the idle CUs.
        if defer_readout is None:
            defer_readout = 16 if (self.hi - self.lo) > 192 else 0
        self.defer_readout = int(defer_readout)
        self._pending = []
        # Device path: the oscillator samples stay in the simulator's probe storage, so the all-gather (and its
        # fixed host + launch cost, ~0.3 ms against a 2.7 ms block) is issued once per `gather_every` blocks.
        # flush() gathers the remainder and is therefore a COLLECTIVE on this path: call it on every rank.
        self.gather_every = max(1, int(gather_every))
        self._ungathered = 0
        # --- this rank's VCO shard: probe = local slice of the oscillator output node ----------------
        with pm.model:
            width = 3 * (self.hi - self.lo)
            self.osc_probe = nengo.Probe(pi.oscillators.output[3 * self.lo:3 * self.hi], synapse=None) \
                if width else None
        if self.osc_probe is not None:
            pm.model.probes.remove(self.osc_probe)
            # oscillator outputs that meet all-zero columns of the read-out (the frequency dimension of every VCO,
            # get_from_Fourier, reference pathintegration.py:824-844) need not be decoded by the shard either: the unsharded
            # model drops those rows through the builder's liveness pass, the shard's probe would otherwise keep them alive
            # (a fifth decoded row per VCO: ~6 % of the shard's step time, and no split ensembles)
            self.osc_probe.unused = np.all(np.asarray(pi.to_SSP)[:, 3 * self.lo:3 * self.hi] == 0.0, axis=0)
        probes = [self.osc_probe] if self.osc_probe is not None else []
        self.model = build(pm.model, dt=dt, n_eval_points=n_eval_points, vco_shard=(rank, world),
                           probes=probes, prune=True)
        self.sim = self._make_sim(self.model, flags)
        # --- rank 0: the read-out replayed from gathered blocks ---------------------------------------
        self.readout = None
        if rank == 0:
            d = pi.to_SSP.shape[0]
            self.feed = _BlockFeed(3 * K, dt)
            psyn = pm.probe.synapse.tau if pm.probe.synapse is not None else None
            with nengo.Network(seed=0) as ro:
                src = nengo.Node(self.feed, size_out=3 * K, label="gathered_osc_output")
                out = nengo.Node(size_in=d, label="pathint_output")
                nengo.Connection(src, out, transform=pi.to_SSP)       # default synapse, as pathintegration.py:191
                self.ro_probe = nengo.Probe(out, synapse=psyn)
            self.readout_model = build(ro, dt=dt)
            self.readout = self._make_sim(self.readout_model, 0)      # (the shard's plan switches do not concern the read-out)
        self.n_steps = 0
        # rank 0 replays block b through the read-out on a worker thread while every rank already steps
        # block b+1 (the library releases the GIL; the two simulators have their own HIP streams)
        self._jobs = self._worker = self._error = None
        if self.readout is not None and async_readout:
            self._jobs = queue.Queue(maxsize=2)
            self._worker = threading.Thread(target=self._readout_loop, daemon=True)
            self._worker.start()
        self._warm = False

    def _make_sim(self, model, flags):
        """``sim_factory(model)`` is the documented injection contract (the oracle-backed test factories follow it); a factory
        that also understands plan switches takes them as ``flags=``."""
        import inspect
        try:
            params = inspect.signature(self._sim_factory).parameters
            takes_flags = "flags" in params or any(p.kind == p.VAR_KEYWORD for p in params.values())
        except (TypeError, ValueError):
            takes_flags = False
        if takes_flags:
            return self._sim_factory(model, flags=flags)
        if flags:
            raise nengo.BuildError("this simulator factory takes no plan switches (flags=%d asked for)" % flags)
        return self._sim_factory(model)

    def choose_plan(self, candidates=(0, 128), steps=None):
        """Time one block of this rank's shard under each candidate plan (simulator flags: 0 = the planner's choice -
        the whole-block kernel k_ens_block where a VCO fits one workgroup; 128 = one k_ensarray launch per timestep,
        which splits an ensemble over several workgroups and so fills the chip when a shard has fewer VCOs than the GPU
        has CUs) and keep the fastest.  Decided collectively - the slowest rank's time counts - so every rank must call
        it, before ``prepare``.  Returns {flags: seconds per block}."""
        import time
        steps = int(steps or self.block)
        seconds = {}
        order = [self._flags] + [fl for fl in candidates if fl != self._flags]      # the simulator already built goes first
        sim = self.sim
        for n_done, fl in enumerate(order):
            if fl not in candidates:
                continue
            if n_done > 0:
                # one candidate resident at a time: the previous one is closed before the next is built (a config-4 shard
                # holds gigabytes of parameters)
                if hasattr(sim, "close"):
                    sim.close()
                sim = self._make_sim(self.model, fl)
                self.sim, self._flags = sim, fl
            sim.prepare(2 * steps)
            sim.run_steps(steps, collect=False)
            t0 = time.perf_counter()
            sim.run_steps(steps, collect=False)
            wall = time.perf_counter() - t0
            if self._grouped():
                import torch
                t = torch.tensor([wall], dtype=torch.float64)
                if self.dist.get_backend() == "nccl":
                    t = t.cuda()
                self.dist.all_reduce(t, op=self.dist.ReduceOp.MAX)
                wall = float(t.item())
            seconds[fl] = wall
            if hasattr(sim, "reset"):
                sim.reset()
            if hasattr(sim, "clear_probe_data"):
                sim.clear_probe_data()
        best = min(seconds, key=lambda fl: (seconds[fl], fl))       # (the times were all-reduced: every rank picks the same plan)
        if best != self._flags:
            if hasattr(sim, "close"):
                sim.close()
            self.sim, self._flags = self._make_sim(self.model, best), best
        return seconds

    def prepare(self, n_steps):
        self.sim.prepare(n_steps)
        if self.readout is not None and hasattr(self.readout, "reserve_probes"):
            self.readout.reserve_probes(n_steps)
        if self.world > 1 and not self._warm and self.dist.is_initialized():
            self._gather(np.zeros((1, 3 * (self.hi - self.lo))))   # the first collective sets up the communicator: untimed
            self._warm = True

    def _replay(self, full, first, n):
        if not isinstance(full, np.ndarray):
            # device path: the gathered block is already in HBM in the simulator's dtype - it becomes the
            # read-out's input table without a host round trip
            self.readout.prepare_tables_device({0: (full.data_ptr(), n, np.arange(n, dtype=np.int32))}, n, reserve=False)
            self.readout.run_steps(n, collect=False)      # samples stay in HBM until probe_data() asks for them
            return
        self.feed.rows, self.feed.first = np.ascontiguousarray(full), first
        self.readout.prepare(n)
        self.readout.run_steps(n)

    def _grouped(self):
        """True when this runner's collectives are real: a process group of the runner's world size (also of size one -
        `bench.py --rehearse-dist` takes one rank through the RCCL calls on a one-GPU box)."""
        if not self.dist.is_initialized():
            return False
        return self.world > 1 or self.dist.get_world_size() == 1

    def _device_exchange(self):
        """True when the block exchange can stay in HBM: RCCL backend and the HIP simulator (not the test double)."""
        if not hasattr(self.sim, "read_probe_device") or self.gather_device is not None or self.device_exchange is False:
            return False
        if self.device_exchange:
            return True
        return self._grouped() and self.dist.get_backend() == "nccl"

    def _stage_send(self, n):
        """Local half of the device exchange (no collective): this rank's last n oscillator samples, padded to the
        common shard width, as an (n, 3*per) device tensor."""
        import torch
        tdt = torch.float32 if self.dtype == "f32" else torch.float64
        dev = torch.device("cuda", torch.cuda.current_device())
        width = 3 * (self.hi - self.lo)
        send = torch.zeros((n, 3 * self.per), dtype=tdt, device=dev)
        torch.cuda.current_stream().synchronize()      # the simulator copies on its own stream: the zero fill must have landed
        if width:
            have = self.sim.probe_count(self.osc_probe)
            if have < n:
                raise nengo.SimulationError(f"rank {self.rank}: {have} oscillator samples on the device, {n} to exchange - "
                                            "call prepare(total_steps) before run_block on the device-exchange path")
            if width == 3 * self.per:
                self.sim.read_probe_device(self.osc_probe, send.data_ptr(), have - n, n)
            else:
                tmp = torch.empty((n, width), dtype=tdt, device=dev)
                self.sim.read_probe_device(self.osc_probe, tmp.data_ptr(), have - n, n)
                send[:, :width] = tmp
        return send

    @staticmethod
    def assemble_gathered(out, K):
        """(world, n, 3*per) all-gather result -> (n, 3K): rank r's columns follow rank r-1's; the zero padding of the
        last shard (K not a multiple of the world size) falls off the end."""
        world, n, w = out.shape
        return out.permute(1, 0, 2).reshape(n, world * w)[:, :3 * K].contiguous()

    def _gather_device(self, n):
        """This rank's last n osc samples (device) -> (n, 3K) device tensor on every rank, one all-gather.

        A failure of the local half must not leave this rank outside a collective its peers have entered (they would
        wait for the RCCL watchdog): every rank first contributes an ok flag to a one-element all-reduce, and either
        all ranks go on to the all-gather or all ranks raise."""
        import torch
        send, err = None, None
        try:
            send = self._stage_send(n)
        except Exception as e:                         # noqa: BLE001 - agreed on collectively below, then re-raised
            err = e
        collective = self._grouped()
        failed = err is not None
        if collective:
            flag = torch.tensor([1.0 if failed else 0.0], device=torch.device("cuda", torch.cuda.current_device()))
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            failed = bool(flag.item() > 0)
        if failed:
            raise nengo.SimulationError(f"device-resident block exchange failed on rank {self.rank}" if err is not None else
                                        "device-resident block exchange failed on another rank") from err
        out = torch.empty((self.world, n, 3 * self.per), dtype=send.dtype, device=send.device)
        if collective:
            self.dist.all_gather_into_tensor(out, send)
        else:                                          # no process group (tests): only this rank's slot is filled
            out.zero_()
            out[self.rank] = send
        full = self.assemble_gathered(out, self.K)
        torch.cuda.current_stream().synchronize()      # the read-out simulator runs on its own HIP stream
        return full

    def _readout_loop(self):
        while True:
            job = self._jobs.get()
            if job is None:
                self._jobs.task_done()
                return
            try:
                self._replay(*job)
            except BaseException as e:       # surfaced by flush()
                self._error = e
            finally:
                self._jobs.task_done()

    def _drain_pending(self):
        pending, self._pending = self._pending, []
        for job in pending:
            self._replay(*job)

    def _drain_jobs(self):
        if self._jobs is not None:
            self._jobs.join()
            if self._error is not None:
                err, self._error = self._error, None
                raise err

    def flush(self):
        """Wait until the read-out has consumed every block handed to it.  On the device path this first gathers the
        timesteps not exchanged yet - a collective: every rank must call it."""
        if self._ungathered:
            self._exchange_pending()
        self._drain_pending()
        if self._jobs is not None:
            self._jobs.join()
            if self._error is not None:
                err, self._error = self._error, None
                raise err

    def _gather(self, local):
        """local (B, 3*(hi-lo)) float64 -> (B, 3K) on every rank."""
        import torch
        B = local.shape[0]
        send = np.zeros((B, 3 * self.per))
        send[:, :local.shape[1]] = local
        dev = self.gather_device
        if dev is None:
            dev = torch.device("cuda", torch.cuda.current_device()) if self.dist.get_backend() == "nccl" else torch.device("cpu")
        t = torch.from_numpy(send).to(dev)
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        full = torch.cat(outs, dim=1).cpu().numpy()
        return full[:, :3 * self.K]

    def run_block(self, n=None):
        n = self.block if n is None else int(n)
        if self._device_exchange():
            if getattr(self.sim, "_prepared_until", 0) < self.sim.n_steps + n:
                # (an unprepared run_steps would re-reserve the probe storage block by block and drop the samples
                #  that wait on the device for the next exchange)
                raise nengo.SimulationError("device-exchange path: call prepare(total_steps) before run_block")
            self.sim.run_steps(n, collect=False)
            self._ungathered += n
            self.n_steps += n
            if self._ungathered >= self.gather_every * self.block:
                self._exchange_pending()
            return
        self.sim.run_steps(n, collect=True)
        if self.osc_probe is not None:
            local = self.sim.probe_tail(self.osc_probe, n)
            self.sim.clear_probe_data()
        else:
            local = np.zeros((n, 0))
        full = self._gather(local) if self.world > 1 else local
        if self.readout is not None:
            if self._jobs is not None:
                if self._error is not None:
                    self.flush()
                self._jobs.put((full, self.n_steps, n))
            else:
                self._replay(full, self.n_steps, n)
        self.n_steps += n

    def _exchange_pending(self):
        """Device path: gather every timestep stepped since the last exchange and hand it to the read-out."""
        n = self._ungathered
        if n == 0:
            return
        self._ungathered = 0
        first = self.n_steps - n
        full = self._gather_device(n)          # raises on every rank alike (see there): no silent change of path mid-run
        if self.readout is None:
            return
        if self.defer_readout > 0:
            self._pending.append((full, first, n))
            if len(self._pending) * self.gather_every >= self.defer_readout:
                self._drain_pending()
        elif self._jobs is not None:
            if self._error is not None:
                self._drain_jobs()
            self._jobs.put((full, first, n))
        else:
            self._replay(full, first, n)

    def run_steps(self, n):
        done = 0
        while done < n:
            c = min(self.block, n - done)
            if self._device_exchange():
                # device path: everything up to the next exchange in ONE call of the simulator - it cuts the run into its blocks
                # itself, and a call per block would put a host synchronisation and a cold launch queue behind every block
                # (bench.py, run_blocks: 3.09 vs 3.03 ms per block)
                c = min(n - done, max(1, self.gather_every * self.block - self._ungathered))
            self.run_block(c)
            done += c

    def probe_data(self):
        """Rank 0: the filtered PathIntegration output, as ``sim.data[probe]`` of the unsharded model."""
        self.flush()
        return self.readout.data[self.ro_probe] if self.readout is not None else None

    def close(self):
        if self._jobs is not None:
            self.flush()
            self._jobs.put(None)
            self._worker.join()
            self._jobs = None
        for s in (self.sim, self.readout):
            if s is not None and hasattr(s, "close"):
                s.close()


class ShardedSLAM:
    """One rank of a SLAMNetwork whose neuron populations are split over the ranks (reference
    ``networks/slam.py:241-307``: the loop through the gate, ``:259,306-307``, closes every timestep, so - unlike the
    path integrator alone - the ranks exchange once per timestep).

    * EnsembleArrays (the VCOs, the product arrays of the two circular convolutions) are split over ensembles, the dense
      ensembles ``memory`` / ``recall`` / ``error`` over neurons: rows of their encoders, columns of their decoders - PES
      learns its own columns, Voja moves its own rows.  ``ovc_ens`` stays whole on every rank: its decoded vector reaches
      the product neurons within the timestep (``slam.py:262-266``, ``synapse=None``).
    * Whatever the sharded neurons decode is a partial sum.  Linear operators commute with the sum, and every non-linear
      consumer sits behind a synapse, so ONE all-reduce per timestep - of the vectors the synapse filters and probes read
      (``builder.shard_phases``: ``model.exchange``) - completes them: phase 0, all-reduce, phase 1 (updates + probes).
    * RCCL (torch.distributed "nccl") keeps the exchange in HBM: ``ssn_exchange_pack`` -> ``all_reduce`` ->
      ``ssn_exchange_unpack``; gloo (tests, CPU) goes through the host.

    ``sim_factory(model)`` is injectable: the tests run the orchestration on the NumPy oracle."""

    def __init__(self, sm, rank, world, dt=0.001, dtype="f32", device=0, n_eval_points=None, sim_factory=None, dist=None,
                 replicate=None, host_loop=False, flags=0, cycles=True):
        """``host_loop=True`` keeps round 2's loop (one blocking ssn_run_phase and one blocking exchange per timestep) for A/B
        measurements; the default enqueues the whole run (``run_steps``).  ``cycles=False`` steps one timestep at a time
        (phases 0 / 2 / 1) instead of through the plan pipelined over the exchange (``ssn_cycle_steps``)."""
        if dist is None:
            import torch.distributed as dist
        self.dist, self.rank, self.world, self.dt, self.dtype = dist, rank, world, dt, dtype
        self.sm, self.host_loop, self.cycles = sm, bool(host_loop), bool(cycles)
        if replicate is None:
            replicate = [sm.slam.ovc_ens]
        self.model = build(sm.model, dt=dt, n_eval_points=n_eval_points, neuron_shard=(rank, world), replicate=replicate)
        if sim_factory is None:
            from .simulator import Simulator

            def sim_factory(model):
                return Simulator(None, model=model, dtype=dtype, device=device, flags=flags)
        self.sim = sim_factory(self.model)
        self.n_steps = 0
        self._buf = None
        self._stream = None
        if not hasattr(self.sim, "run_phase"):          # oracle-backed stand-in: the stepper calls back between the phases
            self.sim.set_exchange(self._allreduce_host)

    # -- the exchange ---------------------------------------------------------------------------------------------
    def _allreduce_host(self, vec):
        """In-place sum of a float64 vector over the ranks."""
        if self.world == 1 or not self.dist.is_initialized():
            return
        import torch
        t = torch.from_numpy(vec)
        if self.dist.get_backend() == "nccl":
            d = t.cuda()
            self.dist.all_reduce(d)
            t.copy_(d.cpu())
        else:
            self.dist.all_reduce(t)

    def _exchange(self):
        if self.world == 1:
            return                                   # a single rank's sums are complete
        dev = self._grouped() and self.dist.get_backend() == "nccl" and hasattr(self.sim, "exchange_pack")
        if not dev:
            self.sim.exchange_host(self._allreduce_host)
            return
        import torch
        if self._buf is None:
            tdt = torch.float32 if self.dtype == "f32" else torch.float64
            self._buf = torch.empty(self.sim.exchange_size(), dtype=tdt, device=torch.device("cuda", torch.cuda.current_device()))
        self.sim.exchange_pack(self._buf.data_ptr())        # (blocking: the copies run on the simulator's stream)
        self.dist.all_reduce(self._buf)
        torch.cuda.current_stream().synchronize()
        self.sim.exchange_unpack(self._buf.data_ptr())

    # -- running ----------------------------------------------------------------------------------------------------
    def prepare(self, n_steps):
        self.sim.prepare(n_steps)

    def _grouped(self):
        """True when this runner's collectives are real (a process group of the runner's world size - also of size one)."""
        if not self.dist.is_initialized():
            return False
        return self.world > 1 or self.dist.get_world_size() == 1

    def _stream_ordered(self):
        """True when a run can be enqueued without the host in the loop: the HIP simulator, and either a single rank or the
        RCCL backend (its collectives are ordered on torch's current stream, where the phase graphs are launched too)."""
        if not hasattr(self.sim, "phase_async") or self.host_loop:
            return False
        if self.world == 1:
            return True
        return self.dist.is_initialized() and self.dist.get_backend() == "nccl"

    def _cycle_steps(self):
        """Timesteps per cycle of the plan pipelined over the exchange (0: not offered, or switched off with ``cycles=False``)."""
        if not self.cycles or not hasattr(self.sim, "cycle_steps"):
            return 0
        return int(self.sim.cycle_steps())

    def _exchange_buffer(self):
        """Device buffer of the per-timestep exchange (None for a single rank: its sums are complete)."""
        if self.world == 1:
            return None
        if self._buf is None:
            import torch
            tdt = torch.float32 if self.dtype == "f32" else torch.float64
            self._buf = torch.zeros(self.sim.exchange_size(), dtype=tdt, device=torch.device("cuda", torch.cuda.current_device()))
        return self._buf.data_ptr()

    def capture(self):
        """Build the stream-ordered step graphs now instead of at the first timestep of the first run (optional; several
        ranks living in one process - tests - do this one after the other before their threads start stepping)."""
        if self._stream_ordered():
            self.sim.phase_async(-1, self._exchange_buffer(), None)

    def _agree(self, err):
        """Every rank learns whether any rank failed before the collectives of a run start (a rank that raised would leave its
        peers waiting in the first all-reduce until the RCCL watchdog fires)."""
        failed = err is not None
        if self._grouped():
            import torch
            flag = torch.tensor([1.0 if failed else 0.0])
            if self.dist.get_backend() == "nccl":
                flag = flag.cuda()
            self.dist.all_reduce(flag, op=self.dist.ReduceOp.MAX)
            failed = bool(flag.item() > 0)
        if failed:
            raise nengo.SimulationError(f"sharded SLAM run refused on rank {self.rank}: {err}" if err is not None else
                                        "sharded SLAM run refused: another rank failed its pre-flight check") from err

    def run_steps(self, n):
        n = int(n)
        if n <= 0:
            return
        if not hasattr(self.sim, "run_phase"):
            self.sim.run_steps(n)
            self.n_steps += n
            return
        err = None
        if getattr(self.sim, "_prepared_until", 0) < self.sim.n_steps + n:
            err = nengo.SimulationError(f"rank {self.rank}: call prepare(n_steps) before run_steps (inputs are tabulated ahead of the run)")
        if self._stream_ordered():
            # The whole run is enqueued: per timestep ONE graph launch ([unpack the reduced sums] -> updates of step s -> step
            # s + 1 up to its exchange -> [pack the partial sums]) and ONE all-reduce, all ordered on one stream - Python only
            # enqueues (nothing calls back into the host from the step loop); the single wait is at the end.
            import torch
            buf = None
            if self.world > 1:
                try:
                    buf = self._exchange_buffer()
                except Exception as e:               # noqa: BLE001 - agreed on collectively, then re-raised
                    err = err or e
                self._agree(err)
            elif err is not None:
                raise err
            # A stream of the runner's own, made torch's current stream for the run: the phase graphs are launched on it and
            # torch.distributed orders its collectives on the CURRENT stream.  (Not torch's default stream: its handle is 0,
            # which ssn_phase_async reads as "the simulator's own stream" - the graphs and the all-reduce would then run on
            # two unordered streams.)
            if self._stream is None:
                self._stream = torch.cuda.Stream()
            self._stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(self._stream):
                stream = self._stream.cuda_stream
                assert stream, "a created HIP stream has a non-zero handle"
                # Whatever fails while the run is being enqueued (a phase refused, a collective that raises), the stream is
                # drained and the library's in-flight mark cleared before the error travels on: reset / read_buffer /
                # set_table synchronise the simulator's own stream only and must not race with graphs still queued here.
                enqueue_error = None
                try:
                    # whole cycles of the plan pipelined over the exchange first (segment, all-reduce, segment, ...), then
                    # what is left one timestep at a time (0, x, 2, x, ..., 1)
                    cyc = self._cycle_steps()
                    done = 0
                    while cyc and n - done >= cyc:
                        for k in range(cyc + 1):
                            self.sim.phase_async(3, buf, stream)
                            if k < cyc and self.world > 1:
                                self.dist.all_reduce(self._buf)
                        done += cyc
                    if done < n:
                        self.sim.phase_async(0, buf, stream)
                        for i in range(done, n):
                            if self.world > 1:
                                self.dist.all_reduce(self._buf)
                            self.sim.phase_async(2 if i + 1 < n else 1, buf, stream)
                except BaseException as e:           # noqa: BLE001 - re-raised below, after the stream has been drained
                    enqueue_error = e
                try:
                    self.sim.phase_sync(stream)
                except Exception:
                    if enqueue_error is None:
                        raise
                if enqueue_error is not None:
                    raise enqueue_error
            torch.cuda.current_stream().wait_stream(self._stream)
        else:
            self._agree(err)
            cyc = self._cycle_steps()
            done = 0
            while cyc and n - done >= cyc:
                for k in range(cyc + 1):
                    self.sim.run_phase(3)
                    if k < cyc:
                        self._exchange()
                done += cyc
            if done < n:
                self.sim.run_phase(0)
                for i in range(done, n):
                    self._exchange()
                    # the updates of this timestep and the next one up to its exchange share a launch (one host round trip)
                    self.sim.run_phase(2 if i + 1 < n else 1)
        self.n_steps += n

    # -- results ----------------------------------------------------------------------------------------------------
    def probe_data(self, probe=None):
        """A signal probe (by default the filtered PathIntegration output): identical on every rank after the exchange."""
        return self.sim.data[self.sm.probe if probe is None else probe]

    def _gather_rows(self, local, n_total):
        """(per, ...) local share of every rank -> (n_total, ...) on every rank."""
        import torch
        local = np.ascontiguousarray(local, dtype=np.float64)
        if self.world == 1 or not self.dist.is_initialized():
            return local[:n_total]
        t = torch.from_numpy(local)
        if self.dist.get_backend() == "nccl":
            t = t.cuda()
        outs = [torch.empty_like(t) for _ in range(self.world)]
        self.dist.all_gather(outs, t)
        return torch.cat(outs, dim=0).cpu().numpy()[:n_total]

    def learned_decoders(self, conn):
        """The whole PES-learned decoder matrix (size_out, n_neurons) of ``conn``, gathered from the ranks' column shares."""
        bc = self.model.params[conn]
        local = self.sim.read_buffer(bc.learned_buffer)
        n = _contig_obj(conn.pre).n_neurons
        return self._gather_rows(local.T, n).T

    def learned_encoders(self, ens):
        """The whole Voja-moved scaled-encoder matrix (n_neurons, d) of ``ens``."""
        local = self.sim.read_buffer(self.model.params[ens].encoder_buffer)
        return self._gather_rows(local, ens.n_neurons)

    def close(self):
        if hasattr(self.sim, "close"):
            self.sim.close()


def _contig_obj(x):
    return x.obj if hasattr(x, "obj") and hasattr(x, "slice") else x
