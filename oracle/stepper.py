"""ORACLE - TEST INFRASTRUCTURE ONLY.  CPU restatement (NumPy float64) of the simulator step loop.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may import
this package, and only as the checker / the timed CPU baseline.  The product path
(``sspslam_amd.simulator``) never imports it and fails loudly when the HIP library is missing.

What is restated, and from where
--------------------------------
The reference repository contains no simulator: every per-timestep operation of its hot path is
executed by the third-party package ``nengo`` (PyPI ``nengo``, **unpinned** in the reference's
``setup.py:21-27``; >= 3.1 implied by ``sspspace.py:5`` and ``slam.py:206``), which is neither
vendored under the reference tree nor installable here.  This file restates nengo's *published*
reference-simulator algorithm (``nengo/simulator.py``, ``nengo/neurons.py``
``LIF.step``, ``nengo/synapses.py`` ``Lowpass``, ``nengo/builder/learning_rules.py`` ``SimPES`` /
``SimVoja``, ``nengo/builder/operator.py``) as an interpreter for the frozen operator list that
``sspslam_amd.builder`` produces, anchored on the reference's own call sites:

* step order / time:   A.1 - ``step += 1; t = step*dt``; operators; then probe sampling
                       (node functions see ``t = dt`` first: ``experiments/run_pathint.py:134``)
* LIF neuron step:     A.4 (``experiments/run_pathint.py:113-114`` selects ``nengo.LIF()``)
* Lowpass synapse:     A.6 (taus at ``networks/pathintegration.py:171,182,191``, ``slam.py:271-307``)
* PES / Voja:          A.7 / A.8 (``networks/associativememory.py:31,41``)
* clean-up / gate:     ``networks/slam.py:212-215`` / ``:233-237`` (in-tree, pinned by golden vectors)

PARITY PINNING: the pieces that exist in the reference tree (SSP algebra, Fourier-layout and
binding matrices, ``feedback``, the gate and clean-up node functions, input tables, network
topology) are pinned by golden vectors captured from the reference itself
(``tests/golden/make_golden.py``).  The nengo step semantics above have **no** fixture anywhere in
the reference (it has no tests) and nengo cannot be run here: for them this oracle is
**parity unpinned** - a restatement of the published algorithm, checked only by the analytic
properties in ``tests/test_oracle.py`` (LIF rate curve, filter step response, oscillator frequency).

Execution structure mirrors nengo's CPU backend after its operator-merge pass: one vectorised NumPy
call per merged operator, a Python loop over operators per step - which is what makes it a fair
"nengo-style CPU" timing baseline.
"""
import numpy as np


def lif_step(J, V, R, dt, tau_rc, tau_ref, min_voltage):
    """One LIF step in place on voltage ``V`` and refractory time ``R``; returns the spike mask.

    Restates nengo ``LIF.step``, operation for operation.
    """
    R -= dt
    delta_t = np.clip(dt - R, 0.0, dt)
    V -= (J - V) * np.expm1(-delta_t / tau_rc)
    spiked = V > 1
    t_spike = dt + tau_rc * np.log1p(-(V[spiked] - 1.0) / (J[spiked] - 1.0))
    V[V < min_voltage] = min_voltage
    V[spiked] = 0.0
    R[spiked] = tau_ref + t_spike
    return spiked


def lif_rate(J, tau_rc, tau_ref):
    out = np.zeros_like(J)
    m = J > 1
    out[m] = 1.0 / (tau_ref + tau_rc * np.log1p(1.0 / (J[m] - 1.0)))
    return out


def neuron_activity(neuron, J, V, R, dt):
    """Unit-amplitude activity: spike indicator (LIF) or rate (LIFRate / ReLU)."""
    t = neuron["type"]
    if t == "lif":
        return lif_step(J, V, R, dt, neuron["tau_rc"], neuron["tau_ref"], neuron["min_voltage"]).astype(J.dtype)
    if t == "lifrate":
        return lif_rate(J, neuron["tau_rc"], neuron["tau_ref"])
    if t == "relu":
        return np.maximum(J, 0.0)
    raise ValueError(t)


class OracleSimulator:
    """Interprets a BuiltModel step by step.  ``dtype`` float64 = nengo's CPU backend arithmetic."""

    def __init__(self, model, dtype=np.float64):
        self.model = model
        self.dt = model.dt
        self.dtype = np.dtype(dtype)
        self.reset()

    def exchange_hook(self, sig, ranges):
        """All-reduce(sum) of ``sig[lo:hi]`` over the ranks of a neuron-sharded run; installed by the sharded runner."""
        raise RuntimeError("this model was built with neuron_shard=...: step it through sharding.ShardedSLAM, which supplies the exchange")

    def reset(self):
        m = self.model
        self.sig = m.sig_init.astype(self.dtype).copy()
        self.buf = []
        for b, meta in zip(m.buffers, m.buffer_meta):
            if meta["role"] in ("state", "learned"):
                self.buf.append(np.array(b, dtype=self.dtype))
            elif b.dtype.kind in "iu":
                self.buf.append(b)
            else:
                self.buf.append(np.asarray(b, dtype=self.dtype))
        self.n_steps = 0
        self.probe_rows = [[] for _ in m.probes]

    # ------------------------------------------------------------------------------------
    def run_steps(self, n):
        for _ in range(int(n)):
            self.step()

    def run(self, t):
        self.run_steps(int(np.round(float(t) / self.dt)))

    def trange(self):
        return self.dt * np.arange(1, self.n_steps + 1)

    def probe_data(self, i):
        rows = self.probe_rows[i]
        return np.array(rows, dtype=np.float64) if rows else np.zeros((0, 0))

    def step(self):
        m, sig, buf, dt = self.model, self.sig, self.buf, self.dt
        self.n_steps += 1
        t = self.n_steps * dt
        # Probes of learned signals ("weights", "scaled_encoders") read the signal as the step leaves it in nengo, where
        # `target += delta` is an inc at the START of the next step (Appendix A.7 / A.8): the sample of step t holds the
        # deltas of steps < t.  The pes / voja operators below add this step's delta at once, so sample first.
        learned_samples = {i: np.array(buf[self._probe_buffer(p)], dtype=np.float64)
                           for i, p in enumerate(m.probes) if "src" not in p and self.n_steps % p["every"] == 0}
        exchanged = not getattr(m, "exchange", None)
        for o in m.ops:
            k = o["kind"]
            if not exchanged and o.get("phase", 0) == 1:
                # neuron-sharded model (builder.shard_phases): the partial sums are completed before the first update
                self.exchange_hook(sig, m.exchange)
                exchanged = True
            if k == "fill":
                sig[o["dst"]:o["dst"] + o["len"]] = o["value"]
            elif k == "table":
                tb = m.tables[o["table"]]
                sig[o["dst"]:o["dst"] + o["width"]] = np.asarray(tb["fn"](t), dtype=self.dtype).reshape(-1)
            elif k == "axpy":
                src = sig[o["src"]:o["src"] + o["len"]]
                if o["mode"] == "set":
                    sig[o["dst"]:o["dst"] + o["len"]] = o["alpha"] * src
                else:
                    sig[o["dst"]:o["dst"] + o["len"]] += o["alpha"] * src
            elif k == "lincomb":      # folded linear glue (glue.py): dst = self * dst + (const + sum_k alpha_k * src_k)
                acc = np.full(o["len"], o["const"], dtype=self.dtype)
                for src, alpha in zip(o["srcs"], o["alphas"]):
                    acc = acc + alpha * sig[src:src + o["len"]]
                if o["self"]:
                    sig[o["dst"]:o["dst"] + o["len"]] = o["self"] * sig[o["dst"]:o["dst"] + o["len"]] + acc
                else:
                    sig[o["dst"]:o["dst"] + o["len"]] = acc
            elif k == "matvec":
                y = buf[o["w"]] @ sig[o["src"]:o["src"] + o["cols"]]
                if o["mode"] == "set":
                    sig[o["dst"]:o["dst"] + o["rows"]] = y
                else:
                    sig[o["dst"]:o["dst"] + o["rows"]] += y
            elif k == "lowpass":
                d = sig[o["dst"]:o["dst"] + o["len"]]
                d *= o["a"]
                d += (1.0 - o["a"]) * o["gain"] * sig[o["src"]:o["src"] + o["len"]]
            elif k == "ensarray":
                K, n, din = o["K"], o["n"], o["din"]
                x = sig[o["x"]:o["x"] + K * din].reshape(K, din)
                J = buf[o["bias"]] + np.einsum("kdn,kd->kn", buf[o["enc"]], x)
                a = neuron_activity(o["neuron"], J, buf[o["v"]], buf[o["r"]], dt)
                dec = np.einsum("krn,kn->kr", buf[o["dec"]], a)
                sig[buf[o["dst_idx"]].reshape(-1)] = dec.reshape(-1)
            elif k == "neurons":
                J = sig[o["j"]:o["j"] + o["n"]]
                a = neuron_activity(o["neuron"], J.copy(), buf[o["v"]], buf[o["r"]], dt)
                sig[o["out"]:o["out"] + o["n"]] = o["amp"] * a
            elif k == "pes":
                err = sig[o["err"]:o["err"] + o["rows"]]
                act = sig[o["act"]:o["act"] + o["cols"]]
                buf[o["w"]] += o["kappa"] * np.outer(err, act)
            elif k == "voja":
                E = buf[o["w"]]
                a = sig[o["spk"]:o["spk"] + o["rows"]]
                key = sig[o["key"]:o["key"] + o["cols"]]
                learning = 1.0 + sig[o["learn"]]
                nz = np.nonzero(a)[0]
                if nz.size:
                    scale = buf[o["scale_buf"]]
                    E[nz] += o["lr_dt"] * learning * (
                        (scale[nz] * a[nz])[:, None] * key[None, :] - a[nz, None] * E[nz])
            elif k == "cleanup":
                # S[argmax(S @ x)] (slam.py:212-215).  The products are summed column by column so the
                # result does not depend on a BLAS's blocking: while the path integrator's output holds only
                # its DC term every grid point has the same similarity up to rounding, and the pick is
                # decided by summation order.
                T = buf[o["w"]]
                x = sig[o["src"]:o["src"] + o["cols"]]
                sims = np.zeros(T.shape[0], dtype=T.dtype)
                for j in range(T.shape[1]):
                    sims += T[:, j] * x[j]
                sig[o["dst"]:o["dst"] + o["cols"]] = T[int(np.argmax(sims))]
            elif k == "gate":
                d = o["d"]
                x = sig[o["src"]:o["src"] + 2 * d + 1]
                est, cur, flag = x[:d], x[d:2 * d], x[2 * d]
                if abs(flag) <= 1e-3 and float(est @ cur) > o["thres"]:
                    sig[o["dst"]:o["dst"] + d] = o["rate"] * (est - cur)
                else:
                    sig[o["dst"]:o["dst"] + d] = 0.0
            else:
                raise ValueError(f"unknown op {k}")
        for i, p in enumerate(m.probes):
            if self.n_steps % p["every"]:
                continue
            if "src" in p:
                self.probe_rows[i].append(sig[p["src"]:p["src"] + p["width"]].astype(np.float64))
            else:
                self.probe_rows[i].append(learned_samples[i])

    def _probe_buffer(self, p):
        b = p["buf"]
        if isinstance(b, tuple):
            b = self.model.params[p["ens"]].encoder_buffer
        return b
