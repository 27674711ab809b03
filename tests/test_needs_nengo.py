"""Cross-check against REAL nengo (last row) - skipped unless the genuine package is importable.

nengo is an un-vendored, unpinned dependency of the reference (``setup.py:21-27``) and is absent from the build
container and the GPU box, so this test normally reports "skipped".  On a machine that has nengo it rebuilds a
model that was built here - the SAME sampled encoders / gains / biases and the SAME solved decoders, injected through
``Ensemble(encoders=, gain=, bias=)`` and ``NoSolver`` - as a genuine ``nengo.Network``, runs ``nengo.Simulator`` on it,
and compares the probe trajectories with ``oracle/stepper.py``.  That pins the restated step semantics (LIF, Lowpass,
one-step synapse delay, PES / Voja timing) to nengo itself; until it has run somewhere, parity is "unpinned".

    pip install nengo && python -m pytest tests/test_needs_nengo.py -m needs_nengo
"""
import numpy as np
import pytest

import sspslam_amd.frontend as fe
from sspslam_amd import harness as H
from sspslam_amd.builder import build
from oracle import OracleSimulator
from oracle.graphwalk import _kind, _obj_and_slice

from helpers import small_pathint


def _real_nengo():
    try:
        import nengo
    except ImportError:
        return None
    if getattr(nengo, "_sspslam_amd_compat", False) or not hasattr(nengo, "builder"):
        return None            # our own object model registered under the name: not the real thing
    return nengo


pytestmark = [pytest.mark.needs_nengo,
              pytest.mark.skipif(_real_nengo() is None, reason="real nengo is not installed (reference dependency, unpinned)")]


def to_real_nengo(network, model):
    """Front-end Network + BuiltModel -> (genuine nengo.Network, {our probe: nengo probe}) with every sampled /
    solved parameter injected, so that nengo steps exactly the arrays the oracle and the GPU step."""
    nengo = _real_nengo()
    objs = {}
    with nengo.Network(seed=0) as net:
        for n in network.all_nodes:
            out = n.output
            if out is None:
                objs[id(n)] = nengo.Node(size_in=n.size_in, label=n.label)
            elif callable(out):
                objs[id(n)] = nengo.Node(out, size_in=n.size_in or None, size_out=n.size_out, label=n.label)
            else:
                objs[id(n)] = nengo.Node(np.asarray(out, dtype=float), label=n.label)
        for e in network.all_ensembles:
            be = model.params[e]
            nd = be.neuron
            nt = {"lif": lambda: nengo.LIF(tau_rc=nd["tau_rc"], tau_ref=nd["tau_ref"], min_voltage=nd["min_voltage"],
                                           amplitude=nd["amplitude"]),
                  "lifrate": lambda: nengo.LIFRate(tau_rc=nd["tau_rc"], tau_ref=nd["tau_ref"], amplitude=nd["amplitude"]),
                  "relu": lambda: nengo.RectifiedLinear(amplitude=nd["amplitude"])}[nd["type"]]()
            objs[id(e)] = nengo.Ensemble(e.n_neurons, e.dimensions, radius=be.radius, encoders=be.encoders,
                                         gain=be.gain, bias=be.bias, neuron_type=nt, normalize_encoders=False,
                                         label=e.label)

        def end(x, post):
            obj, idx = _obj_and_slice(x)
            k = _kind(obj)
            if k == "neurons":
                r = objs[id(obj.ensemble)].neurons
            elif k == "rule":
                r = rules[id(obj)]
            else:
                r = objs[id(obj)]
            return r if idx is None else r[slice(int(idx[0]), int(idx[-1]) + 1)]

        rules = {}
        pending = []
        for c in network.all_connections:
            bc = model.params.get(c)
            if bc is None:
                continue
            if _kind(_obj_and_slice(c.post)[0]) == "rule":
                pending.append(c)          # its target exists once the learned connection does
                continue
            pre_obj = _obj_and_slice(c.pre)[0]
            syn = None if c.synapse is None else nengo.Lowpass(c.synapse.tau)
            kw = {}
            rt = getattr(c, "learning_rule_type", None)
            if rt is not None:
                if type(rt).__name__ == "PES":
                    kw["learning_rule_type"] = nengo.PES(rt.learning_rate, pre_synapse=None if rt.pre_synapse is None
                                                         else nengo.Lowpass(rt.pre_synapse.tau))
                else:
                    kw["learning_rule_type"] = nengo.Voja(rt.learning_rate, post_synapse=None)
            if _kind(pre_obj) == "ensemble":
                W = np.array(bc.weights, dtype=float)            # transform @ decoders: (size_out, n)
                rc = nengo.Connection(objs[id(pre_obj)], end(c.post, True), synapse=syn,
                                      solver=nengo.solvers.NoSolver(W.T), function=lambda x, _n=W.shape[0]: np.zeros(_n), **kw)
            else:
                rc = nengo.Connection(end(c.pre, False), end(c.post, True), synapse=syn,
                                      transform=np.asarray(c.transform, dtype=float), **kw)
            objs[id(c)] = rc
            if rt is not None:
                rules[id(c.learning_rule)] = rc.learning_rule
        for c in pending:
            syn = None if c.synapse is None else nengo.Lowpass(c.synapse.tau)
            pre_obj = _obj_and_slice(c.pre)[0]
            if _kind(pre_obj) == "ensemble":
                W = np.array(model.params[c].weights, dtype=float)
                nengo.Connection(objs[id(pre_obj)], end(c.post, True), synapse=syn, solver=nengo.solvers.NoSolver(W.T),
                                 function=lambda x, _n=W.shape[0]: np.zeros(_n))
            else:
                nengo.Connection(end(c.pre, False), end(c.post, True), synapse=syn,
                                 transform=np.asarray(c.transform, dtype=float))
        probes = {}
        for p in network.all_probes:
            obj, idx = _obj_and_slice(p.target)
            k = _kind(obj)
            syn = None if p.synapse is None else nengo.Lowpass(p.synapse.tau)
            if k == "connection":
                probes[p] = nengo.Probe(objs[id(obj)], "weights", sample_every=p.sample_every)
            elif k == "rule":
                probes[p] = nengo.Probe(rules[id(obj)], "scaled_encoders", sample_every=p.sample_every)
            elif k == "ensemble":
                continue               # decoded-output probes would need the probe's own decoders injected
            else:
                probes[p] = nengo.Probe(end(p.target, False), synapse=syn, sample_every=p.sample_every)
    return net, probes


def _compare(network, model, steps, atol):
    nengo = _real_nengo()
    net, probes = to_real_nengo(network, model)
    ref = OracleSimulator(model)
    ref.run_steps(steps)
    with nengo.Simulator(net, dt=model.dt, progress_bar=False, optimize=False) as sim:
        sim.run_steps(steps)
        for i, p in enumerate(model.probes):
            if p["probe"] in probes:
                got = np.asarray(sim.data[probes[p["probe"]]], dtype=float)
                want = ref.probe_data(i)
                np.testing.assert_allclose(got.reshape(want.shape), want, atol=atol, rtol=0, err_msg=repr(p["probe"]))


def test_pathintegration_oracle_equals_real_nengo():
    pm = small_pathint(ssp_dim=55, n=60, T=10.0, limit=0.2)
    _compare(pm.model, build(pm.model), 300, 1e-9)


def test_slam_oracle_equals_real_nengo():
    s = H.make_ssp_space(2, 55)
    path, vels = H.make_random_path(20.0, limit=0.1, seed=0)
    sm = H.make_slam_model(s, path, vels, n_landmarks=10, pi_n_neurons=60, mem_n_neurons=120, circonv_n_neurons=30,
                           view_rad=0.6, weights_sample_every=0.05)
    with sm.model:
        fe.Probe(sm.slam.assomemory.conn_in.learning_rule, "scaled_encoders", sample_every=0.05)
    _compare(sm.model, build(sm.model), 300, 1e-9)
