"""ORACLE - TEST INFRASTRUCTURE ONLY.  Independent check of the *lowering*: a naive simulator that walks the
front-end object graph directly.

``oracle/stepper.py`` interprets the operator list that ``sspslam_amd.builder`` lowers for the GPU, so a lowering
mistake (synapse order, the one-step delay of a filtered connection, slice aliasing, the order in which merged
operators accumulate) would be reproduced identically on both sides of every parity test.  This module does not
look at that operator list at all.  It keeps one Python object per Node, Ensemble and Connection of the network,
and steps them the way nengo's reference simulator orders its operators (per signal
*sets -> incs -> reads -> updates*):

  1. ``step += 1; t = step * dt``                                              (A.1)
  2. incs of learned targets: ``W += delta`` / ``scaled_encoders += delta`` with the delta computed at the END of
     the previous step (A.7, A.8: the delta is an update, the ``+=`` copy an inc)
  3. every input accumulator is reset; every connection WITH a synapse adds the filter state left by the previous
     step (reads happen before the update: one dt of delay per synapse, A.5)
  4. objects in topological order over the ``synapse=None`` connections: Node functions / passthroughs (A.2),
     ensembles ``J = bias + scaled_encoders . x + direct neuron input`` -> neuron step (A.3, A.4); each outgoing
     connection computes its ``weighted`` output and, if it has no synapse, adds it to its target at once (A.5)
  5. updates: PES delta from the error and the *previous* step's filtered activities, Voja delta from this step's
     spikes and key (A.7, A.8); then every synapse state ``y <- a y + (1 - a) u`` with this step's ``u`` (A.6)
  6. probes are sampled (a probe's own synapse is one more filter updated in 5)

What it takes from the builder are *parameters only* - the sampled encoders / gains / biases of each ensemble and
the solved decoder matrices of each connection (``model.params``) - never operators, offsets or the schedule.
Which reference call sites it follows: ``networks/pathintegration.py:162-191`` (EnsembleArray, recurrent
connections, default-synapse read-in / read-out), ``networks/associativememory.py:30-54`` (Voja, PES, direct
neuron inhibition), ``networks/slam.py:241-307`` (function nodes with inputs are called as Python functions here,
exactly as nengo's reference simulator would call them).

Like the stepper this is a restatement of nengo's published semantics, not nengo: parity unpinned (see stepper.py).
"""
import numpy as np

from .stepper import lif_rate, lif_step


def _obj_and_slice(x):
    """(object, index array | None) for an object or a slice view of one."""
    if hasattr(x, "obj") and hasattr(x, "indices"):
        return x.obj, np.asarray(x.indices, dtype=np.int64)
    if hasattr(x, "obj") and hasattr(x, "slice"):
        size = max(x.obj.size_in, x.obj.size_out)
        return x.obj, np.atleast_1d(np.arange(size)[x.slice])
    return x, None


def _kind(obj):
    n = type(obj).__name__
    return {"Node": "node", "Ensemble": "ensemble", "Neurons": "neurons", "LearningRule": "rule",
            "Connection": "connection"}.get(n, n)


class _Node:
    def __init__(self, node):
        self.node = node
        self.inp = np.zeros(node.size_in) if node.size_in else None
        self.out = np.zeros(node.size_out)


class _Ens:
    def __init__(self, ens, built):
        self.ens, self.built = ens, built
        self.inp = np.zeros(ens.dimensions)
        self.direct = np.zeros(ens.n_neurons)
        self.E = np.array(built.scaled_encoders, dtype=float)      # learned by Voja
        self.V = np.zeros(ens.n_neurons)
        self.R = np.zeros(ens.n_neurons)
        self.out = np.zeros(ens.n_neurons)                          # neuron output (spikes * amplitude / dt, or rates)


class _Conn:
    pass


class GraphWalkSimulator:
    def __init__(self, network, model, dt=None):
        """``network``: the front-end Network; ``model``: the BuiltModel it was built into (parameters only)."""
        self.dt = float(model.dt if dt is None else dt)
        self.net, self.model = network, model
        self.n_steps = 0
        self.nodes = {id(n): _Node(n) for n in network.all_nodes}
        self.ens = {id(e): _Ens(e, model.params[e]) for e in network.all_ensembles}
        self.conns, self.rules = [], []
        self.skipped = []
        for c in network.all_connections:
            bc = model.params.get(c)
            if bc is None:
                self.skipped.append(c)       # the builder found it a dead end; checked below
                continue
            self.conns.append(self._make_conn(c, bc))
        self._check_skipped(network)
        self._order()
        self.probes = []
        for p in network.all_probes:
            self.probes.append(self._make_probe(p))

    # -- construction ------------------------------------------------------------------------------------------
    def _state(self, obj):
        k = _kind(obj)
        if k == "node":
            return self.nodes[id(obj)]
        if k == "ensemble":
            return self.ens[id(obj)]
        if k == "neurons":
            return self.ens[id(obj.ensemble)]
        raise ValueError(f"no state for {obj!r}")

    def _make_conn(self, c, bc):
        s = _Conn()
        s.c = c
        s.pre_obj, s.pre_idx = _obj_and_slice(c.pre)
        s.post_obj, s.post_idx = _obj_and_slice(c.post)
        s.pre_kind, s.post_kind = _kind(s.pre_obj), _kind(s.post_obj)
        s.size_out = int(c.post.size_in)
        s.weighted = np.zeros(s.size_out)
        syn = c.synapse
        s.a = None if syn is None else (np.exp(-self.dt / syn.tau) if syn.tau > 0 else 0.0)
        s.filtered = None if syn is None else np.zeros(s.size_out)
        s.rule = None
        rt = getattr(c, "learning_rule_type", None)
        T = np.asarray(c.transform, dtype=float)
        if s.pre_kind == "ensemble":
            # decoded connection: weights = transform @ decoders(function), shape (size_out, n_neurons)
            s.W = np.array(bc.weights, dtype=float)
            assert s.W.shape == (s.size_out, s.pre_obj.n_neurons), (c, s.W.shape)
        else:
            s.W = float(T) if T.ndim == 0 else T
        if rt is not None:
            r = _Conn()
            r.kind = type(rt).__name__
            r.conn = s
            r.lr = float(rt.learning_rate)
            r.obj = c.learning_rule
            if r.kind == "PES":
                n = s.pre_obj.n_neurons
                r.inp = np.zeros(s.size_out)
                ps = rt.pre_synapse
                r.pre_a = 0.0 if ps is None or ps.tau <= 0 else np.exp(-self.dt / ps.tau)
                r.pre_filtered = np.zeros(n)
                r.delta = np.zeros_like(s.W)
            elif r.kind == "Voja":
                post = s.post_obj
                assert s.post_kind == "ensemble" and rt.post_synapse is None
                r.inp = np.zeros(1)
                be = self.model.params[post]
                r.scale = np.asarray(be.gain, dtype=float) / be.radius
                r.delta = np.zeros_like(self.ens[id(post)].E)
            else:
                raise ValueError(r.kind)
            s.rule = r
            self.rules.append(r)
        return s

    def _check_skipped(self, network):
        """A connection without built parameters must end in a passthrough node nothing reads."""
        has_out = {id(_obj_and_slice(c.pre)[0]) for c in network.all_connections if c not in self.skipped}
        probed = {id(_obj_and_slice(p.target)[0]) for p in network.all_probes}
        for c in self.skipped:
            post = _obj_and_slice(c.post)[0]
            assert _kind(post) == "node" and getattr(post, "output", None) is None, c
            assert id(post) not in has_out and id(post) not in probed, c

    def _rule_of(self, obj):
        for r in self.rules:
            if r.obj is obj:
                return r
        raise KeyError(obj)

    def _order(self):
        """Topological order of nodes and ensembles over the synapse=None connections."""
        verts = {**{k: v for k, v in self.nodes.items()}, **{k: v for k, v in self.ens.items()}}
        succ = {k: set() for k in verts}
        indeg = {k: 0 for k in verts}
        self.out_conns = {k: [] for k in verts}
        for s in self.conns:
            pre = s.pre_obj.ensemble if s.pre_kind == "neurons" else s.pre_obj
            self.out_conns[id(pre)].append(s)
            if s.a is not None or s.post_kind == "rule":
                continue
            post = s.post_obj.ensemble if s.post_kind == "neurons" else s.post_obj
            if id(post) not in succ[id(pre)]:
                succ[id(pre)].add(id(post))
                indeg[id(post)] += 1
        ready = [k for k in verts if indeg[k] == 0]
        order = []
        while ready:
            k = ready.pop()
            order.append(verts[k])
            for j in succ[k]:
                indeg[j] -= 1
                if indeg[j] == 0:
                    ready.append(j)
        if len(order) != len(verts):
            raise ValueError("a loop of synapse=None connections")
        self.order = order

    def _make_probe(self, p):
        s = _Conn()
        s.p = p
        s.obj, s.idx = _obj_and_slice(p.target)
        s.kind = _kind(s.obj)
        s.every = 1 if p.sample_every is None else max(1, int(round(p.sample_every / self.dt)))
        s.rows = []
        s.a = None
        if p.synapse is not None:
            s.a = np.exp(-self.dt / p.synapse.tau) if p.synapse.tau > 0 else 0.0
            s.state = None
        if s.kind == "ensemble":
            bp = self.model.params.get(p)
            s.W = None if bp is None else np.asarray(bp.weights, dtype=float)
        return s

    # -- stepping ------------------------------------------------------------------------------------------------
    def _add(self, s, vec):
        """Add a connection's output to its target's input accumulator."""
        if s.post_kind == "rule":
            tgt = self._rule_of(s.post_obj).inp
        elif s.post_kind == "neurons":
            tgt = self.ens[id(s.post_obj.ensemble)].direct
        else:
            tgt = self._state(s.post_obj).inp
        if s.post_idx is None:
            tgt += vec
        else:
            tgt[s.post_idx] += vec

    def _neuron_out(self, es, J):
        nd = es.built.neuron
        if nd["type"] == "lif":
            spiked = lif_step(J, es.V, es.R, self.dt, nd["tau_rc"], nd["tau_ref"], nd["min_voltage"])
            return spiked * (nd["amplitude"] / self.dt)
        if nd["type"] == "lifrate":
            return nd["amplitude"] * lif_rate(J, nd["tau_rc"], nd["tau_ref"])
        return nd["amplitude"] * np.maximum(J, 0.0)

    def step(self):
        self.n_steps += 1
        t = self.n_steps * self.dt
        # 2. learned targets take the delta of the previous step
        for r in self.rules:
            if r.kind == "PES":
                r.conn.W += r.delta
            else:
                self.ens[id(r.conn.post_obj)].E += r.delta
        # 3. resets, then the filter states of the previous step
        for ns in self.nodes.values():
            if ns.inp is not None:
                ns.inp[:] = 0.0
        for es in self.ens.values():
            es.inp[:] = 0.0
            es.direct[:] = 0.0
        for r in self.rules:
            r.inp[:] = 1.0 if r.kind == "Voja" else 0.0
        for s in self.conns:
            if s.a is not None:
                self._add(s, s.filtered)
        # 4. the walk
        for v in self.order:
            if isinstance(v, _Node):
                out = v.node.output
                if out is None:
                    v.out = v.inp.copy()
                elif callable(out):
                    v.out = np.asarray(out(t) if v.inp is None else out(t, v.inp.copy()), dtype=float).reshape(-1)
                else:
                    v.out = np.asarray(out, dtype=float).reshape(-1)
            else:
                J = es_bias(v) + v.E @ v.inp + v.direct
                v.out = self._neuron_out(v, J)
            for s in self.out_conns[id(v.node if isinstance(v, _Node) else v.ens)]:
                if s.pre_kind == "ensemble":
                    u = s.W @ v.out
                else:
                    x = v.out if s.pre_idx is None else v.out[s.pre_idx]
                    u = s.W * x if np.ndim(s.W) == 0 else s.W @ x
                s.weighted = np.asarray(u, dtype=float).reshape(-1)
                if s.a is None:
                    self._add(s, s.weighted)
        # 5. updates: learning-rule deltas first (they read the filtered activities of the previous step) ...
        for r in self.rules:
            s = r.conn
            if r.kind == "PES":
                n = s.pre_obj.n_neurons
                r.delta = -(r.lr * self.dt / n) * np.outer(r.inp, r.pre_filtered)
            else:
                es = self.ens[id(s.post_obj)]
                a = es.out
                learning = float(r.inp[0])
                r.delta = r.lr * self.dt * learning * ((r.scale * a)[:, None] * s.weighted[None, :] - a[:, None] * es.E)
        # ... then the synapses
        for s in self.conns:
            if s.a is not None:
                s.filtered = s.a * s.filtered + (1.0 - s.a) * s.weighted
        for r in self.rules:
            if r.kind == "PES":
                spikes = self.ens[id(r.conn.pre_obj)].out
                r.pre_filtered = r.pre_a * r.pre_filtered + (1.0 - r.pre_a) * spikes
        # 6. probes
        for s in self.probes:
            if s.kind == "node":
                u = self.nodes[id(s.obj)].out
            elif s.kind == "neurons":
                u = self.ens[id(s.obj.ensemble)].out
            elif s.kind == "ensemble":
                u = s.W @ self.ens[id(s.obj)].out
            elif s.kind == "connection":          # "weights": the signal as the step leaves it (delta of this step not yet added)
                u = self._conn_of(s.obj).W
            elif s.kind == "rule":                # "scaled_encoders"
                u = self.ens[id(_obj_and_slice(s.obj.connection.post)[0])].E
            else:
                raise ValueError(s.kind)
            if s.idx is not None and s.kind in ("node", "neurons", "ensemble"):
                u = u[s.idx]
            if s.a is not None:
                s.state = (1.0 - s.a) * u if s.state is None else s.a * s.state + (1.0 - s.a) * u
                u = s.state
            if self.n_steps % s.every == 0:
                s.rows.append(np.array(u, dtype=float))

    def _conn_of(self, c):
        for s in self.conns:
            if s.c is c:
                return s
        raise KeyError(c)

    def run_steps(self, n):
        for _ in range(int(n)):
            self.step()

    def probe_data(self, probe):
        for s in self.probes:
            if s.p is probe:
                return np.array(s.rows)
        raise KeyError(probe)

    def weights(self, conn):
        return self._conn_of(conn).W

    def scaled_encoders(self, ens):
        return self.ens[id(ens)].E


def es_bias(es):
    return np.asarray(es.built.bias, dtype=float)
