"""Test helpers: an oracle-backed stand-in with the Simulator's Python surface (tests only)."""
import numpy as np

import sspslam_amd.frontend as nengo
from oracle import OracleSimulator


class OracleBackedSimulator:
    """Same methods as sspslam_amd.simulator.Simulator, stepping on the NumPy oracle."""

    def __init__(self, model, dtype=np.float64):
        self.model = model
        self.o = OracleSimulator(model, dtype=dtype)
        self._index = {p["probe"]: i for i, p in enumerate(model.probes)}
        self.n_steps = 0

    class _Data:
        def __init__(self, outer):
            self.outer = outer

        def __getitem__(self, key):
            if key in self.outer._index:
                return self.outer.o.probe_data(self.outer._index[key])
            return self.outer.model.params[key]

    @property
    def data(self):
        return OracleBackedSimulator._Data(self)

    def prepare(self, n):
        pass

    def set_exchange(self, allreduce):
        """Neuron-sharded models: ``allreduce(vector)`` sums a float64 vector over the ranks in place."""
        def hook(sig, ranges):
            vec = np.concatenate([sig[lo:hi] for lo, hi in ranges]) if ranges else np.zeros(0)
            allreduce(vec)
            off = 0
            for lo, hi in ranges:
                sig[lo:hi] = vec[off:off + hi - lo]
                off += hi - lo
        self.o.exchange_hook = hook

    def read_buffer(self, buffer_id):
        return np.array(self.o.buf[buffer_id], dtype=np.float64)

    def run_steps(self, n, collect=True, profile=False):
        self.o.run_steps(n)
        self.n_steps += n

    def probe_tail(self, key, n):
        return self.o.probe_data(self._index[key])[-n:]

    def clear_probe_data(self):
        pass

    def close(self):
        pass


def small_pathint(ssp_dim=7, n=64, T=2.0, seed=1, limit=0.5, **kw):
    from sspslam_amd import harness as H
    space = H.make_ssp_space(2, ssp_dim=ssp_dim)
    path, vels = H.make_random_path(T, limit=limit, seed=seed)
    return H.make_pathint_model(space, path, vels, n, **kw)


def random_network(seed, big=False, shardable=False, learned_probes=False):
    """(``big``: populations of 1500 - 9000 neurons, arrays of up to 20 x 2600, 64 - 128-point convolutions - the sizes at which the
    device leaves the glue micro-operators for its big kernels.)
    Nodes, ensembles of every neuron type, an ensemble array, pass-through nodes; decoded connections with functions and
    transforms, slices, neuron-to-node read-outs, recurrent and feedback connections through synapses, direct (synapse=None)
    connections only forwards in declaration order (no algebraic loops), an optional PES rule; probes with and without
    synapses on everything."""
    rng = np.random.RandomState(seed)
    rng2 = np.random.RandomState(seed + 7919)       # later additions draw from here: the networks of old seeds keep their shape
    probes = []
    with nengo.Network(seed=seed) as net:
        n_in = rng.randint(1, 3)
        sources = []                                     # (object, dimensions, may be the pre of a direct connection)
        for k in range(n_in):
            d = int(rng.choice([1, 2, 3, 5, 17, 24]))
            w = rng.uniform(2.0, 12.0, size=d)
            ph = rng.uniform(0, 6.28, size=d)
            amp = rng.uniform(0.3, 0.9)
            sources.append((nengo.Node(lambda t, w=w, ph=ph, amp=amp: amp * np.sin(w * t + ph)), d))
        if rng.rand() < 0.4:                          # a constant node, and a piecewise-constant one (few distinct table rows)
            dcn = int(rng.choice([1, 3, 7]))
            sources.append((nengo.Node(rng.uniform(-0.6, 0.6, size=dcn)), dcn))
        if rng.rand() < 0.4:
            levels = rng.uniform(-0.8, 0.8, size=(5, 2))
            sources.append((nengo.Node(lambda t, L=levels: L[int(t * 40) % 5]), 2))
        objs = list(sources)
        n_ens = rng.randint(2, 5)
        ens = []
        for k in range(n_ens):
            d = int(rng.choice([1, 2, 3, 4, 9, 17, 20])) if not big else int(rng.choice([3, 20, 33, 64]))
            n = int(rng.choice([30, 64, 100, 257, 300, 700, 1100])) if not big else int(rng.choice([1500, 4500, 9000]))
            nt = [nengo.LIF(), nengo.LIF(), nengo.LIF(tau_rc=0.03, tau_ref=0.001), nengo.LIFRate(), nengo.RectifiedLinear()][rng.randint(0, 5)]
            if rng2.rand() < 0.35:                   # other parameters of the same types: amplitude, a voltage floor below zero, time constants
                amp_ = float(rng2.choice([0.5, 2.0]))
                if type(nt) is nengo.LIF:
                    nt = [nengo.LIF(amplitude=amp_), nengo.LIF(min_voltage=-1.0), nengo.LIF(tau_rc=0.05, tau_ref=0.0015, amplitude=amp_)][rng2.randint(0, 3)]
                elif type(nt) is nengo.LIFRate:
                    nt = [nengo.LIFRate(amplitude=amp_), nengo.LIFRate(tau_rc=0.04, tau_ref=0.001)][rng2.randint(0, 2)]
                else:
                    nt = nengo.RectifiedLinear(amplitude=amp_)
            kw = {}
            if rng.rand() < 0.3:
                kw["intercepts"] = nengo.Uniform(-0.5, 0.9)
                kw["max_rates"] = nengo.Uniform(100, 300)
            if rng.rand() < 0.2:
                enc = np.random.RandomState(seed * 131 + k).randn(n, d)
                kw["encoders"] = enc / np.linalg.norm(enc, axis=1, keepdims=True)
            given_gb = rng2.rand() < 0.15
            if given_gb and "intercepts" not in kw:      # gain and bias given instead of solved from intercepts / maximum rates
                g2 = np.random.RandomState(seed * 977 + k)
                kw["gain"], kw["bias"] = g2.uniform(0.5, 2.5, size=n), g2.uniform(-1.0, 1.5, size=n)
            e = nengo.Ensemble(n, d, neuron_type=nt, radius=float(rng.choice([1.0, 1.5])), **kw)
            ens.append((e, d))
            objs.append((e, d))
        if rng.rand() < 0.5:
            K, dk = int(rng.choice([3, 8, 21])), int(rng.choice([1, 2]))
            if big:
                K, dk = int(rng.choice([6, 20])), int(rng.choice([1, 2, 3]))
            ea_kw = {}
            if rng2.rand() < 0.3:                       # arrays of other neuron types / radii (the generic array body instead of the LIF fast path)
                ea_kw["neuron_type"] = [nengo.LIFRate(), nengo.RectifiedLinear(), nengo.LIF(tau_rc=0.03, tau_ref=0.001), nengo.LIF(min_voltage=-0.5)][rng2.randint(0, 4)]
            if rng2.rand() < 0.3:
                ea_kw["radius"] = 1.5
            ea = nengo.EnsembleArray(int(rng.choice([40, 90])) if not big else int(rng.choice([1200, 2600])), K, ens_dimensions=dk, **ea_kw)
            objs.append((ea.input, K * dk))
            arr_out = (ea.output, K * dk)
        else:
            arr_out = None
        n_pass = rng.randint(0, 3)
        passes = []
        for k in range(n_pass):
            d = int(rng.choice([1, 2, 3, 6]))
            pnode = nengo.Node(size_in=d)
            passes.append((pnode, d))
            objs.append((pnode, d))
        order = {id(o): i for i, (o, _) in enumerate(objs)}

        def transform(d_out, d_in):
            if d_out == d_in and rng.rand() < 0.4:
                return float(rng.uniform(0.3, 1.2))
            return rng.uniform(-1.0, 1.0, size=(d_out, d_in)) / np.sqrt(d_in)

        def connect(pre, d_pre, post, d_post, synapse, allow_function):
            kw = {}
            if allow_function and rng.rand() < 0.4:
                f_d = int(rng.choice([1, 2]))
                if f_d == 1:
                    kw["function"] = lambda x: x[0] ** 2
                else:
                    kw["function"] = lambda x: [x[0] * x[-1], x[0]]
                d_pre = f_d
            kw["transform"] = transform(d_post, d_pre)
            return nengo.Connection(pre, post, synapse=synapse, **kw)

        # every ensemble gets at least one input, from something declared before it (directly or through a synapse)
        targets = [(e, d) for e, d in ens] + ([(ea.input, arr_out[1])] if arr_out else []) + passes
        producers = list(sources)
        learned = voja_conn = None
        for post, d_post in targets:
            pre, d_pre = producers[rng.randint(0, len(producers))]
            is_ens = isinstance(pre, nengo.Ensemble)
            syn = [None, 0.005, 0.02][rng.randint(0, 3)]
            if is_ens and learned is None and isinstance(post, nengo.Node) and syn is not None and rng.rand() < 0.7:
                # a PES rule on this decoded connection: error = post - (a transform of the first input), as in the reference's
                # associative memory (associativememory.py:41-54)
                learned = nengo.Connection(pre, post, synapse=syn, transform=transform(d_post, d_pre),
                                           learning_rule_type=nengo.PES(learning_rate=float(rng.choice([1e-4, 5e-4]))))
                err = nengo.Node(size_in=d_post)
                nengo.Connection(post, err, synapse=None)
                nengo.Connection(sources[0][0], err, transform=-rng.uniform(-1.0, 1.0, size=(d_post, sources[0][1])), synapse=None)
                nengo.Connection(err, learned.learning_rule, synapse=None)
            else:
                connect(pre, d_pre, post, d_post, syn, is_ens)
            producers.append((post, d_post))
        if arr_out:
            producers.append(arr_out)
        # extra connections: backwards or recurrent ones only through a synapse
        for k in range(rng.randint(1, 5)):
            pre, d_pre = producers[rng.randint(0, len(producers))]
            post, d_post = targets[rng.randint(0, len(targets))]
            if pre is post:
                syn = 0.05
            else:
                syn = [0.005, 0.01, 0.05][rng.randint(0, 3)]
            connect(pre, d_pre, post, d_post, syn, isinstance(pre, nengo.Ensemble))
        # a circular convolution of two of the producers (reference networks/binding.py:297-317; the transforms are FFTs on the device)
        if rng.rand() < 0.45:
            from sspslam_amd.networks import CircularConvolution
            dc = int(rng.choice([9, 16, 25, 36])) if not big else int(rng.choice([64, 97, 128]))
            cc = CircularConvolution(int(rng.choice([20, 50])), dc, invert_b=bool(rng.rand() < 0.5))
            for inp in (cc.input_a, cc.input_b):
                pre, d_pre = producers[rng.randint(0, len(producers))]
                nengo.Connection(pre, inp, transform=rng.uniform(-1.0, 1.0, size=(dc, d_pre)) / np.sqrt(d_pre),
                                 synapse=[0.005, 0.02][rng.randint(0, 2)])
            probes.append(nengo.Probe(cc.output, synapse=[None, 0.01][rng.randint(0, 2)]))
            if rng.rand() < 0.5:
                post, d_post = targets[rng.randint(0, len(targets))]
                nengo.Connection(cc.output, post, transform=rng.uniform(-1.0, 1.0, size=(d_post, dc)) / np.sqrt(dc), synapse=0.01)
        # slices on both ends, a neuron-to-neuron weight matrix, a Voja rule on a decoded connection into an ensemble
        if rng.rand() < 0.5 and len(ens) >= 2:
            (ea_, da), (eb_, db) = ens[0], ens[-1]
            k = min(da, db)
            lo_a, lo_b = int(rng.randint(0, da - k + 1)), int(rng.randint(0, db - k + 1))
            pre = ea_[lo_a:lo_a + k] if k > 1 else ea_[lo_a]
            post = eb_[lo_b:lo_b + k] if k > 1 else eb_[lo_b]
            nengo.Connection(pre, post, synapse=0.01, transform=float(rng.uniform(0.3, 1.0)))
        if rng.rand() < 0.35 and len(ens) >= 2:
            (ea_, da), (eb_, db) = ens[rng.randint(0, len(ens))], ens[rng.randint(0, len(ens))]
            if ea_ is not eb_ and not shardable:       # (neuron slices and neuron probes are refused for neuron-sharded builds)
                na, nb = min(ea_.n_neurons, 64), min(eb_.n_neurons, 48)
                nengo.Connection(ea_.neurons[:na], eb_.neurons[:nb], synapse=0.005,
                                 transform=rng.uniform(-1e-3, 1e-3, size=(nb, na)))
        if rng.rand() < 0.35:
            post, d_post = ens[rng.randint(0, len(ens))]
            if isinstance(post.neuron_type, nengo.LIF):
                # (the builder takes Voja on a plain Node -> Ensemble connection, as the reference's memory has it: associativememory.py:31)
                wv, phv = rng.uniform(2.0, 9.0, size=d_post), rng.uniform(0, 6.28, size=d_post)
                vsrc = nengo.Node(lambda t, w=wv, ph=phv: 0.6 * np.sin(w * t + ph))
                vc = voja_conn = nengo.Connection(vsrc, post, synapse=None,
                                      learning_rule_type=nengo.Voja(learning_rate=float(rng.choice([1e-3, 5e-3])), post_synapse=None))
                if rng.rand() < 0.5:
                    lsig = nengo.Node(lambda t: -1.0 if (t % 0.05) < 0.02 else 0.0)
                    nengo.Connection(lsig, vc.learning_rule, synapse=None)
        # read-outs
        for e, d in ens:
            r = rng.rand()
            if r < 0.5:
                probes.append(nengo.Probe(e, synapse=[None, 0.01, 0.03][rng.randint(0, 3)], sample_every=[None, None, 0.003][rng.randint(0, 3)]))
            if rng.rand() < 0.4 and not shardable:
                probes.append(nengo.Probe(e.neurons[:min(7, e.n_neurons)]))
            if rng.rand() < 0.4:
                o = nengo.Node(size_in=1)
                nengo.Connection(e, o, synapse=0.01, function=lambda x: x[0] ** 2)
                probes.append(nengo.Probe(o, synapse=[None, 0.02][rng.randint(0, 2)]))
        for pnode, d in passes:
            probes.append(nengo.Probe(pnode, synapse=[None, 0.01][rng.randint(0, 2)]))
        if arr_out:
            probes.append(nengo.Probe(arr_out[0], synapse=0.01))
        if not probes:
            probes.append(nengo.Probe(ens[0][0], synapse=0.01))
        if arr_out and rng2.rand() < 0.4:       # a second, function-valued output of the array (EnsembleArray.add_output, reference binding.py:304-306)
            sq = ea.add_output("square", lambda x: x[0] ** 2)
            probes.append(nengo.Probe(sq, synapse=0.01))
        if learned_probes:      # (after everything random, so that the network of a seed is the same with and without them)
            if learned is not None:
                probes.append(nengo.Probe(learned, "weights", sample_every=0.01))
            if voja_conn is not None:
                probes.append(nengo.Probe(voja_conn.learning_rule, "scaled_encoders", sample_every=0.02))
    return net, probes
