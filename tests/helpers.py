"""Test helpers: an oracle-backed stand-in with the Simulator's Python surface (tests only)."""
import numpy as np

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
