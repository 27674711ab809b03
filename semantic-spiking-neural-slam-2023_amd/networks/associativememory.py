"""Key -> value associative memory learned online with Voja (encoders) + PES (decoders).

Mirrors the reference's ``AssociativeMemory`` (``sspslam/networks/associativememory.py:11-54``):
``key_input -> memory`` (Voja on the encoders, ``:30-33``), ``memory -> recall`` with PES-learned
decoders initialised to the zero function (``:38-43``), an ``error`` ensemble computing
``recall - value`` through ``Lowpass(tau)`` that drives PES (``:52-54``), and a ``learning`` node
(0 = learn, positive = off) that inhibits ``error`` directly on its neurons (``:47-49``) and gates Voja.
"""
import numpy as np

from .. import frontend as nengo


class AssociativeMemory(nengo.Network):
    def __init__(self, n_neurons, d_key, d_value, intercept, voja_learning_rate=5e-2,
                 pes_learning_rate=1e-3, encoders=None, radius=1, voja=True, tau=0.05, **kwargs):
        super().__init__(**kwargs)
        with self:
            self.key_input = nengo.Node(size_in=d_key, label="memory_input")
            self.value_input = nengo.Node(size_in=d_value)
            self.learning = nengo.Node(size_in=1)
            self.recall = nengo.Ensemble(n_neurons, d_value, label="memory_recall")

            mem_kw = dict(intercepts=[intercept] * n_neurons, radius=radius, label="memory")
            if encoders is not None:
                mem_kw["encoders"] = encoders
            self.memory = nengo.Ensemble(n_neurons, d_key, **mem_kw)

            if voja:
                rule = nengo.Voja(learning_rate=voja_learning_rate, post_synapse=None)
                self.conn_in = nengo.Connection(self.key_input, self.memory, synapse=None,
                                                learning_rule_type=rule, label="map_conn_in")
                nengo.Connection(self.learning, self.conn_in.learning_rule, synapse=None)
            else:
                self.conn_in = nengo.Connection(self.key_input, self.memory, synapse=None, label="map_conn_in")

            def null_value(x, _n=d_value):
                return np.zeros(_n)

            self.conn_out = nengo.Connection(self.memory, self.recall,
                                             learning_rule_type=nengo.PES(pes_learning_rate),
                                             function=null_value, label="map_conn_pes")

            self.error = nengo.Ensemble(n_neurons, d_value, label="memory_pes_error")
            nengo.Connection(self.learning, self.error.neurons, transform=-2.5 * np.ones((n_neurons, 1)),
                             synapse=None)
            nengo.Connection(self.value_input, self.error, transform=-1, synapse=tau)
            nengo.Connection(self.recall, self.error, synapse=tau)
            nengo.Connection(self.error, self.conn_out.learning_rule, synapse=tau)
