"""Neural circular convolution (bind / unbind) and element-wise product networks.

Mirrors the reference's ``sspslam/networks/binding.py``: ``circconv`` (``:12-20``, the NumPy
known-answer oracle), ``transform_in`` (``:23-54``), ``transform_out`` (``:57-74``), ``dft_half``
(``:86-89``), ``CircularConvolution`` (``:92-218``) and ``Product`` (``:233-324``).

Layout of the Fourier-domain vector (length ``D2 = 4*(d//2+1)``): four slots per half-spectrum bin
w so that the element-wise product of ``tr_a @ a`` and ``tr_b @ b`` yields the four real products
``[Re.Re, Im.Im, Re.Im, Im.Re]`` of the complex multiply; ``transform_out`` recombines them
(``Re = s0 - s1``, ``Im = s2 + s3``) and applies the inverse real DFT.  The purely-imaginary slots
of the DC (and Nyquist) bins are identically zero and are kept, like the reference
(its ``remove_imag_rows`` is a no-op).
"""
import numpy as np

from .. import frontend as nengo


def circconv(a, b, invert_a=False, invert_b=False, axis=-1):
    """FFT-domain circular convolution (correlation when one side is inverted)."""
    A = np.fft.fft(a, axis=axis)
    B = np.fft.fft(b, axis=axis)
    return np.fft.ifft((A.conj() if invert_a else A) * (B.conj() if invert_b else B), axis=axis).real


def dft_half(n):
    """(n//2+1, n) complex DFT rows ``exp(-2 pi i w x / n)`` of the non-redundant half spectrum."""
    wx = np.outer(np.arange(n // 2 + 1), np.arange(n)) % n
    return np.exp(-2j * np.pi * wx / n)


def transform_in(dims, align, invert):
    """(D2, dims) real matrix mapping a vector to the 4-slot Fourier layout for operand 'A' or 'B'."""
    if align not in ("A", "B"):
        raise nengo.ValidationError("'align' must be either 'A' or 'B'", "align")
    rows = dft_half(dims)
    if invert:
        rows = rows.conj()
    re, im = rows.real, rows.imag
    slots = (re, im, re, im) if align == "A" else (re, im, im, re)
    return np.stack(slots, axis=1).reshape(-1, dims)


def transform_out(dims):
    """(dims, D2) real matrix: complex recombination + inverse real DFT (with the 1/dims factor)."""
    rows = dft_half(dims).conj()
    w = np.arange(dims // 2 + 1)
    scale = np.where((w == 0) | (2 * w == dims), 1.0, 2.0)[:, None] / dims
    re, im = rows.real * scale, rows.imag * scale
    return np.stack((re, -re, -im, -im), axis=1).reshape(-1, dims).T


def dot_product_transform(dimensions, scale=1.0):
    return scale * np.ones((1, dimensions))


class Product(nengo.Network):
    """Element-wise product via two squaring ensemble arrays: ``ab = ((a+b)^2 - (a-b)^2)/4``."""

    def __init__(self, n_neurons, dimensions, input_magnitude=1.0, dot_product=False,
                 label="product", solver=nengo.Default, **kwargs):
        super().__init__(label=label, **kwargs)
        half = max(1, n_neurons // 2)
        radius = input_magnitude * np.sqrt(2)
        g = 1.0 / np.sqrt(2.0)
        with self:
            self.input_a = nengo.Node(size_in=dimensions, label=label + "_input_a")
            self.input_b = nengo.Node(size_in=dimensions, label=label + "_input_b")
            self.output = nengo.Node(size_in=dimensions, label=label + "_output")
            self.sq1 = nengo.EnsembleArray(half, n_ensembles=dimensions, ens_dimensions=1,
                                           radius=radius, label=label + "_sq1")
            self.sq2 = nengo.EnsembleArray(half, n_ensembles=dimensions, ens_dimensions=1,
                                           radius=radius, label=label + "_sq2")
            for src, dst, sign in ((self.input_a, self.sq1, 1), (self.input_b, self.sq1, 1),
                                   (self.input_a, self.sq2, 1), (self.input_b, self.sq2, -1)):
                nengo.Connection(src, dst.input, transform=sign * g, synapse=None)
            kw = {} if solver is nengo.Default else {"solver": solver}
            sq1_out = self.sq1.add_output("square", np.square, **kw)
            sq2_out = self.sq2.add_output("square", np.square, **kw)
            if dot_product:
                nengo.Connection(sq1_out, self.output, transform=dot_product_transform(dimensions, 0.5),
                                 synapse=None)
                nengo.Connection(sq2_out, self.output, transform=dot_product_transform(dimensions, -0.5),
                                 synapse=None)
            else:
                nengo.Connection(sq1_out, self.output, transform=0.5, synapse=None)
                nengo.Connection(sq2_out, self.output, transform=-0.5, synapse=None)


class CircularConvolution(nengo.Network):
    """``output = input_a (*) input_b`` (optionally with either operand involuted)."""

    def __init__(self, n_neurons, dimensions, invert_a=False, invert_b=False, input_magnitude=1.0,
                 label="circonv", solver=nengo.Default, **kwargs):
        if "net" in kwargs:
            raise nengo.ObsoleteError("The 'net' argument is no longer supported.")
        super().__init__(label=label, **kwargs)
        self.transform_a = transform_in(dimensions, "A", invert_a)
        self.transform_b = transform_in(dimensions, "B", invert_b)
        self.transform_out = transform_out(dimensions)
        with self:
            self.input_a = nengo.Node(size_in=dimensions, label=label + "_input_a")
            self.input_b = nengo.Node(size_in=dimensions, label=label + "_input_b")
            self.product = Product(n_neurons, self.transform_out.shape[1],
                                   input_magnitude=input_magnitude * 2, label=label + "_product",
                                   solver=solver)
            self.output = nengo.Node(size_in=dimensions, label=label + "_output")
            nengo.Connection(self.input_a, self.product.input_a, transform=self.transform_a, synapse=None)
            nengo.Connection(self.input_b, self.product.input_b, transform=self.transform_b, synapse=None)
            nengo.Connection(self.product.output, self.output, transform=self.transform_out, synapse=None)
