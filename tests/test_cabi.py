"""The C-ABI library loads and exports every symbol include/ssn.h declares (no compute without a GPU)."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from sspslam_amd import _lib
from sspslam_amd.simulator import pack_model
from sspslam_amd.builder import build

from helpers import small_pathint

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def lib():
    if not os.path.exists(_lib.LIB_PATH):
        import __graft_entry__ as g
        g.build()
    return _lib.load()


def test_header_symbols_are_exported_and_declared(lib):
    hdr = open(os.path.join(ROOT, "include", "ssn.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    names = set(re.findall(r"\b(ssn_[a-z_0-9]+)\s*\(", hdr))
    assert len(names) >= 18
    for n in names:
        assert hasattr(lib, n), f"{n} declared in ssn.h but not exported"
    assert names == set(_lib.EXPORTS), names ^ set(_lib.EXPORTS)
    assert b"gfx950" in lib.ssn_version()


def test_struct_layouts_match_header():
    assert C.sizeof(_lib.OpDesc) == 6 * 4 + 12 * 8 + 4 * 8 and _lib.OpDesc.i.offset == 24
    assert C.sizeof(_lib.BufferDesc) == 24 and C.sizeof(_lib.ProbeDesc) == 32 and C.sizeof(_lib.Range) == 16
    assert C.sizeof(_lib.Counters) == 120 and _lib.Counters.fused_populations.offset == 112 and _lib.Counters.block_slots.offset == 96 and _lib.Counters.block_tpb.offset == 64 and _lib.Counters.fft_transforms.offset == 80
    assert _lib.Counters.block_members.offset == 88
    assert _lib.ModelDesc.dt.offset == 16 and _lib.ModelDesc.buffers.offset == 56
    assert _lib.ModelDesc.pre_to_core.offset == 88 and _lib.ModelDesc.exchange.offset == 112 and C.sizeof(_lib.ModelDesc) == 128


def test_create_fails_loudly_without_gpu_or_with_bad_args(lib):
    h = C.c_void_p()
    assert lib.ssn_create(None, C.byref(h)) == -1 and b"null" in lib.ssn_last_error()
    m = build(small_pathint(ssp_dim=7, n=8).model)
    desc, keep, _ = pack_model(m, "f32")
    desc.abi_version = 99
    assert lib.ssn_create(C.byref(desc), C.byref(h)) == -1 and b"ABI" in lib.ssn_last_error()
    desc.abi_version = _lib.SSN_ABI_VERSION
    if lib.ssn_device_count() == 0:
        rc = lib.ssn_create(C.byref(desc), C.byref(h))
        assert rc == -2 and b"no CPU fallback" in lib.ssn_last_error()      # SSN_EHIP
        assert not h.value
        from sspslam_amd.simulator import Simulator
        import sspslam_amd.frontend as fe
        with pytest.raises(fe.BuildError, match="no CPU fallback"):
            Simulator(None, model=m)
    assert lib.ssn_run_steps(None, 1, 0) == -1
    assert lib.ssn_n_steps(None) == -1


def test_pack_model_encodes_ops():
    m = build(small_pathint(ssp_dim=7, n=8).model)
    desc, keep, sig_probes = pack_model(m, "f64", steps_per_graph=4)
    assert desc.dtype == _lib.SSN_F64 and desc.n_ops == len(m.ops) and desc.n_tables == 2
    assert desc.n_signals == m.sig_size and desc.steps_per_graph == 4
    kinds = [desc.ops[i].kind for i in range(desc.n_ops)]
    assert kinds.count(_lib.OP_CODE["ensarray"]) == 1
    j = kinds.index(_lib.OP_CODE["ensarray"])
    e = next(o for o in m.ops if o["kind"] == "ensarray")
    assert [desc.ops[j].i[a] for a in (1, 2, 3, 4)] == [e["K"], e["n"], e["din"], e["dout"]]
    assert abs(desc.ops[j].f[0] - 0.02) < 1e-15 and abs(desc.ops[j].f[1] - 0.002) < 1e-15
    assert len(sig_probes) == 1 and desc.probes[0].width == 7
