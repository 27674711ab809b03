"""ctypes binding of ``libssn_hip.so`` (C ABI in ``include/ssn.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C csrc``.  There is no CPU
fallback: if the shared object is missing or cannot be loaded, ``load()`` raises.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SSN_HIP_LIB") or os.path.join(_HERE, "libssn_hip.so")     # override: A/B builds of the library

SSN_ABI_VERSION = 8
SSN_F32, SSN_F64 = 0, 1
SSN_BUF_REAL, SSN_BUF_I32 = 0, 1
NEURON_CODE = {"lif": 0, "lifrate": 1, "relu": 2}
OP_CODE = {"fill": 1, "table": 2, "axpy": 3, "matvec": 4, "lowpass": 5, "ensarray": 6, "neurons": 7,
           "pes": 8, "voja": 9, "cleanup": 10, "gate": 11, "lincomb": 12}
PROBE_KINDS = {"v_pk_fma_f32": 0, "v_pk_mul_f32": 1, "v_pk_add_f32": 2, "trans": 3, "v_fma_f32": 4, "v_add_f32": 5, "dpp": 6,
               "v_mov_b32": 7, "lane": 8, "v_cndmask_b32": 9, "other": 10}
STATUS = {0: "SSN_OK", -1: "SSN_EINVAL", -2: "SSN_EHIP", -3: "SSN_ERCCL", -4: "SSN_ENOMEM", -5: "SSN_EUNSUPPORTED"}


class BufferDesc(C.Structure):
    _fields_ = [("data", C.c_void_p), ("count", C.c_int64), ("kind", C.c_int32), ("reserved", C.c_int32)]


class OpDesc(C.Structure):
    _fields_ = [("kind", C.c_int32), ("level", C.c_int32), ("stage", C.c_int32), ("border", C.c_int32),
                ("src_prev", C.c_int32), ("phase", C.c_int32), ("i", C.c_int64 * 12), ("f", C.c_double * 4)]


class ProbeDesc(C.Structure):
    _fields_ = [("src", C.c_int64), ("width", C.c_int64), ("every", C.c_int64), ("stage", C.c_int64)]


class Range(C.Structure):
    _fields_ = [("lo", C.c_int64), ("hi", C.c_int64)]


class ModelDesc(C.Structure):
    _fields_ = [("abi_version", C.c_int32), ("dtype", C.c_int32), ("device", C.c_int32), ("n_tables", C.c_int32),
                ("dt", C.c_double), ("n_signals", C.c_int64), ("signal_init", C.POINTER(C.c_double)),
                ("n_buffers", C.c_int32), ("n_ops", C.c_int32), ("n_probes", C.c_int32),
                ("steps_per_graph", C.c_int32), ("buffers", C.POINTER(BufferDesc)), ("ops", C.POINTER(OpDesc)),
                ("probes", C.POINTER(ProbeDesc)), ("n_pre_to_core", C.c_int32), ("n_core_to_post", C.c_int32),
                ("pre_to_core", C.POINTER(Range)), ("core_to_post", C.POINTER(Range)),
                ("n_exchange", C.c_int32), ("reserved", C.c_int32), ("exchange", C.POINTER(Range)),
                ("block_steps", C.c_int32), ("flags", C.c_int32)]


class Counters(C.Structure):
    _fields_ = [("n_steps", C.c_int64), ("launches_per_step", C.c_int64), ("dominant_launches", C.c_int64),
                ("dominant_ms_total", C.c_double), ("dominant_bytes_per_launch", C.c_double),
                ("dominant_units_per_launch", C.c_int64), ("last_run_ms", C.c_double), ("device_bytes", C.c_int64),
                ("block_tpb", C.c_int32), ("block_npt", C.c_int32), ("block_enc_lds", C.c_int32),
                ("block_threads", C.c_int32), ("fft_transforms", C.c_int32), ("fft_bluestein", C.c_int32),
                ("block_members", C.c_int32), ("batch_products_skipped", C.c_int32),
                ("block_slots", C.c_int64), ("block_slots_silent", C.c_int64),
                ("fused_populations", C.c_int32), ("serial_chains", C.c_int32)]


class KernelTime(C.Structure):
    _fields_ = [("name", C.c_char * 32), ("launches", C.c_int64), ("ms_total", C.c_double)]


EXPORTS = {
    "ssn_create": (C.c_int, [C.POINTER(ModelDesc), C.POINTER(C.c_void_p)]),
    "ssn_destroy": (None, [C.c_void_p]),
    "ssn_reset": (C.c_int, [C.c_void_p]),
    "ssn_set_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]),
    "ssn_set_table_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]),
    "ssn_stage_table": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64, C.c_void_p, C.c_int64, C.c_int64]),
    "ssn_commit_tables": (C.c_int, [C.c_void_p]),
    "ssn_reserve_probes": (C.c_int, [C.c_void_p, C.c_int64]),
    "ssn_run_steps": (C.c_int, [C.c_void_p, C.c_int64, C.c_int32]),
    "ssn_read_probe": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64]),
    "ssn_read_probe_device": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64, C.c_int64]),
    "ssn_probe_count": (C.c_int64, [C.c_void_p, C.c_int32]),
    "ssn_read_signal": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "ssn_write_signal": (C.c_int, [C.c_void_p, C.c_int64, C.c_int64, C.c_void_p]),
    "ssn_read_buffer": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "ssn_write_buffer": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int64]),
    "ssn_run_phase": (C.c_int, [C.c_void_p, C.c_int32]),
    "ssn_exchange_size": (C.c_int64, [C.c_void_p]),
    "ssn_cycle_steps": (C.c_int64, [C.c_void_p]),
    "ssn_exchange_pack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssn_exchange_unpack": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssn_phase_async": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p]),
    "ssn_phase_sync": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ssn_probe_issue_rate": (C.c_int, [C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ssn_get_counters": (C.c_int, [C.c_void_p, C.POINTER(Counters)]),
    "ssn_get_kernel_times": (C.c_int, [C.c_void_p, C.POINTER(KernelTime), C.c_int32]),
    "ssn_n_steps": (C.c_int64, [C.c_void_p]),
    "ssn_device_count": (C.c_int, []),
    "ssn_last_error": (C.c_char_p, []),
    "ssn_version": (C.c_char_p, []),
}

_lib = None


def load():
    """Load libssn_hip.so and declare every entry point of include/ssn.h.  Raises if unavailable."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                          "or `make -C semantic-spiking-neural-slam-2023_amd/csrc` (there is no CPU fallback)")
    # PyTorch-ROCm ships its own libamdhip64 / HSA runtime.  When the system runtime this library links against is
    # mapped first and torch (used by the decoder solver) initialises the GPU afterwards, later HIP calls from here
    # fail with "no ROCm-capable device is detected"; the other order works.  So: torch first, when it is installed.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in EXPORTS.items():
        fn = getattr(lib, name)          # AttributeError if a declared symbol is not exported
        fn.restype, fn.argtypes = res, args
    _lib = lib
    return lib


def last_error():
    return load().ssn_last_error().decode("utf-8", "replace")
