"""ctypes binding of libmoira_pb.so (C ABI: include/moira_pb.h).

The library is the product; there is no Python or CPU fallback.  Importing this module
fails loudly when the shared library cannot be found or built, and creating a context
fails loudly (NoDeviceError) when no MI355X is visible.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmoira_pb.so")

MAX_LEN = 65535     # MPB_MAX_LEN (csrc/mpb_internal.h): longest read; a read may NEED up to 16384 DP rows (1024: one wave; more: k_wide, 16 waves)
OK, E_INVALID, E_NODEVICE, E_HIP, E_NOMEM, E_RANGE = 0, -1, -2, -3, -4, -5
AMBIG = {"treat_as_errors": 0, "ignore": 1, "disallow": 2}
FLAG_ROUND, FLAG_FAST_FMA, FLAG_TEST_UNDERPREDICT, FLAG_DECISION_ONLY, FLAG_BATCHED_ONLY, FLAG_COUNT_CELLS = 1, 2, 4, 8, 16, 32
FLAG_NO_NARROW = 64


def FLAG_NARROW_ROWS(r):
    """Test / measurement hook (MPB_FLAG_NARROW_ROWS): force the natural-order narrow pass with r rows (2..4)."""
    return (int(r) & 15) << 8


K_PREPASS, K_SCAN, K_SCATTER, K_DP, K_OVERFLOW, K_LAMBDA, K_WIDE, K_NARROW, K_FALLBACK, K_SAMPLE = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9
KERNEL_NAMES = {K_PREPASS: "prepass", K_SCAN: "scan", K_SCATTER: "scatter", K_DP: "dp", K_OVERFLOW: "overflow", K_LAMBDA: "lambda",
                K_WIDE: "wide", K_NARROW: "narrow", K_FALLBACK: "fallback", K_SAMPLE: "sample"}


class MoiraPBError(RuntimeError):
    pass


class NoDeviceError(MoiraPBError):
    pass


class FilterParams(C.Structure):
    _fields_ = [("alpha", C.c_double), ("uncert", C.c_double), ("maxerrors", C.c_double),
                ("ambig_mode", C.c_int32), ("flags", C.c_uint32)]


class PathInfo(C.Structure):
    _fields_ = [("narrow_rows", C.c_int32), ("sampled", C.c_int32), ("n_fallback", C.c_int64),
                ("sample_hist", C.c_int32 * 16), ("narrow_split", C.c_int32), ("reserved_", C.c_int32)]


class FilterCounts(C.Structure):
    _fields_ = [("n_reads", C.c_int64), ("n_pass", C.c_int64), ("n_fail", C.c_int64),
                ("n_overflow", C.c_int64)]


# name -> (restype, argtypes): every symbol include/moira_pb.h declares
_VP, _I32P, _U8P, _DP = C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_double)
PROTOTYPES = {
    "mpb_version": (C.c_char_p, []),
    "mpb_last_error": (C.c_char_p, []),
    "mpb_device_count": (C.c_int, []),
    "mpb_create": (C.c_int, [C.c_int, C.POINTER(_VP)]),
    "mpb_destroy": (C.c_int, [_VP]),
    "mpb_host_lut": (C.c_int, [_VP, _VP]),
    "mpb_device_lut": (C.c_int, [_VP, _VP, _VP]),
    "mpb_stream": (C.c_int, [_VP, C.POINTER(_VP)]),
    "mpb_synchronize": (C.c_int, [_VP]),
    "mpb_malloc": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    "mpb_free": (C.c_int, [_VP, _VP]),
    "mpb_memcpy_h2d": (C.c_int, [_VP, _VP, _VP, C.c_int64]),
    "mpb_memcpy_d2h": (C.c_int, [_VP, _VP, _VP, C.c_int64]),
    "mpb_memset": (C.c_int, [_VP, _VP, C.c_int, C.c_int64]),
    "mpb_host_alloc": (C.c_int, [_VP, C.c_int64, C.POINTER(_VP)]),
    "mpb_host_free": (C.c_int, [_VP, _VP]),
    "mpb_pack_read": (C.c_int, [C.c_char_p, _VP, C.c_int32, _VP, C.c_int32]),
    "mpb_pack_read_ascii": (C.c_int, [C.c_char_p, C.c_char_p, C.c_int32, C.c_int32, _VP, C.c_int32]),
    "mpb_pack_batch_ascii": (C.c_int, [C.c_char_p, C.c_char_p, _VP, C.c_int64, C.c_int32, C.c_int32, C.c_int64, _VP, _VP]),
    "mpb_decode_ascii_device": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32, C.c_int32, _VP, _VP]),
    "mpb_encode_ascii_device": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, _VP, _VP]),
    "mpb_decode_classify_device": (C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32, C.c_int32,
                                             C.POINTER(FilterParams), _VP, _VP, _VP, _VP, _VP]),
    "mpb_filter_device_classified": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                               C.POINTER(FilterParams), _VP, _VP, _VP, C.POINTER(FilterCounts)]),
    "mpb_filter_device": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                    C.POINTER(FilterParams), _VP, _VP, _VP, C.POINTER(FilterCounts)]),
    "mpb_filter_host": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                  C.POINTER(FilterParams), _VP, _VP, _VP, C.POINTER(FilterCounts)]),
    "mpb_shard_bounds": (C.c_int, [C.c_int64, C.c_int32, C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]),
    "mpb_filter_host_multi": (C.c_int, [_VP, C.c_int32, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                        C.POINTER(FilterParams), _VP, _VP, _VP, C.POINTER(FilterCounts), C.c_int32]),
    "mpb_numa_cpulist_for_pci": (C.c_int, [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.c_char_p, C.c_int32]),
    "mpb_calculate_errors_PB": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int32, C.c_double,
                                          C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "mpb_calculate_errors_poisson": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int32, C.c_double,
                                               C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "mpb_pack_batch_coded": (C.c_int, [C.c_char_p, _VP, _VP, C.c_int64, C.c_int32, C.c_int64, _VP, _VP, _VP]),
    "mpb_filter_host_coded": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                        C.POINTER(FilterParams), _VP, _VP, _VP, _VP, C.POINTER(FilterCounts)]),
    "mpb_broker_serve": (C.c_int, [_VP, C.c_char_p, C.c_int32, C.c_int32]),
    "mpb_broker_attach": (C.c_int, [C.c_char_p, C.c_int32, C.POINTER(_VP)]),
    "mpb_broker_call": (C.c_int, [_VP, C.c_char_p, _VP, C.c_int32, C.c_double,
                                  C.POINTER(C.c_double), C.POINTER(C.c_int32)]),
    "mpb_broker_detach": (C.c_int, [_VP]),
    "mpb_broker_shutdown": (C.c_int, [C.c_char_p]),
    "mpb_broker_stats": (C.c_int, [C.c_char_p, C.POINTER(C.c_int64), C.POINTER(C.c_int64), C.POINTER(C.c_int64),
                                   C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
    "mpb_poisson_lambda_device": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32, _VP, _VP]),
    "mpb_poisson_finish_host": (C.c_int, [_VP, _VP, _VP, C.c_int32, C.c_int64, C.POINTER(FilterParams), _VP, _VP]),
    "mpb_filter_poisson_host": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int32,
                                          C.POINTER(FilterParams), _VP, _VP, _VP, C.POINTER(FilterCounts)]),
    "mpb_synth_fill_device": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                        C.c_int32, _VP, C.c_uint64, C.c_int64]),
    "mpb_synth_fill_device_profile": (C.c_int, [_VP, _VP, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                                C.c_int32, _VP, C.c_uint64, C.c_int64, C.c_int32]),
    "mpb_timing_enable": (C.c_int, [_VP, C.c_int]),
    "mpb_timing_reset": (C.c_int, [_VP]),
    "mpb_kernel_time": (C.c_int, [_VP, C.c_int, C.POINTER(C.c_double), C.POINTER(C.c_int64)]),
    "mpb_last_class_histogram": (C.c_int, [_VP, _VP, _VP, C.c_int32]),
    "mpb_last_read_budgets": (C.c_int, [_VP, _VP, C.c_int64]),
    "mpb_last_algorithmic_cells": (C.c_int, [_VP, C.POINTER(C.c_int64)]),
    "mpb_last_path": (C.c_int, [_VP, C.POINTER(PathInfo)]),
}

_lib = None


def load(build_if_missing=True):
    """Load (building in-tree with hipcc when stale and possible) and return the CDLL."""
    global _lib
    if _lib is not None:
        return _lib
    global LIB_PATH
    override = os.environ.get("MOIRA_PB_LIB")
    if override:
        # experiment builds (tools/experiments/variants.sh) live OUTSIDE the tree and are selected here; the in-tree library
        # and its stamp are never touched by an experiment
        if not os.path.exists(override):
            raise MoiraPBError("MOIRA_PB_LIB=%s does not exist" % override)
        LIB_PATH, build_if_missing = override, False
    if build_if_missing:
        from . import build as _build
        if _build.stale():
            if os.path.exists(_build.HIPCC):
                _build.build()
            elif not os.path.exists(LIB_PATH):
                raise MoiraPBError("libmoira_pb.so is not built and hipcc is not available at %s; "
                                   "there is no fallback implementation" % _build.HIPCC)
    if not os.path.exists(LIB_PATH):
        raise MoiraPBError("%s not found; run `python -m moira_amd.build`" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in PROTOTYPES.items():
        f = getattr(lib, name)          # AttributeError here = header/library mismatch: fail loudly
        f.restype = res
        f.argtypes = args
    _lib = lib
    return lib


def check(rc):
    if rc == OK:
        return
    msg = load().mpb_last_error().decode(errors="replace")
    if rc in (E_INVALID, E_RANGE):
        raise ValueError(msg)
    if rc == E_NODEVICE:
        raise NoDeviceError(msg)
    if rc == E_NOMEM:
        raise MemoryError(msg)
    raise MoiraPBError("HIP error: " + msg)
