"""Drop-in for moira's CPython-2 extension `bernoulli` (moira/bernoullimodule.c) that runs under PYTHON 2.7 AS WELL AS
Python 3: ctypes and the standard library only, no syntax that either version lacks.

moira.py is Python 2 (moira/moira.py:247-251 `import bernoulli`, :817 `bernoulli.calculate_errors_PB(contig, contig_quals,
args.alpha)`).  Put THIS directory first on PYTHONPATH of the unchanged script:

    PYTHONPATH=/path/to/repo/moira_amd/dropin/py2 python2 moira.py --forward_fastq reads.fastq --processors 16

Same signature, same exceptions as the extension (moira/bernoullimodule.c:74-108).  In the main process the module opens
a GPU context of its own (mpb_create); in a worker process of moira.py's multiprocessing.Pool (moira/moira.py:398-399) it
attaches to the one GPU-owning broker process instead (mpb_broker_attach / mpb_broker_call, include/moira_pb.h) -- the
broker itself is `python3 -m moira_amd.broker` (any Python 3 with this package; started by hand or by the first worker
that finds none), so the workers never touch the GPU.  MOIRA_PB_BROKER=1 / 0 forces / forbids the broker,
MOIRA_PB_DEVICE picks the GPU, MOIRA_PB_LIB the library (default: ../../libmoira_pb.so next to this package).

This image has no Python 2: the file is exercised under Python 3 (tests/test_gpu_broker.py) and parsed with lib2to3's
Python-2 grammar (tests/test_library_abi.py); it has never run under a real Python 2.7.
"""
import array as _array
import ctypes as _C
import os as _os
import subprocess as _subprocess
import sys as _sys
import time as _time

try:
    _INTS = (int, long)            # noqa: F821  (Python 2)
except NameError:
    _INTS = (int,)

_HERE = _os.path.dirname(_os.path.abspath(__file__))
_ROOT = _os.path.dirname(_os.path.dirname(_os.path.dirname(_HERE)))
_LIB_PATH = _os.environ.get("MOIRA_PB_LIB") or _os.path.join(_ROOT, "moira_amd", "libmoira_pb.so")
_E_INVALID, _E_NODEVICE, _E_HIP, _E_NOMEM, _E_RANGE = -1, -2, -3, -4, -5
_state = {"lib": None, "pid": None, "call": None}


def _lib():
    if _state["lib"] is None:
        lib = _C.CDLL(_LIB_PATH)
        lib.mpb_last_error.restype = _C.c_char_p
        lib.mpb_create.argtypes = [_C.c_int, _C.POINTER(_C.c_void_p)]
        lib.mpb_calculate_errors_PB.argtypes = [_C.c_void_p, _C.c_char_p, _C.c_void_p, _C.c_int32, _C.c_double,
                                                _C.POINTER(_C.c_double), _C.POINTER(_C.c_int32)]
        lib.mpb_broker_attach.argtypes = [_C.c_char_p, _C.c_int32, _C.POINTER(_C.c_void_p)]
        lib.mpb_broker_call.argtypes = [_C.c_void_p, _C.c_char_p, _C.c_void_p, _C.c_int32, _C.c_double,
                                        _C.POINTER(_C.c_double), _C.POINTER(_C.c_int32)]
        _state["lib"] = lib
    return _state["lib"]


def _raise(rc):
    msg = _lib().mpb_last_error()
    if not isinstance(msg, str):
        msg = msg.decode("ascii", "replace")
    if rc in (_E_INVALID, _E_RANGE):
        raise ValueError(msg)
    if rc == _E_NOMEM:
        raise MemoryError(msg)
    raise RuntimeError(msg)


def _in_pool_worker():
    try:
        import multiprocessing
        parent = getattr(multiprocessing, "parent_process", None)
        if parent is not None:                              # Python 3.8+
            return parent() is not None
        return multiprocessing.current_process().name != "MainProcess"      # Python 2.7
    except Exception:
        return False


def _broker_name(device):
    return _os.environ.get("MOIRA_PB_BROKER_NAME") or "u%d_d%d" % (_os.getuid(), device)


def _runtime_dir():
    """This user's own directory for the start lock and the broker's log (as moira_amd/broker.py: runtime_dir)."""
    import stat
    d = _os.environ.get("XDG_RUNTIME_DIR")
    if d and _os.path.isdir(d) and _os.stat(d).st_uid == _os.getuid():
        return d
    d = _os.path.join("/dev/shm" if _os.path.isdir("/dev/shm") else "/tmp", "moira_pb_%d" % _os.getuid())
    try:
        _os.mkdir(d, 0o700)
    except OSError:
        pass
    st = _os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != _os.getuid() or (st.st_mode & 0o077):
        raise RuntimeError("%s is not a private directory of this user" % d)
    return d


def _open_private(path, flags, mode):
    """never through a symbolic link, never readable by others"""
    return _os.fdopen(_os.open(path, flags | _os.O_CREAT | getattr(_os, "O_NOFOLLOW", 0), 0o600), mode)


def _attach(device):
    """-> broker client handle; starts `python3 -m moira_amd.broker` when none is serving (under a file lock)."""
    import fcntl
    lib, name = _lib(), _broker_name(device).encode("ascii")
    h = _C.c_void_p()
    if lib.mpb_broker_attach(name, 0, _C.byref(h)) == 0:
        return h
    lock = _open_private(_os.path.join(_runtime_dir(), "start_%s.lock" % name.decode("ascii")), _os.O_RDWR, "r+b")
    fcntl.flock(lock, fcntl.LOCK_EX)
    try:
        if lib.mpb_broker_attach(name, 0, _C.byref(h)) == 0:
            return h
        env = dict(_os.environ)
        env["PYTHONPATH"] = _ROOT
        env.pop("MOIRA_PB_BROKER", None)
        log = _open_private(_os.path.join(_runtime_dir(), "broker_%s.log" % name.decode("ascii")), _os.O_WRONLY | _os.O_APPEND, "ab")
        py3 = _os.environ.get("MOIRA_PB_PYTHON3") or ("python3" if _sys.version_info[0] < 3 else _sys.executable)
        proc = _subprocess.Popen([py3, "-m", "moira_amd.broker", "--device", str(device), "--name", name.decode("ascii")],
                                 cwd=_ROOT, env=env, stdin=open(_os.devnull, "rb"), stdout=log, stderr=log, close_fds=True,
                                 preexec_fn=_os.setsid)
        t0 = _time.time()
        while lib.mpb_broker_attach(name, 200, _C.byref(h)) != 0:
            if proc.poll() is not None:
                raise RuntimeError("the broker process exited with code %s before serving (no GPU?)" % proc.returncode)
            if _time.time() - t0 > 120:
                raise RuntimeError("the broker did not come up within 120 s")
        return h
    finally:
        fcntl.flock(lock, fcntl.LOCK_UN)
        lock.close()


def _choose():
    mode = _os.environ.get("MOIRA_PB_BROKER", "auto").lower()
    device = int(_os.environ.get("MOIRA_PB_DEVICE", "0"))
    lib = _lib()
    if mode in ("1", "on", "yes", "true") or (mode == "auto" and _in_pool_worker()):
        h = _attach(device)
        return lambda seq, q, n, alpha, ee, ns: lib.mpb_broker_call(h, seq, q, n, alpha, ee, ns)
    ctx = _C.c_void_p()
    rc = lib.mpb_create(device, _C.byref(ctx))
    if rc:
        _raise(rc)
    return lambda seq, q, n, alpha, ee, ns: lib.mpb_calculate_errors_PB(ctx, seq, q, n, alpha, ee, ns)


def calculate_errors_PB(contig, contig_quals, alpha):
    """This function returns the expected errors of a given sequence with a given confidence value
    using a sum of Bernoulli random variables."""
    if not isinstance(contig, (str, bytes)):                                   # "s" of PyArg_ParseTuple "sO!d"
        raise TypeError("argument 1 must be string, not %s" % type(contig).__name__)
    if not isinstance(contig_quals, list):                                     # "O!" with &PyList_Type
        raise TypeError("argument 2 must be list, not %s" % type(contig_quals).__name__)
    alpha = float(alpha)                                                       # "d"
    if alpha <= 0 or alpha >= 1:
        raise ValueError("Alpha must be between 0 and 1")
    n = len(contig_quals)
    if n != len(contig):
        raise ValueError("contig and contig_quals must have the same length")
    try:
        q = _array.array("i", contig_quals)                                    # one C loop; TypeError for a non-integer
    except TypeError:
        raise TypeError("an integer is required")
    except OverflowError:                                                      # beyond 32 bits: wraps as (int)PyInt_AsLong does
        for v in contig_quals:
            if not isinstance(v, _INTS):
                raise TypeError("an integer is required")
        q = _array.array("i", [((v + 0x80000000) & 0xFFFFFFFF) - 0x80000000 for v in contig_quals])
    seq = contig if isinstance(contig, bytes) else contig.encode("ascii")
    addr = q.buffer_info()[0]
    pid = _os.getpid()
    if _state["pid"] != pid:                                                   # first call in this process (a forked worker decides for itself)
        _state["call"], _state["pid"] = _choose(), pid
    ee, ns = _C.c_double(), _C.c_int32()
    rc = _state["call"](seq, addr, n, alpha, _C.byref(ee), _C.byref(ns))
    if rc:
        _raise(rc)
    return ee.value, ns.value


calculate_errors = calculate_errors_PB
