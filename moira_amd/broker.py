"""One GPU-owning broker process for per-read callers in many worker processes.

Reference shape served: moira/moira.py:398-399,431-454 -- `Pool(args.processors)` workers, each calling
`bernoulli.calculate_errors_PB(contig, contig_quals, alpha)` once per read (moira/moira.py:817).  With a GPU context per
worker those one-read launches time-share the card (2.4e4 calls/s in all whatever P is); with the broker the workers
never touch the GPU: a call packs its read into a shared-memory slot and waits, the broker micro-batches whatever is
pending into one launch (include/moira_pb.h: mpb_broker_*; moira_amd/csrc/mpb_broker.cpp).

    python -m moira_amd.broker [--device 0] [--name NAME] [--slots 64] [--idle-exit 10]     # the broker itself

A worker does not start it by hand: `client()` attaches to the broker of (user, device) and, when there is none, starts
one as a FRESH CHILD PROCESS (a new interpreter, nothing inherited) before this process has touched the GPU -- under a
file lock, so that the P workers of a pool start exactly one.  The broker leaves by itself `idle_exit` seconds after
its last attached process has gone.
"""
import array
import ctypes as C
import fcntl
import os
import stat
import subprocess
import sys
import tempfile
import time

import numpy as np

from . import _lib as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def default_name(device=0):
    return os.environ.get("MOIRA_PB_BROKER_NAME") or "u%d_d%d" % (os.getuid(), int(device))


def marshal_read(contig, contig_quals, alpha):
    """Argument rules of bernoulli.calculate_errors_PB (moira/bernoullimodule.c:66-108: "sO!d", alpha in (0, 1), equal
    lengths, PyInt_AsLong per element) -> (bytes, int32 buffer, its address, float).  Shared by the direct and the broker
    entry.  The list goes through array.array('i', ...): one C loop, half the time of numpy's list conversion, and a
    TypeError for anything that is not an integer -- as PyInt_AsLong would raise."""
    if not isinstance(contig, str):
        raise TypeError("argument 1 must be str, not %s" % type(contig).__name__)
    if not isinstance(contig_quals, list):
        raise TypeError("argument 2 must be list, not %s" % type(contig_quals).__name__)
    alpha = float(alpha)                                   # "d" format: TypeError if not a number
    if alpha <= 0 or alpha >= 1:
        raise ValueError("Alpha must be between 0 and 1")
    if len(contig_quals) != len(contig):
        raise ValueError("contig and contig_quals must have the same length")
    try:
        qi = array.array("i", contig_quals)
        return contig.encode(), qi, qi.buffer_info()[0], alpha
    except TypeError:
        raise TypeError("an integer is required")
    except OverflowError:                                  # a Python int beyond 32 bits: wraps as (int)PyInt_AsLong does
        qi = np.empty(len(contig_quals), np.int32)         # (bernoullimodule.c:97)
        for i, v in enumerate(contig_quals):
            if not isinstance(v, int):
                raise TypeError("an integer is required")
            qi[i] = ((v + 0x80000000) & 0xFFFFFFFF) - 0x80000000
        return contig.encode(), qi, qi.ctypes.data, alpha


def runtime_dir():
    """Where this user's start lock and broker log live: $XDG_RUNTIME_DIR when it is this user's own directory, else a 0700
    directory of this user under /dev/shm (or the temp dir).  ADVICE r4: not a predictable name in a world-writable directory."""
    d = os.environ.get("XDG_RUNTIME_DIR")
    if d and os.path.isdir(d) and os.stat(d).st_uid == os.getuid():
        return d
    base = "/dev/shm" if os.path.isdir("/dev/shm") else tempfile.gettempdir()
    d = os.path.join(base, "moira_pb_%d" % os.getuid())
    try:
        os.mkdir(d, 0o700)
    except FileExistsError:
        pass
    st = os.lstat(d)
    if not stat.S_ISDIR(st.st_mode) or st.st_uid != os.getuid() or (st.st_mode & 0o077):
        raise L.MoiraPBError("%s is not a private directory of this user" % d)
    return d


def open_private(path, flags):
    """A file of this user only; a symbolic link in its place is an error, never followed."""
    return os.open(path, flags | os.O_CREAT | os.O_NOFOLLOW | os.O_CLOEXEC, 0o600)


class BrokerGone(L.MoiraPBError):
    pass


class BrokerClient:
    """This process's attachment to a broker: no GPU context, no HIP call."""

    def __init__(self, name, wait_ms=0):
        self.lib = L.load()
        self.name = name
        h = C.c_void_p()
        L.check(self.lib.mpb_broker_attach(name.encode(), int(wait_ms), C.byref(h)))
        self.h, self.pid = h, os.getpid()
        self._ee, self._ns = C.c_double(), C.c_int32()

    def calculate_errors_PB(self, contig, contig_quals, alpha):
        """bernoulli.calculate_errors_PB(contig, contig_quals, alpha) -> (expected_errors, Ns), through the broker."""
        seq, qi, addr, alpha = marshal_read(contig, contig_quals, alpha)
        rc = self.lib.mpb_broker_call(self.h, seq, addr, len(qi), alpha, C.byref(self._ee), C.byref(self._ns))
        if rc == L.E_HIP:
            raise BrokerGone(self.lib.mpb_last_error().decode(errors="replace"))
        L.check(rc)
        return self._ee.value, self._ns.value

    def close(self):
        if self.h is not None and self.h.value:
            self.lib.mpb_broker_detach(self.h)
        self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


def stats(name):
    """{'served', 'batches', 'solo', 'pid', 'attached'} of a broker, or None when there is none."""
    lib = L.load()
    a, b, s = C.c_int64(), C.c_int64(), C.c_int64()
    p, n = C.c_int32(), C.c_int32()
    if lib.mpb_broker_stats(name.encode(), C.byref(a), C.byref(b), C.byref(s), C.byref(p), C.byref(n)) != L.OK:
        return None
    return {"served": a.value, "batches": b.value, "solo": s.value, "pid": p.value, "attached": n.value}


def shutdown(name, wait_s=10.0):
    """Ask a broker to leave and wait until it has."""
    lib = L.load()
    st = stats(name)
    lib.mpb_broker_shutdown(name.encode())
    t0 = time.time()
    while st and st["pid"] and time.time() - t0 < wait_s:
        try:
            os.kill(st["pid"], 0)
        except OSError:
            break
        if stats(name) is None:
            break
        time.sleep(0.02)
    # the start lock of this name stays where it is: unlinking it while another process waits on the old inode would give two
    # starters "the lock" at once; it is an empty file in this user's runtime directory


def start(device=0, name=None, slots=64, idle_exit=10.0, log=None):
    """Start `python -m moira_amd.broker` as a fresh child (new session, nothing inherited but the environment)."""
    name = name or default_name(device)
    log = log or os.path.join(runtime_dir(), "broker_%s.log" % name)
    env = dict(os.environ)
    env["PYTHONPATH"] = ROOT + os.pathsep + env.get("PYTHONPATH", "")
    env.pop("MOIRA_PB_BROKER", None)
    with os.fdopen(open_private(log, os.O_WRONLY | os.O_APPEND), "ab") as lf:
        return subprocess.Popen([sys.executable, "-m", "moira_amd.broker", "--device", str(int(device)), "--name", name,
                                 "--slots", str(int(slots)), "--idle-exit", str(float(idle_exit))],
                                cwd=ROOT, env=env, stdin=subprocess.DEVNULL, stdout=lf, stderr=lf,
                                start_new_session=True, close_fds=True), log


def client(device=0, name=None, slots=64, idle_exit=10.0, start_timeout=120.0):
    """Attach to the broker of (user, device); start one first when there is none (exactly one, whatever the number of
    processes that ask at the same time)."""
    name = name or default_name(device)
    try:
        return BrokerClient(name, 0)
    except (ValueError, L.MoiraPBError):
        pass
    # one starter at a time (a broker process builds a GPU context before it asks for the name: starting P of them to have
    # P - 1 refused by mpb_broker_serve -- which IS race-free by itself -- would cost seconds)
    lock = os.path.join(runtime_dir(), "start_%s.lock" % name)
    with os.fdopen(open_private(lock, os.O_RDWR), "r+b") as lk:
        fcntl.flock(lk, fcntl.LOCK_EX)
        try:
            try:
                return BrokerClient(name, 0)               # someone else started it while we waited for the lock
            except (ValueError, L.MoiraPBError):
                pass
            proc, log = start(device, name, slots, idle_exit)
            t0 = time.time()
            while True:
                try:
                    return BrokerClient(name, 200)
                except (ValueError, L.MoiraPBError) as e:
                    if proc.poll() is not None:
                        tail = ""
                        try:
                            tail = open(log, "rb").read()[-600:].decode(errors="replace")
                        except OSError:
                            pass
                        raise L.NoDeviceError("the broker process exited with code %s before serving (no GPU?): %s"
                                              % (proc.returncode, tail.strip() or e))
                    if time.time() - t0 > start_timeout:
                        raise L.MoiraPBError("the broker did not come up within %.0f s: %s" % (start_timeout, e))
        finally:
            fcntl.flock(lk, fcntl.LOCK_UN)


def serve(device=0, name=None, slots=64, idle_exit=10.0):
    """The broker loop (blocking): owns the GPU context of `device` until shut down or idle."""
    from .engine import Engine
    name = name or default_name(device)
    eng = Engine(int(device))
    try:
        L.check(eng.lib.mpb_broker_serve(eng.ctx, name.encode(), int(slots), int(float(idle_exit) * 1000)))
    finally:
        eng.close()


def main(argv=None):
    import argparse
    ap = argparse.ArgumentParser(description="GPU-owning broker for per-read callers (bernoulli.calculate_errors_PB "
                                             "from the worker processes of moira.py --processors P)")
    ap.add_argument("--device", type=int, default=0)
    ap.add_argument("--name", type=str, default=None)
    ap.add_argument("--slots", type=int, default=64, help="most worker processes served at once")
    ap.add_argument("--idle-exit", type=float, default=10.0, help="leave this many seconds after the last attached "
                    "process has gone (0: never)")
    a = ap.parse_args(argv)
    serve(a.device, a.name, a.slots, a.idle_exit)
    return 0


if __name__ == "__main__":
    sys.exit(main())
