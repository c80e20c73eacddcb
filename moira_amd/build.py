"""Build libmoira_pb.so (the HIP library behind the C ABI) in-tree with hipcc for gfx950.

    python -m moira_amd.build          # builds moira_amd/libmoira_pb.so if stale
    python -m moira_amd.build --force

hipcc cross-compiles gfx950 code objects without a GPU, so this runs in the build
container; the .so travels to the GPU box with the source snapshot.
-ffp-contract=off is REQUIRED: the DP cell and the interpolation must round exactly as the
reference (gcc, baseline x86-64, no FMA) does.
"""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libmoira_pb.so")
SOURCES = [os.path.join(CSRC, "mpb_kernels.hip"), os.path.join(CSRC, "mpb_api.cpp")]
DEPS = SOURCES + [os.path.join(CSRC, "mpb_internal.h"),
                  os.path.join(ROOT, "include", "moira_pb.h"),
                  os.path.join(ROOT, "include", "mpb_synth.h")]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["-O3", "--offload-arch=gfx950", "-fPIC", "-shared", "-std=c++17",
         "-ffp-contract=off", "-fno-fast-math", "-Wall", "-Wno-unused-function",
         "-x", "hip"]


CONTIG_LIB = os.path.join(HERE, "libmoira_contig.so")
CONTIG_SRC = os.path.join(CSRC, "contig.cpp")
CXX = os.environ.get("CXX", "g++")


def contig_stale():
    if not os.path.exists(CONTIG_LIB):
        return True
    t = os.path.getmtime(CONTIG_LIB)
    return any(os.path.getmtime(d) > t for d in (CONTIG_SRC, os.path.join(ROOT, "include", "moira_contig.h")))


def build_contig(force=False, verbose=False):
    """CPU-only contig construction library (g++, no HIP): north_star keeps NW on the CPU."""
    if not force and not contig_stale():
        return CONTIG_LIB
    cmd = [CXX, "-O2", "-fPIC", "-shared", "-std=c++17", "-ffp-contract=off", "-pthread", "-Wall",
           CONTIG_SRC, "-o", CONTIG_LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return CONTIG_LIB


def stale():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(d) > t for d in DEPS)


def build(force=False, verbose=False, extra=()):
    if not force and not stale():
        return LIB
    cmd = [HIPCC] + FLAGS + list(extra) + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    build(force="--force" in sys.argv, verbose=True)
    build_contig(force="--force" in sys.argv, verbose=True)
    print(LIB)
    print(CONTIG_LIB)
