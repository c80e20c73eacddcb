#!/usr/bin/env python3
"""Instruction mix of the hot loop of every DP class body (from hipcc -S), and the loop text of two of them.
    python tools/experiments/isa_mix.py > profiles/r01_k_dp_hot_loops.txt"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
out = os.path.join(tempfile.mkdtemp(), "k.s")
subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-ffp-contract=off",
                       "-fno-fast-math", "-S", "--cuda-device-only", "-x", "hip",
                       os.path.join(ROOT, "moira_amd", "csrc", "mpb_kernels.hip"), "-o", out],
                      stderr=subprocess.DEVNULL)
funcs, cur = {}, None
for line in open(out).read().split("\n"):
    m = re.match(r"^(_Z\S+):", line)
    if m:
        cur = m.group(1)
        funcs[cur] = []
    elif cur is not None:
        funcs[cur].append(line)


def f64(b):
    return sum(1 for x in b if re.search(r"v_(mul|add)_f64", x))


print("hot loop (the block with the most FP64 instructions) of dp_tiles<R, G, FMA=false>, gfx950, %s" %
      subprocess.check_output(["/opt/rocm/bin/hipcc", "--version"]).decode().split("\n")[0])
print("%3s %3s %6s %6s %10s %6s %6s %8s" % ("R", "G", "bases", "f64", "other VALU", "ds", "salu", "f64/VALU"))
keep = {}
for name, lines in funcs.items():
    m = re.search(r"dp_tilesILi(\d+)ELi(\d+)ELb0", name)
    if not m:
        continue
    R, G = int(m.group(1)), int(m.group(2))
    blocks, b = [], []
    for x in lines:
        if re.match(r"^\.LBB\d+_\d+:", x):
            blocks.append(b)
            b = [x]
        else:
            b.append(x)
    blocks.append(b)
    best = max(blocks, key=f64)
    valu = sum(1 for x in best if re.match(r"\s+v_", x))
    ds = sum(1 for x in best if re.match(r"\s+ds_", x))
    salu = sum(1 for x in best if re.match(r"\s+s_", x))
    print("%3d %3d %6d %6d %10d %6d %6d %8.3f" % (R, G, ds, f64(best), valu - f64(best), ds, salu, f64(best) / valu))
    keep[(R, G)] = best
for key in ((4, 1), (10, 8)):
    print("\n==== dp_tiles<R=%d, G=%d>: hot loop ====" % key)
    body = keep[key]
    print("\n".join(body[:140]))
    if len(body) > 140:
        print("        ; ... %d more lines of the same pattern" % (len(body) - 140))
