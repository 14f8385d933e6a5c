"""Registers, spills and scratch of every gfx950 kernel in the built library, read from the code objects' metadata
(no compile): `python tools/kernel_resources.py [pattern]`.  tests/test_kernel_resources.py holds the hot kernels to
their budgets with it -- a change that pushes one of them over an occupancy step or into spills shows up on the CPU,
before any GPU time is spent (round 3: a run-time test inside the local bit-parallel column group cost 28 spilled
registers and 40 % of the kernel's speed, and only the next bench run showed it)."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "bio_amd", "lib", "libbioalign.so")
LLVM = "/opt/rocm/lib/llvm/bin"


def kernel_resources(lib=LIB):
    """-> {mangled kernel name: dict(vgpr, sgpr, spill, scratch, lds)}"""
    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", so], check=True, capture_output=True, cwd=tmp)
        for f in sorted(os.listdir(tmp)):
            if "amdgcn" not in f:
                continue
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", os.path.join(tmp, f)], check=True,
                                   capture_output=True, text=True).stdout
            # one YAML-like block per kernel under amdhsa.kernels: fields in a fixed (alphabetical) order, .name among them
            for block in re.split(r"\n\s+- \.agpr_count:", notes)[1:]:
                def field(key, block=block):
                    m = re.search(r"\." + key + r":\s+(\S+)", block)
                    return m.group(1) if m else None
                name = field("name")
                if name is None:
                    continue
                out[name] = dict(vgpr=int(field("vgpr_count")), sgpr=int(field("sgpr_count")),
                                 spill=int(field("vgpr_spill_count")), scratch=int(field("private_segment_fixed_size")),
                                 lds=int(field("group_segment_fixed_size")))
    return out


def demangle(names):
    r = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True, check=True)
    return dict(zip(names, r.stdout.splitlines()))


if __name__ == "__main__":
    res = kernel_resources()
    pretty = demangle(sorted(res))
    pat = sys.argv[1] if len(sys.argv) > 1 else ""
    print(f"{'VGPRs':>5} {'spill':>5} {'scratch':>7} {'LDS':>6}  kernel")
    for k in sorted(res, key=lambda k: pretty[k]):
        if pat in pretty[k]:
            v = res[k]
            print(f"{v['vgpr']:5d} {v['spill']:5d} {v['scratch']:7d} {v['lds']:6d}  {pretty[k].replace('(BaDpParams, int)', '').replace('(BaTraceParams)', '')[:110]}")
