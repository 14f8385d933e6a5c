"""The hot kernels keep the registers and scratch their occupancy was tuned for (tools/kernel_resources.py reads them
from the built library's code objects; nothing is compiled or run).  A change that pushes one of them over an occupancy
step (128 VGPRs: four wavefronts per SIMD, 168: three) or into register spills fails here, on the CPU, instead of
showing up as a slower bench line."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))
import kernel_resources as KR  # noqa: E402

# kernel (demangled, without its parameter list) -> (most VGPRs, most scratch bytes per lane)
BUDGET = {
    # configs[1], unit scoring: the bit-parallel kernels
    "void ba_dp_bp_kernel<1, 4, 5>": (168, 16),    # local, five words of rows: three wavefronts per SIMD, one 8-byte spill per column group
    "void ba_dp_bp_kernel<1, 4, 3>": (128, 0),
    "void ba_dp_bp_kernel<0, 8, 5>": (128, 0),
    "void ba_dp_bp_kernel<2, 8, 5>": (128, 0),
    "void ba_dp_bp_kernel<3, 8, 5>": (128, 0),
    "void ba_trace_bp_ckpt_kernel<8, 0>": (128, 0),
    "void ba_trace_bp_ckpt_kernel<8, 2>": (128, 0),
    "void ba_trace_bp_ckpt_kernel<8, 3>": (128, 0),
    "void ba_trace_bp_kernel<true>": (128, 0),
    "ba_expand_bundle_kernel": (128, 0),
    # the general linear kernels (any small-integer scoring)
    "void ba_dp_linear_kernel<0, 32, 2>": (128, 0),
    "void ba_dp_linear_kernel<1, 32, 2>": (128, 0),
    # configs[2]: protein local affine (nibble profile, float keys), configs[3]: DNA fitting / overlap affine
    "void ba_dp_affine_kernel<1, 16, true, true, true>": (128, 0),
    "void ba_dp_affine_kernel<2, 32, true, false, false>": (168, 64),
    "void ba_dp_affine_kernel<3, 32, true, false, false>": (168, 64),
    # configs[4]: one long pair
    "void ba_long_dp_kernel<0, 4, 2, false, false, false>": (128, 0),
}


def test_hot_kernels_keep_their_register_budgets():
    if not os.path.exists(os.path.join(KR.LLVM, "llvm-readelf")):
        pytest.skip("no llvm-readelf")
    res = KR.kernel_resources()
    pretty = KR.demangle(sorted(res))
    by_name = {}
    for k, v in res.items():
        by_name[pretty[k].split("(")[0]] = v
    for name, (vgpr, scratch) in BUDGET.items():
        assert name in by_name, f"{name}: not in the library"
        got = by_name[name]
        assert got["vgpr"] <= vgpr and got["scratch"] <= scratch, (name, got)
