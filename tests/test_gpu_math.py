"""The shared headers rt_math.h / rt_rng.h must be the SAME function on the device and on the
host (that is what makes CPU-oracle parity meaningful).  Each function is evaluated on the GPU
(rt_debug_eval) and by a host build of the same header on ~1e6 inputs, including the special
values, and compared bit for bit; likewise the two IEEE operations everything else leans on
(sqrtf and the f32 divide)."""
import ctypes as C
import os
import subprocess

import numpy as np
import pytest

from conftest import ROOT

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def host_eval(tmp_path_factory):
    so = str(tmp_path_factory.mktemp("hosteval") / "libhosteval.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + (["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []) + [ "-shared", "-fPIC", os.path.join(ROOT, "tests", "math_host_eval.c"), "-o", so, "-lm"])
    L = C.CDLL(so)
    L.host_eval.argtypes = [C.c_int, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int]

    def run(op, bits):
        a = np.ascontiguousarray(bits, dtype=np.uint32)
        out = np.empty_like(a)
        L.host_eval(op, a.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_uint32)), a.size)
        return out
    return run


def _float_inputs(rng, lo, hi, n):
    x = rng.uniform(lo, hi, n).astype(np.float32)
    special = np.array([0.0, -0.0, 1.0, -1.0, 0.5, 2.0 ** -32, 1e-40, 1e-45, np.inf, -np.inf, np.nan, 3.4e38, 0.70710677, 0.78539816,
                        1.5707964, 3.1415927, 6.28318, 0.99999994, 1.0000001], np.float32)
    return np.concatenate([x, special]).view(np.uint32)


CASES = [
    (0, "rt_logf", lambda r: np.concatenate([_float_inputs(r, 0, 1, 400000), _float_inputs(r, 0, 1e6, 100000),
                                             (r.integers(0, 2**32, 300000, dtype=np.uint64).astype(np.float64) / 4294967295.0).astype(np.float32).view(np.uint32)])),
    (1, "rt_cosf", lambda r: np.concatenate([_float_inputs(r, 0, 6.2832, 600000), _float_inputs(r, -5000, 5000, 200000), _float_inputs(r, -1e9, 1e9, 50000)])),
    (2, "rt_sinf", lambda r: np.concatenate([_float_inputs(r, 0, 6.2832, 600000), _float_inputs(r, -5000, 5000, 200000), _float_inputs(r, -1e9, 1e9, 50000)])),
    (3, "rt_asinf", lambda r: _float_inputs(r, -1.001, 1.001, 800000)),
    (4, "rt_acosf", lambda r: _float_inputs(r, -1.001, 1.001, 800000)),
    (5, "rt_u01", lambda r: np.concatenate([r.integers(0, 2**32, 1000000, dtype=np.uint64).astype(np.uint32), np.array([0, 1, 0xfffffd7f, 0xffffff7f, 0xffffff80, 0xffffffff, 0x01000001], np.uint32)])),
    (6, "rt_jitter", lambda r: np.concatenate([r.integers(0, 2**32, 1000000, dtype=np.uint64).astype(np.uint32), np.array([0, 0x7fffffff, 0x80000000, 0xffffffff], np.uint32)])),
    (7, "rt_theta", lambda r: np.concatenate([r.integers(0, 2**32, 1000000, dtype=np.uint64).astype(np.uint32), np.array([0, 0x7fffffff, 0x80000000, 0xffffffff], np.uint32)])),
    (8, "sqrtf", lambda r: np.concatenate([_float_inputs(r, 0, 4, 500000), r.integers(0, 0x7f800000, 500000, dtype=np.uint64).astype(np.uint32)])),
    (9, "1/x", lambda r: np.concatenate([_float_inputs(r, -4, 4, 500000), r.integers(0, 2**32, 500000, dtype=np.uint64).astype(np.uint32)])),
    (10, "rt_pow5", lambda r: _float_inputs(r, 0, 1.2, 300000)),
    # the Box-Muller calls: the device evaluates rt_logf_0_1 / rt_cosf_0_2pi (log and cos without the cases a draw cannot produce), the
    # host the general rt_logf / rt_cosf on the same hash outputs (all 2^32 of them on the host alone: tests/test_rng_exhaustive.py)
    (13, "log of a draw", lambda r: np.concatenate([r.integers(0, 2**32, 4000000, dtype=np.uint64).astype(np.uint32), np.arange(0, 4096, dtype=np.uint32),
                                                    np.array([0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], np.uint32)])),
    (15, "log of a draw, division operator", lambda r: np.concatenate([r.integers(0, 2**32, 2000000, dtype=np.uint64).astype(np.uint32), np.arange(0, 4096, dtype=np.uint32),
                                                                       np.array([0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], np.uint32)])),
    (14, "cos of a draw", lambda r: np.concatenate([r.integers(0, 2**32, 4000000, dtype=np.uint64).astype(np.uint32), np.arange(0, 4096, dtype=np.uint32),
                                                    np.array([0x1fffffff, 0x20000000, 0x3fffffff, 0x40000000, 0x7fffffff, 0x80000000, 0xfffffffe, 0xffffffff], np.uint32)])),
]


@pytest.mark.parametrize("op,name,gen", CASES, ids=[c[1] for c in CASES])
def test_device_equals_host(rt, ctx, host_eval, op, name, gen):
    bits = gen(np.random.default_rng(100 + op))
    dev = rt.debug_eval(ctx, op, bits)
    host = host_eval(op, bits)
    # NaN results may differ in payload / sign between an x86 and a GPU operation; every
    # consumer only tests NaN-ness (comparisons that fail), so NaNs are compared as a class
    dev_nan = np.isnan(dev.view(np.float32))
    host_nan = np.isnan(host.view(np.float32))
    assert np.array_equal(dev_nan, host_nan)
    mism = (dev != host) & ~dev_nan
    assert not mism.any(), "%s: %d of %d differ, first input 0x%08x" % (name, int(mism.sum()), bits.size, int(bits[np.argmax(mism)]))


def test_short_reciprocal_and_square_root_are_the_ieee_ones_for_all_inputs(rt, ctx):
    """rt_pixel.h computes 1 / x as v_rcp_f32 + one Newton step and sqrt(x) as v_sqrt_f32 + a +-1 ulp residual test whenever every
    lane's operand is inside the range those short forms are exact on (2^-126 <= |x| <= 2^126; 2^-64 <= x < inf), and as the
    compiler's IEEE expansion otherwise.  "Exact" is checked here for EVERY binary32 input on the device: inside the ranges
    not one of the 4.2e9 (reciprocal) and 1.6e9 (square root) inputs may differ from 1.0f / x and sqrtf(x)."""
    bad_r, in_r, bad_s, in_s = rt.debug_exhaustive(ctx)
    assert in_r == 2 * (0x7e800000 - 0x00800000 + 1)           # both signs, 2^-126 .. 2^126 inclusive
    assert in_s == 0x7f800000 - 0x1f800000                      # 2^-64 .. the largest finite float
    assert bad_r == 0 and bad_s == 0, (bad_r, bad_s)


def test_guarded_forms_equal_the_operators_outside_the_ranges_too(rt, ctx):
    """per lane what rt_rcp / rt_sqrt decide per wave (short form in range, the operator outside): specials, denormals, huge and
    tiny values, negative arguments of the root"""
    rng = np.random.default_rng(9)
    bits = np.concatenate([rng.integers(0, 2**32, 200000, dtype=np.uint64).astype(np.uint32),
                           np.array([0, 0x80000000, 1, 0x007fffff, 0x00800000, 0x7e800000, 0x7e800001, 0x7f7fffff, 0x7f800000, 0xff800000, 0x7fc00000,
                                     0x1f7fffff, 0x1f800000, 0x3f800000, 0xbf800000], np.uint32)])
    with np.errstate(all="ignore"):
        x = bits.view(np.float32)
        want_r = (np.float32(1.0) / x).view(np.uint32)
        want_s = np.sqrt(x).view(np.uint32)
    got_r = rt.debug_eval(ctx, 11, bits)
    got_s = rt.debug_eval(ctx, 12, bits)
    nan = lambda u: (u & 0x7fffffff) > 0x7f800000
    assert np.array_equal(np.where(nan(got_r), 0x7fc00000, got_r), np.where(nan(want_r), 0x7fc00000, want_r))
    assert np.array_equal(np.where(nan(got_s), 0x7fc00000, got_s), np.where(nan(want_s), 0x7fc00000, want_s))
