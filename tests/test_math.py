"""rt_math.h: the deterministic log / sin / cos / tan shared by the HIP kernel and the oracle's
det mode.  CPU-side checks (through the oracle's orc_math_* hooks, which call rt_math.h when
mode == DET): accuracy against float64 numpy, special values, and exact known-answer bits so a
compiler that evaluates the header differently is caught."""
import os

import numpy as np

from conftest import GOLDEN


def _ulps(got, ref64):
    ref32 = ref64.astype(np.float32)
    ulp = np.abs(np.nextafter(ref32, np.float32(np.inf)).astype(np.float64) - ref32.astype(np.float64))
    return np.abs(got.astype(np.float64) - ref64) / ulp


def test_logf_accuracy_on_rng_range(orc):
    L = orc.lib()
    rng = np.random.default_rng(1)
    u = (rng.integers(1, 2**32, 20000, dtype=np.uint64).astype(np.float64) / 4294967295.0).astype(np.float32)
    got = np.array([L.orc_math_logf(float(x), orc.MATH_DET) for x in u], np.float32)
    mask = u < 1.0                    # log(1) = 0 exactly; ulps of 0 are meaningless
    assert _ulps(got[mask], np.log(u[mask].astype(np.float64))).max() <= 1.0


def test_log_special_values(orc):
    L = orc.lib()
    f = lambda x: L.orc_math_logf(x, orc.MATH_DET)
    assert f(0.0) == -np.inf and f(1.0) == 0.0 and np.isnan(f(-1.0)) and f(np.inf) == np.inf
    assert abs(f(1e-40) - np.log(1e-40)) < 1e-4       # subnormal input
    assert np.isnan(f(float("nan")))


def test_trig_accuracy(orc):
    L = orc.lib()
    xs = np.linspace(0, 6.2832, 20001).astype(np.float32)
    c = np.array([L.orc_math_cosf(float(x), orc.MATH_DET) for x in xs], np.float32)
    s = np.array([L.orc_math_sinf(float(x), orc.MATH_DET) for x in xs], np.float32)
    # absolute error in units of 2^-24 (one ulp at 1.0)
    assert (np.abs(c.astype(np.float64) - np.cos(xs.astype(np.float64))) / 2.0**-24).max() <= 1.5
    assert (np.abs(s.astype(np.float64) - np.sin(xs.astype(np.float64))) / 2.0**-24).max() <= 1.5
    big = np.array([-4000.0, 3217.0, 1e6, 1e9], np.float32)
    for x in big:
        assert abs(L.orc_math_cosf(float(x), orc.MATH_DET) - np.cos(np.float64(x))) < 1e-6
        assert abs(L.orc_math_sinf(float(x), orc.MATH_DET) - np.sin(np.float64(x))) < 1e-6
    assert np.isnan(L.orc_math_cosf(float("inf"), orc.MATH_DET))


def test_math_known_answer_bits(orc):
    kat = np.load(os.path.join(GOLDEN, "math_kat.npz"))
    L = orc.lib()
    for fname, xs in (("logf", kat["xs_log"]), ("cosf", kat["xs_trig"]), ("sinf", kat["xs_trig"]), ("tanf", kat["xs_trig"])):
        f = getattr(L, "orc_math_" + fname)
        got = np.array([f(float(x), orc.MATH_DET) for x in xs], np.float32).view(np.uint32)
        assert np.array_equal(got, kat[fname]), fname


def test_asin_acos_accuracy_and_known_answers(orc):
    """binary64 asin / acos used by the refraction code (src/ray.cu:101-102) and, rounded to
    float, by the sphere texture coordinates (src/objects.cu:84-85)"""
    L = orc.lib()
    xs = np.linspace(-1, 1, 4001)
    a = np.array([L.orc_math_asin(float(x), orc.MATH_DET) for x in xs])
    c = np.array([L.orc_math_acos(float(x), orc.MATH_DET) for x in xs])
    assert np.abs(a - np.arcsin(xs)).max() < 1e-15 and np.abs(c - np.arccos(xs)).max() < 1e-15
    assert L.orc_math_asin(1.0, orc.MATH_DET) == np.pi / 2 and L.orc_math_acos(1.0, orc.MATH_DET) == 0.0
    assert np.isnan(L.orc_math_asin(1.0000001, orc.MATH_DET)) and np.isnan(L.orc_math_acos(-2.0, orc.MATH_DET))
    kat = np.load(os.path.join(GOLDEN, "math_kat.npz"))
    for name in ("asin", "acos"):
        f = getattr(L, "orc_math_" + name)
        got = np.array([f(float(x), orc.MATH_DET) for x in kat["xs_inv"]], np.float64).view(np.uint64)
        assert np.array_equal(got, kat[name]), name


def test_dense_accuracy_against_libm(tmp_path):
    """The oracle's det mode and the HIP kernel share rt_math.h, so the bit-exact parity tests cannot see an
    error in it; this is the independent check: EVERY binary32 in (1e-10, 1] for log (the RNG's whole range),
    every binary32 in [1e-6, 6.2832] for sin / cos (the Box-Muller angle), a dense sweep of +-[6.28, 3000], and
    4 million points of the binary64 asin / acos / pow5, each against the platform libm in binary64
    (tests/math_accuracy.c, ~7 s)."""
    import re
    import subprocess
    from conftest import ROOT
    exe = str(tmp_path / "math_accuracy")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + (["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []) + [ os.path.join(ROOT, "tests", "math_accuracy.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe, os.environ.get("RT_MATH_STRIDE", "1")], capture_output=True, text=True, timeout=900).stdout
    v = {k: float(x) for k, x in re.findall(r"(\w+) ([0-9.e+-]+)(?= |$)", out) if k != "at"}
    assert v["log_ulp"] <= 1.0, out                    # measured 0.983 ulp
    assert v["sin_2m24"] <= 1.5 and v["cos_2m24"] <= 1.5 and v["bigarg_2m24"] <= 1.5, out     # measured 1.17 / 1.31 / 1.34 x 2^-24
    assert v["asin_abs"] < 1e-15 and v["acos_abs"] < 1e-15 and v["pow5_abs"] < 1e-15, out


def test_short_division_equals_the_operator_where_its_call_sites_use_it(tmp_path):
    """rt_math.h rt__div_benign (reciprocal + one residual step; Markstein) against `/` on 60 million random operand pairs per class:
    a general benign class, the operands of Box-Muller's f / (2 + f), the sphere test's dividend / (2 |d|^2) from 2^-90 up.  (Below
    ~2^-103 the residual underflows and quotients differ - the sphere test rejects such distances either way: rt_pixel.h.)"""
    import os
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "divchk")
    fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off"] + fma + [os.path.join(root, "tests", "div_benign_check.c"), "-o", exe, "-lm"])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout
    assert "general 0, logarithm 0, sphere 0" in out.stdout
