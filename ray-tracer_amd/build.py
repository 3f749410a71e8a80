"""Builds ray-tracer_amd/libraytracer_amd.so (HIP kernels + C ABI) in-tree with hipcc for gfx950.

hipcc cross-compiles without a GPU; the built .so travels with the repo snapshot to the GPU
box.  -ffp-contract=off is mandatory: the reference's `a*b+c` are two roundings and a fused
multiply-add changes hit/miss decisions (hard part 1).
"""
import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
LIB = os.path.join(HERE, "libraytracer_amd.so")
SOURCES = [os.path.join(HERE, "csrc", f) for f in ("rt_kernel.hip", "rt_capi.cpp", "rt_host.cpp")]
HEADERS = [os.path.join(HERE, "csrc", f) for f in ("rt_math.h", "rt_rng.h", "rt_device_scene.h", "rt_pixel.h", "rt_host.h")] + [
    os.path.join(ROOT, "include", "rt_amd.h")]
# -fno-slp-vectorize: the SLP vectorizer pairs the scalar f32 adds / multiplies of the vector math into v_pk_*_f32, which
# are not faster on gfx950 and need register pairs: 128 instead of ~90 VGPRs and ~10 % more time (same-box A/B, round 2).
FLAGS = ["--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-fPIC", "-shared", "-std=c++17",
         "-Wall", "-Wno-unused-result", "-I" + os.path.join(ROOT, "include")]


def hipcc():
    exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    if not os.path.exists(exe):
        raise RuntimeError("hipcc not found: the HIP extension cannot be built")
    return exe


def needs_build():
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    return any(os.path.getmtime(p) > t for p in SOURCES + HEADERS)


def build_variant(name, extra_flags):
    """development builds (e.g. -DRT_STATS) next to the product library; never loaded by default"""
    out = os.path.join(HERE, "libraytracer_amd_%s.so" % name)
    subprocess.check_call([hipcc()] + FLAGS + list(extra_flags) + SOURCES + ["-o", out])
    return out


def build(force=False, verbose=False):
    if not force and not needs_build():
        return LIB
    cmd = [hipcc()] + FLAGS + SOURCES + ["-o", LIB]
    if verbose:
        print(" ".join(cmd))
    subprocess.check_call(cmd)
    return LIB


EXAMPLE = os.path.join(HERE, "host", "example_main")


def build_example():
    """The C++ host-side mirror (host/raytracer.hpp) compiled into a small program with g++."""
    build()
    cmd = ["g++", "-std=c++17", "-O2", "-Wall", os.path.join(HERE, "host", "example_main.cpp"), "-o", EXAMPLE,
           "-L" + HERE, "-lraytracer_amd", "-Wl,-rpath," + HERE, "-Wl,-rpath,/opt/rocm/lib", "-Wl,--allow-shlib-undefined"]
    subprocess.check_call(cmd)
    return EXAMPLE


if __name__ == "__main__":
    print(build(force=True, verbose=True))
