"""The library's host-side code under AddressSanitizer + UndefinedBehaviorSanitizer (CPU; the GPU pool has no sanitizer runs).

ray-tracer_amd/csrc/rt_host.cpp - the .obj and baked-texture parsers, the scene builder, the BVH build and the flattening into the
device layout: everything that reads files a user hands over - is compiled with -fsanitize=address,undefined together with
tests/sanitize/host_fuzz.cpp, which drives it through the C ABI with random well-formed and malformed input.  (Round 4's first run of
it found rt_obj_get_triangles reading vertices a face named but the file did not have.)"""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_fuzzed_under_asan_and_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = str(tmp_path / "host_fuzz")
    cmd = [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "ray-tracer_amd", "csrc"), os.path.join(ROOT, "tests", "sanitize", "host_fuzz.cpp"),
           os.path.join(ROOT, "ray-tracer_amd", "csrc", "rt_host.cpp"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    if r.returncode != 0 and "sanitize" in (r.stderr or "") and "unrecognized" in r.stderr:
        pytest.skip("this g++ has no sanitizer runtime")
    assert r.returncode == 0, r.stderr[-2000:]
    scratch = tmp_path / "scratch"
    scratch.mkdir()
    for seed in (11, 12, 13):
        r = subprocess.run([exe, str(scratch), str(seed), "1500"], capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "sanitizers silent" in r.stdout, (seed, (r.stderr or r.stdout)[-3000:])


def test_a_face_naming_a_missing_vertex_is_an_error_not_a_read(rt, tmp_path):
    p = tmp_path / "bad.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 1 2 7\n")
    m = rt.ObjFileMesh(str(p))
    assert m.num_vertices == 3 and m.num_faces == 1
    with pytest.raises(ValueError, match="missing vertex"):
        m.triangles()
    p.write_text("v 0 0 0\nv 1 0 0\nv 0 1 0\nf 0 1 2\n")          # index 0: one before the first
    with pytest.raises(ValueError, match="missing vertex"):
        rt.ObjFileMesh(str(p)).triangles()
    so = rt.SceneObjects()
    with pytest.raises(ValueError, match="missing vertex"):
        so.create_mesh(rt.ObjFileMesh(str(p)), rt.Material.create_standard((1, 1, 1), 0.0))


def test_gpu_free_entry_points_of_the_c_abi_under_asan_and_ubsan(tmp_path):
    """rt_capi.cpp as host C++ (kernel launchers stubbed out): rt_partition_tiles on random costs and sizes, rt_tile_owned_rows, and
    every entry point handed null / bad arguments - refused before HIP is touched, sanitizers silent"""
    gxx = shutil.which("g++")
    rocm = "/opt/rocm"
    if gxx is None or not os.path.exists(os.path.join(rocm, "include", "hip", "hip_runtime_api.h")):
        pytest.skip("no g++ or no HIP headers")
    exe = str(tmp_path / "capi_fuzz")
    cmd = [gxx, "-std=c++17", "-g", "-O1", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-D__HIP_PLATFORM_AMD__",
           "-I", os.path.join(rocm, "include"), "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "ray-tracer_amd", "csrc"),
           os.path.join(ROOT, "tests", "sanitize", "capi_host_fuzz.cpp"), os.path.join(ROOT, "ray-tracer_amd", "csrc", "rt_capi.cpp"),
           os.path.join(ROOT, "ray-tracer_amd", "csrc", "rt_host.cpp"), "-L" + os.path.join(rocm, "lib"), "-lamdhip64",
           "-Wl,-rpath," + os.path.join(rocm, "lib"), "-o", exe]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stderr[-2000:]
    env = dict(os.environ, ASAN_OPTIONS="detect_leaks=0")          # (the HIP runtime keeps what it allocates at start-up)
    for seed in (1, 2):
        r = subprocess.run([exe, str(seed), "3000"], capture_output=True, text=True, timeout=300, env=env)
        assert r.returncode == 0 and "sanitizers silent" in r.stdout, (seed, (r.stderr or r.stdout)[-3000:])


_ORACLE_SCRIPT = r"""
import sys, importlib
sys.path.insert(0, ROOT); sys.path.insert(0, ROOT + "/tests")
from oracle import binding as B
B._LIB_PATH = LIB
scenes = importlib.import_module("ray-tracer_amd.scenes")
from test_gpu_parity import _random_scene
import numpy as np
F = np.float32
def cam(W, H):
    # any camera will do here: the reference's default pose written out (no product library in this process)
    return [0.0, 0.0, -1.0, -0.5, 0.5 * H / W, 0.0, 1.0 / W, 0.0, 0.0, 0.0, -1.0 / W, 0.0]
models = scenes.models_dir()
n = 0
for name in ["monkey", "cube", "three_sphere", "reference_scene0", "reference_scene2", "reference_scene3", "reference_scene4"]:
    objs, sky = scenes.CONFIG_SCENES[name]()
    o = B.Scene(objs, B.MATH_DET, models)
    prev = None
    for f in range(2):
        prev = o.render(cam(45, 31), 45, 31, 2, 6, sky, time_ms=99 + f, frame_num=f, prev=prev)
    n += 1
for seed in range(40):
    objs, sky = _random_scene(seed)
    B.Scene(objs, B.MATH_DET if seed % 2 else B.MATH_LIBM, models).render(cam(33, 27), 33, 27, 2, 5, sky, time_ms=seed)
    n += 1
print("oracle: %d scenes rendered, sanitizers silent" % n)
"""


def test_oracle_under_asan_and_ubsan(tmp_path):
    """the checker itself: oracle/rt_oracle.c built with -fsanitize=address,undefined renders config and random scenes (meshes, textures,
    refraction, both math modes) in a child process with the sanitizer runtimes preloaded"""
    import sys
    gcc = shutil.which("gcc")
    if gcc is None:
        pytest.skip("no gcc")
    asan = subprocess.run([gcc, "-print-file-name=libasan.so"], capture_output=True, text=True).stdout.strip()
    ubsan = subprocess.run([gcc, "-print-file-name=libubsan.so"], capture_output=True, text=True).stdout.strip()
    if not (os.path.isabs(asan) and os.path.exists(asan) and os.path.isabs(ubsan) and os.path.exists(ubsan)):
        pytest.skip("no sanitizer runtimes")
    lib = str(tmp_path / "librt_oracle_asan.so")
    fma = ["-mfma"] if "fma" in open("/proc/cpuinfo").read().split() else []
    cmd = [gcc, "-std=gnu11", "-O1", "-g", "-ffp-contract=off"] + fma + ["-fno-fast-math", "-fno-builtin-tanf", "-fno-builtin-sinf", "-fno-builtin-cosf",
           "-fno-builtin-logf", "-fPIC", "-pthread", "-fsanitize=address,undefined", "-fno-sanitize-recover=all", "-shared", "-o", lib,
           os.path.join(ROOT, "oracle", "rt_oracle.c"), "-lm"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    script = tmp_path / "run.py"
    script.write_text("ROOT = %r\nLIB = %r\n" % (ROOT, lib) + _ORACLE_SCRIPT)
    env = dict(os.environ, LD_PRELOAD=asan + ":" + ubsan, ASAN_OPTIONS="detect_leaks=0")
    r = subprocess.run([sys.executable, str(script)], capture_output=True, text=True, timeout=900, env=env)
    assert r.returncode == 0 and "sanitizers silent" in r.stdout, (r.stderr or r.stdout)[-3000:]
