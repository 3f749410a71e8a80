"""Generates tests/golden/* from the CPU oracle (run in the build container; commit the output).

The oracle is pinned to the reference first (tests/test_oracle_pin.py: in libm mode it
reproduces the reference outputs recorded in SURVEY.md); the fixtures written here come from
its det mode (transcendentals from rt_math.h), which is what the HIP kernel must equal bit
for bit.  Nothing here reads /root/reference.
"""
import hashlib
import importlib
import json
import os
import platform
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")
from oracle import binding as B  # noqa: E402

GOLD = os.path.join(ROOT, "tests", "golden")
os.makedirs(GOLD, exist_ok=True)
md = rt.scenes.models_dir()

SCENES = [("three_sphere", 4), ("cube", 8), ("monkey", 8), ("reference_scene0", 5), ("reference_scene1", 5),
          ("reference_scene2", 5), ("reference_scene3", 5), ("reference_scene4", 5)]
meta = {"time_ms": 12345, "frame_num": 0, "antialias": True, "generator": "tools/make_golden.py",
        "python": platform.python_version(), "libc": " ".join(platform.libc_ver()), "frames": {}, "sha256_256x256_s16": {}}

for name, limit in SCENES:
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    sc = B.Scene(objs, B.MATH_DET, md)
    W = H = 128
    img = sc.render(B.camera_default(W, H, B.MATH_DET), W, H, 16, limit, sky)
    fn = "fb_%s_128x128_s16.npy" % name
    np.save(os.path.join(GOLD, fn), img)
    meta["frames"][name] = {"file": fn, "W": W, "H": H, "spp": 16, "limit": limit, "sky": list(sky),
                            "mean": float(img.mean(dtype=np.float64)), "sha256": hashlib.sha256(img.tobytes()).hexdigest()}
    big = sc.render(B.camera_default(256, 256, B.MATH_DET), 256, 256, 16, limit, sky)
    meta["sha256_256x256_s16"][name] = {"limit": limit, "sha256": hashlib.sha256(big.tobytes()).hexdigest(),
                                        "mean": float(big.mean(dtype=np.float64))}
    print(name, meta["frames"][name]["sha256"][:16], meta["sha256_256x256_s16"][name]["sha256"][:16])

# progressive two-frame fixture (frame_num / prev blending, reference src/raytracer.cu:109-112)
objs, sky = rt.scenes.three_sphere()
sc = B.Scene(objs, B.MATH_DET, md)
cam = B.camera_default(96, 64, B.MATH_DET)
f0 = sc.render(cam, 96, 64, 4, 4, sky, time_ms=111, frame_num=0)
f1 = sc.render(cam, 96, 64, 4, 4, sky, time_ms=222, frame_num=1, prev=f0)
np.save(os.path.join(GOLD, "fb_three_sphere_96x64_progressive.npy"), np.stack([f0, f1]))

# cameras
meta["camera_det"] = {"%dx%d" % (w, h): [float(x) for x in B.camera_default(w, h, B.MATH_DET)] for w, h in ((256, 256), (1920, 1080), (3840, 2160))}

# The reference's own camera floats (libm mode = the glibc 2.35 tanf the survey's CPU build of the reference used;
# the values are SURVEY.md App. A.12's, tests/test_oracle_pin.py checks them) as DATA: passed verbatim through
# rt_camera they make the HIP kernel reproduce the reference outputs the survey recorded (App. C.2) - the math
# binding (rt_math.h vs glibc) flips no decision in those frames, the 1-ulp tanf of the camera does.
meta["camera_libm"] = {"%dx%d" % (w, h): [float(x) for x in B.camera_default(w, h, B.MATH_LIBM)] for w, h in ((256, 256), (1920, 1080), (3840, 2160))}
meta["reference_recorded_256x256_s16"] = {"three_sphere": {"limit": 4, "sha256_prefix": "479589c110c5b34e", "mean": 0.472370008},
                                          "cube": {"limit": 8, "sha256_prefix": "b4dcdd058b1bc676", "mean": 0.650935728},
                                          "monkey": {"limit": 8, "sha256_prefix": "24682f69ae058766", "mean": 0.19924736, "max": 7.828125},
                                          "source": "SURVEY.md App. C.2 (outputs of the reference's sources compiled CPU-only by the survey session)"}

# transformed monkey triangles (det-mode sin/cos), 723 x 9
ob = B.Obj(os.path.join(md, "low_poly_monkey.obj"), B.MATH_DET)
ob.enlarge(0.3); ob.rotate(0, 2.3, 0); ob.translate(0.1, -0.1, 1.6)
np.save(os.path.join(GOLD, "monkey_triangles_det.npy"), ob.triangles())

# rt_math known answers: inputs and the exact output bits
rng = np.random.default_rng(7)
u = (rng.integers(0, 2**32, 4000, dtype=np.uint64).astype(np.float64) / 4294967295.0).astype(np.float32)
xs_log = np.concatenate([u, np.array([0.0, 1.0, 2.0 ** -32, 1e-40, 0.5, 0.70710677, 0.70710683, 3.0, 1e30], np.float32)])
xs_trig = np.concatenate([(u * np.float32(6.28318)).astype(np.float32), np.array([0, 0.5235988, 0.7853982, 0.78539824, 1.5707964, 3.1415927, 100.0, -3.0, 3000.0, 4000.0, 1e6], np.float32)])
L = B.lib()
kat = {"xs_log": xs_log, "xs_trig": xs_trig}
for fname, xs in (("logf", xs_log), ("cosf", xs_trig), ("sinf", xs_trig), ("tanf", xs_trig)):
    f = getattr(L, "orc_math_" + fname)
    kat[fname] = np.array([f(float(x), B.MATH_DET) for x in xs], np.float32).view(np.uint32)
xs_inv = np.concatenate([rng.uniform(-1, 1, 2000), np.array([-1.0, -0.5, 0.0, 0.5, 1.0, 0.5000001, 0.9999999, 1e-9])])
kat["xs_inv"] = xs_inv
kat["asin"] = np.array([L.orc_math_asin(float(x), B.MATH_DET) for x in xs_inv], np.float64).view(np.uint64)
kat["acos"] = np.array([L.orc_math_acos(float(x), B.MATH_DET) for x in xs_inv], np.float64).view(np.uint64)
np.savez_compressed(os.path.join(GOLD, "math_kat.npz"), **kat)

with open(os.path.join(GOLD, "meta.json"), "w") as fh:
    json.dump(meta, fh, indent=1, sort_keys=True)
print("wrote", GOLD)
