"""Scene descriptions: the BASELINE.json config scenes and the reference's
own hard-coded demo scenes (reference src/main.cu:150-288), written as plain data.

A description is a list of object tuples in list order (order matters: the later object wins
distance ties, reference src/raytracer.cu:36):

    ('sphere', center, radius, material)
    ('triangle', p1, p2, p3, material)
    ('triangle_uv', [p1,p2,p3], [uv1,uv2,uv3], material)
    ('quad', p1, p2, p3, p4, material)
    ('one_way_quad', p1, p2, p3, p4, invert_normal, material)
    ('cuboid', tl_near_pos, width, height, depth, material)
    ('mesh', triangles[n,9], material)
    ('obj', filename, [('enlarge', s) | ('rotate', x, y, z) | ('translate', x, y, z), ...], material)

and a material is ('standard', colour, smoothness) | ('emissive', colour, strength) |
('checkerboard', light, dark, num_squares, smoothness) | ('gradient', smoothness) |
('refractive', colour, n) | ('image', rgb[h,w,3], smoothness) — the arguments of the reference's Material::create_* /
Texture::create_* factories (src/material.cu:21-51, :157-185).
"""

import os
import tempfile

import numpy as np

_MODELS_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "models")
_models_dir = None


def write_obj(path, vertices, faces):
    """Wavefront text the reference's loader accepts (src/obj_read.cu:92-147): `v x y z` lines
    (%.9g round-trips float32), `vn`/`vt` lines it must ignore, 1-based `f i/j/k` faces."""
    with open(path, "w") as f:
        f.write("# written by ray-tracer_amd.scenes.write_obj\no mesh\n")
        for v in np.asarray(vertices, np.float32):
            f.write("v %.9g %.9g %.9g\n" % (v[0], v[1], v[2]))
        f.write("vn 0.0000 1.0000 0.0000\nvt 0.500000 0.500000\ns 0\n")
        for face in faces:
            f.write("f " + " ".join("%d/1/1" % (i + 1) for i in face) + "\n")


def load_model_arrays(name):
    """(vertices[n,3] float32, faces as lists of 0-based indices) of a shipped model.  The
    arrays were extracted from the reference's models/*.obj by tools/make_model_fixtures.py."""
    z = np.load(os.path.join(_MODELS_NPZ, os.path.splitext(name)[0] + ".npz"))
    faces, k = [], 0
    for a in z["face_arity"]:
        faces.append([int(i) for i in z["face_indices"][k:k + a]])
        k += int(a)
    return z["vertices"].astype(np.float32), faces


def models_dir():
    """A directory holding cube.obj and low_poly_monkey.obj, written from the shipped arrays."""
    global _models_dir
    if _models_dir is None or not os.path.isdir(_models_dir):
        d = tempfile.mkdtemp(prefix="rt_amd_models_")
        for name in ("cube", "low_poly_monkey"):
            v, f = load_model_arrays(name)
            write_obj(os.path.join(d, name + ".obj"), v, f)
        _models_dir = d
    return _models_dir


SKY_COLOUR = (0.8, 1.0, 1.0)          # reference src/main.cu:13
NO_SKY = (0.0, 0.0, 0.0)              # reference src/main.cu:328


def std(colour, smoothness):
    return ("standard", tuple(colour), float(smoothness))


def three_sphere():
    """BASELINE configs[0], configs[1]; SURVEY.md App. C.1 'three-sphere'."""
    return [
        ("sphere", (0, -100.5, 1.5), 100, std((0.8, 0.8, 0.0), 0)),
        ("sphere", (-0.6, 0, 1.5), 0.5, std((0.7, 0.3, 0.3), 0)),
        ("sphere", (0.6, 0, 1.5), 0.5, std((0.3, 0.3, 0.7), 0)),
    ], SKY_COLOUR


def cube():
    """BASELINE configs[2]; the rotation is mandatory (an axis-aligned cube loses faces to the
    reference BVH's strict slab test)."""
    return [
        ("obj", "cube.obj", [("enlarge", 0.3), ("rotate", 0.4, 0.7, 0), ("translate", 0, 0, 1.8)], std((0.8, 0.4, 0.2), 0)),
        ("sphere", (0, -100.5, 1.5), 100, std((0.5, 0.5, 0.5), 0)),
    ], SKY_COLOUR


def monkey():
    """BASELINE configs[3], configs[4]; mesh transform = reference src/main.cu:157-160."""
    return [
        ("obj", "low_poly_monkey.obj", [("enlarge", 0.3), ("rotate", 0, 2.3, 0), ("translate", 0.1, -0.1, 1.6)], std((1, 1, 1), 0)),
        ("sphere", (0, 1.2, 1.2), 0.5, ("emissive", (1, 1, 1), 6)),
        ("sphere", (0, -100.5, 1.5), 100, std((0.5, 0.5, 0.5), 0.3)),
    ], NO_SKY


def _f(x):
    return np.float32(x)


def _add(a, b):
    """Vec3 + Vec3 in float32, like the reference's host-side Vec3 arithmetic."""
    return tuple(float(_f(x) + _f(y)) for x, y in zip(a, b))


def _sub(a, b):
    return tuple(float(_f(x) - _f(y)) for x, y in zip(a, b))


def cornell_box(tl=(-0.5, 0.5, 1.2), width=1.0, height=1.0, depth=1.0, light_width=0.5):
    """reference src/main.cu:252-288 (create_cornell_box); vector sums in float32, left to right."""
    w, h, d = (width, 0, 0), (0, height, 0), (0, 0, depth)
    floor = ("checkerboard", (0.1, 0.8, 0.1), (0.1, 0.5, 0.1), 8, 0)
    objs = [
        ("quad", _sub(tl, h), _add(_sub(tl, h), w), _add(_add(_sub(tl, h), w), d), _add(_sub(tl, h), d), floor),
        ("quad", tl, _sub(tl, h), _add(_sub(tl, h), d), _add(tl, d), std((1, 0.2, 0.2), 0)),
        ("quad", _add(tl, w), _sub(_add(tl, w), h), _add(_sub(_add(tl, w), h), d), _add(_add(tl, w), d), std((0.3, 0.3, 1), 0)),
        ("quad", _add(tl, d), _add(_add(tl, w), d), _add(_sub(_add(tl, w), h), d), _add(_sub(tl, h), d), std((0.2, 0.2, 0.2), 0)),
        ("quad", tl, _add(tl, d), _add(_add(tl, w), d), _add(tl, w), std((0.9, 0.9, 0.9), 0)),
        ("one_way_quad", tl, _add(tl, w), _sub(_add(tl, w), h), _sub(tl, h), False, std((1, 1, 1), 0)),
        ("cuboid", (float(_f(tl[0]) + _f(width) / _f(2) - _f(light_width) / _f(2)), tl[1],
                    float(_f(tl[2]) + _f(depth) / _f(2) - _f(light_width) / _f(2))),
         light_width, 0.04, light_width, ("emissive", (1, 1, 1), 6)),
    ]
    return objs


def reference_scene0():
    """reference src/main.cu:150-170 (monkey_test_scene)."""
    objs = cornell_box()
    objs.append(("obj", "low_poly_monkey.obj", [("enlarge", 0.3), ("rotate", 0, 2.3, 0), ("translate", 0.1, -0.1, 1.6)], std((1, 1, 1), 0)))
    objs.append(("sphere", (-0.25, -0.25, 1.95), 0.25, std((0.8, 0.8, 0.8), 1)))
    return objs, NO_SKY


def reference_scene1():
    """reference src/main.cu:172-187 (reflection_test_scene)."""
    objs = cornell_box()
    for c, s in (((-0.2, 0.2, 1.7), 0), ((0.2, 0.2, 1.7), 0.33), ((-0.2, -0.2, 1.7), 0.66), ((0.2, -0.2, 1.7), 1)):
        objs.append(("sphere", c, 0.15, std((1, 1, 1), s)))
    return objs, NO_SKY


def reference_scene3():
    """reference src/main.cu:206-213 (refract_test_scene)."""
    objs = cornell_box()
    objs.append(("sphere", (0, -0.1, 1.7), 0.3, ("refractive", (1, 1, 1), 1.5)))
    return objs, NO_SKY


def procedural_image(width=64, height=32, seed=1):
    """A stand-in for the reference's earth.png (git-ignored upstream): smooth
    coloured bands + noise, float32 [height, width, 3] in [0, 1), quantised to n/256 like
    textures/parse_textures.py does."""
    rng = np.random.default_rng(seed)
    y, x = np.mgrid[0:height, 0:width]
    img = np.stack([0.5 + 0.5 * np.sin(x / 5.0), 0.5 + 0.5 * np.cos(y / 3.0), (x + y) % 16 / 16.0], axis=2)
    img = np.clip(img * 0.85 + rng.uniform(0, 0.15, img.shape), 0, 0.999)
    return (np.floor(img * 256) / 256).astype(np.float32)


def reference_scene2(image=None):
    """reference src/main.cu:189-204 (texture_test_scene) with a procedural image in place of
    the absent earth.png."""
    objs = cornell_box()
    objs.append(("sphere", (0, 0, 1.7), 0.25, ("image", procedural_image() if image is None else image, 0)))
    objs.append(("triangle_uv", [(0.1, 0, 1.7), (0.6, 0.5, 1.9), (0.8, 0.4, 2)], [(0, 0), (0, 1), (1, 1)],
                 ("checkerboard", (1, 1, 1), (0, 0, 0), 4, 0)))
    return objs, NO_SKY


def reference_scene4(seed=2024, num_spheres=100):
    """reference src/main.cu:215-250 (rand_sphere_test_scene) with a SEEDED host generator (the
    reference draws from std::random_device, so its scene differs on every run).  30 % standard,
    30 % refractive, 40 % default-constructed Material — undefined in the reference (SURVEY.md
    App. A.9), defined here as a black diffuse STANDARD material."""
    rng = np.random.default_rng(seed)

    def host_rng(lo, hi):
        return float(np.float32(lo + np.float32(rng.random()) * (hi - lo)))

    floor_y, floor_w, floor_d = -1.0, 10.0, 10.0
    objs = []
    for _ in range(num_spheres):
        colour = (host_rng(0, 1), host_rng(0, 1), host_rng(0, 1))
        mat_num = host_rng(0, 1)
        if mat_num < 0.3:
            mat = std(colour, host_rng(0, 1))
        elif mat_num < 0.6:
            mat = ("refractive", colour, host_rng(0.5, 2))
        else:
            mat = std((0, 0, 0), 0)
        radius = host_rng(0.1, 0.5)
        center = (host_rng(-floor_w / 2, floor_w / 2), float(np.float32(floor_y) + np.float32(radius)), host_rng(0, floor_d))
        objs.append(("sphere", center, radius, mat))
    objs.append(("quad", (-floor_w / 2, floor_y, 0), (floor_w / 2, floor_y, 0), (floor_w / 2, floor_y, floor_d), (-floor_w / 2, floor_y, floor_d),
                 ("checkerboard", (0.7, 0.7, 0.7), (0.4, 0.4, 0.4), 10, 0)))
    return objs, SKY_COLOUR


def soup6k(n=6000, seed=9):
    """A mesh too large for a CU's LDS (the kernel's global-memory path): n random small triangles over a
    checkerboard ground.  Synthetic: the reference ships no model of this size (its Mesh takes any triangle count,
    src/objects.cu:780-787)."""
    rng = np.random.default_rng(seed)
    centres = rng.uniform([-1.2, -0.6, 1.2], [1.2, 0.8, 3.5], (n, 3))
    tris = (centres[:, None, :] + rng.normal(0, 0.05, (n, 3, 3))).astype(np.float32).reshape(n, 9)
    objs = [("mesh", tris, ("standard", (0.8, 0.7, 0.6), 0.1)),
            ("sphere", (0, -100.5, 1.5), 100, ("checkerboard", (0.9, 0.9, 0.9), (0.3, 0.3, 0.3), 4000, 0))]
    return objs, SKY_COLOUR


def bumpy_sphere(rings=160, segments=160, seed=5):
    """A closed tessellated surface of 2 * rings * segments - 2 * segments triangles (50,880 by default: ~50 per leaf
    of the reference's fixed-depth-10 BVH), radius 0.55 with a smooth radial bump, under the monkey scene's light and
    over its ground.  Synthetic, for the beyond-LDS measurements."""
    rng = np.random.default_rng(seed)
    ph = rng.uniform(0, 2 * np.pi, 6)
    th = np.linspace(0, np.pi, rings + 1)[:, None]
    ps = np.linspace(0, 2 * np.pi, segments, endpoint=False)[None, :]
    r = 0.55 * (1 + 0.08 * np.sin(5 * th + ph[0]) * np.cos(4 * ps + ph[1]) + 0.04 * np.sin(11 * th + ph[2]) * np.sin(9 * ps + ph[3]))
    P = np.stack([r * np.sin(th) * np.cos(ps), r * np.cos(th) + 0 * ps, r * np.sin(th) * np.sin(ps)], axis=2) + np.array([0.1, 0.0, 1.7])
    P = P.astype(np.float32)
    tris = []
    for i in range(rings):
        for j in range(segments):
            a, b = P[i, j], P[i, (j + 1) % segments]
            c, d = P[i + 1, j], P[i + 1, (j + 1) % segments]
            if i > 0:
                tris.append(np.concatenate([a, b, d]))
            if i < rings - 1:
                tris.append(np.concatenate([a, d, c]))
    objs = [("mesh", np.asarray(tris, np.float32), std((0.9, 0.85, 0.8), 0.05)),
            ("sphere", (0, 1.2, 1.2), 0.5, ("emissive", (1, 1, 1), 6)),
            ("sphere", (0, -100.5, 1.5), 100, std((0.5, 0.5, 0.5), 0.3))]
    return objs, NO_SKY


CONFIG_SCENES = {"three_sphere": three_sphere, "cube": cube, "monkey": monkey, "soup6k": soup6k, "sphere50k": bumpy_sphere,
                 "reference_scene0": reference_scene0, "reference_scene1": reference_scene1,
                 "reference_scene2": reference_scene2, "reference_scene3": reference_scene3, "reference_scene4": reference_scene4}
