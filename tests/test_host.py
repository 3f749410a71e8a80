"""CPU-only tests of the product's host code (libraytracer_amd.so loads and runs its host-side
entry points without a GPU): camera, .obj loader + transforms, triangle precompute, BVH
build + flattening — each against the oracle's det mode, bit for bit — and the reference's
error behaviour.  No compute kernel is launched here."""
import os

import numpy as np
import pytest

from conftest import GOLDEN
from flat_emulator import FlatScene


def u32(a):
    return np.ascontiguousarray(a, np.float32).view(np.uint32)


@pytest.mark.parametrize("W,H", [(256, 256), (1920, 1080), (3840, 2160), (250, 203), (1000, 800)])
def test_camera_matches_oracle(rt, orc, golden_meta, W, H):
    got = rt.Camera(W, H).floats()
    assert np.array_equal(u32(got), u32(orc.camera_default(W, H, orc.MATH_DET)))
    key = "%dx%d" % (W, H)
    if key in golden_meta["camera_det"]:
        assert np.array_equal(u32(got), u32(np.array(golden_meta["camera_det"][key], np.float32)))


def test_camera_verbatim_floats(rt):
    f = np.arange(12, dtype=np.float32) * np.float32(0.25)
    assert np.array_equal(rt.Camera(10, 20, floats=f).floats(), f)


@pytest.mark.parametrize("name,transforms", [
    ("cube.obj", [("enlarge", 0.3), ("rotate", 0.4, 0.7, 0), ("translate", 0, 0, 1.8)]),
    ("low_poly_monkey.obj", [("enlarge", 0.3), ("rotate", 0, 2.3, 0), ("translate", 0.1, -0.1, 1.6)]),
    ("low_poly_monkey.obj", [("rotate", 1.1, -0.4, 3.0), ("enlarge", 2.5), ("translate", -1, 2, 3)]),
])
def test_obj_loader_and_transforms_match_oracle(rt, orc, models_dir, name, transforms):
    a = rt.ObjFileMesh(os.path.join(models_dir, name))
    b = orc.Obj(os.path.join(models_dir, name), orc.MATH_DET)
    assert (a.num_vertices, a.num_faces) == (b.num_vertices, b.num_faces)
    assert a.face_arities() == b.face_arities()
    for t in transforms:
        getattr(a, t[0])(*t[1:])
        getattr(b, t[0])(*t[1:])
    assert np.array_equal(u32(a.vertices()), u32(b.vertices()))
    assert np.array_equal(u32(a.triangles()), u32(b.triangles()))


def test_monkey_triangles_match_golden(rt, models_dir):
    m = rt.ObjFileMesh(os.path.join(models_dir, "low_poly_monkey.obj"))
    m.enlarge(0.3); m.rotate(0, 2.3, 0); m.translate(0.1, -0.1, 1.6)
    assert np.array_equal(u32(m.triangles()), u32(np.load(os.path.join(GOLDEN, "monkey_triangles_det.npy"))))


def test_obj_counts(rt, models_dir):
    # SURVEY.md §4 loader row: 8/6 -> 12 tris; 384/723
    c = rt.ObjFileMesh(os.path.join(models_dir, "cube.obj"))
    assert (c.num_vertices, c.num_faces, c.triangles().shape[0]) == (8, 6, 12) and set(c.face_arities()) == {4}
    m = rt.ObjFileMesh(os.path.join(models_dir, "low_poly_monkey.obj"))
    assert (m.num_vertices, m.num_faces, m.triangles().shape[0]) == (384, 723, 723)


def test_obj_from_arrays_equals_file(rt, models_dir):
    v, faces = rt.scenes.load_model_arrays("cube")
    a = rt.ObjFileMesh.from_arrays(v, faces)
    b = rt.ObjFileMesh(os.path.join(models_dir, "cube.obj"))
    assert np.array_equal(a.vertices(), b.vertices()) and a.faces() == b.faces()


def test_error_behaviour(rt, tmp_path):
    # read_file src/obj_read.cu:10 -> std::runtime_error("Could not find file to open.")
    with pytest.raises(rt.RayTracerError, match="Could not find file to open."):
        rt.ObjFileMesh(str(tmp_path / "missing.obj"))
    # src/main.cu:141 -> std::logic_error("Only triangle or quad meshes are supported.")
    p = tmp_path / "pent.obj"
    p.write_text("v 0 0 0\nv 1 0 0\nv 1 1 0\nv 0.5 1.5 0\nv 0 1 0\nf 1 2 3 4 5\n")
    m = rt.ObjFileMesh(str(p))
    with pytest.raises(rt.UnsupportedMeshError, match="Only triangle or quad meshes are supported."):
        m.triangles()
    so = rt.SceneObjects()
    with pytest.raises(rt.UnsupportedMeshError):
        so.create_mesh(m, rt.Material.create_standard((1, 1, 1), 0))
    assert so.num_objects == 0
    # an IMAGE material without texels is refused, never dereferenced
    bad = rt.Material.create_image(np.zeros((2, 2, 3), np.float32), 0)
    bad.c.img_w = 0
    with pytest.raises(ValueError):
        so.create_sphere((0, 0, 1), 0.5, bad)


def test_baked_texture_file(rt, tmp_path):
    """ImageTexture src/main.cu:40-91 on the textures/parse_textures.py format"""
    img = rt.scenes.procedural_image(5, 3)
    p = tmp_path / "parsed_textures.txt"
    with open(p, "w") as f:
        f.write("other.png\n1\n1\n0.5 0.5 0.5 \n")
        f.write("earth.png\n5\n3\n" + "".join("%s %s %s " % tuple(repr(float(c)) for c in px) for row in img for px in row) + "\n")
    got = rt.load_image_texture(str(p), "earth.png")
    assert got.shape == (3, 5, 3) and np.array_equal(got, img)
    with pytest.raises(rt.RayTracerError, match="Image file not found"):
        rt.load_image_texture(str(p), "mars.png")
    with pytest.raises(rt.RayTracerError, match="Could not find file to open"):
        rt.load_image_texture(str(tmp_path / "nope.txt"), "earth.png")


def test_ragged_obj_text(rt, tmp_path):
    """empty lines, comments, vn/vt lines, i/j/k and bare indices, a trailing empty line"""
    p = tmp_path / "r.obj"
    p.write_text("# c\n\nmtllib x.mtl\no thing\nv 0.5 -1.25 3\nv 1 0 0\nvn 0 1 0\nvt 0.5 0.5\nv 0 1 0\nv 1 1 0\ns 0\nf 1/2/3 2/2/3 3/1/1\nf 1 2 4 3\n\n")
    m = rt.ObjFileMesh(str(p))
    assert m.num_vertices == 4 and m.faces() == [[0, 1, 2], [0, 1, 3, 2]]
    assert m.vertices()[0].tolist() == [0.5, -1.25, 3.0]
    t = m.triangles()
    assert t.shape == (3, 9)
    assert t[1].reshape(3, 3).tolist() == m.vertices()[[0, 1, 3]].tolist()      # quad -> (v1,v2,v3)
    assert t[2].reshape(3, 3).tolist() == m.vertices()[[0, 2, 3]].tolist()      # ... + (v1,v4,v3)


def test_empty_scene_and_empty_mesh_flatten(rt):
    so = rt.SceneObjects()
    f = so.debug_flatten()
    assert f["num_triangles"] == 0 and f["num_nodes"] == 0 and len(f["objects"]) == 0
    so.create_mesh(np.zeros((0, 9), np.float32), rt.Material.create_standard((1, 1, 1), 0))
    f = so.debug_flatten()
    assert f["has_mesh"] and int(f["objects"][0]["root_ref"]) == 0x80000000


def _rays(n, seed, target=(0.0, -0.05, 1.6), spread=0.9):
    rng = np.random.default_rng(seed)
    o = np.zeros((n, 3), np.float32)
    o[n // 2:] = (rng.normal(size=(n - n // 2, 3)) * 0.8 + np.array(target)).astype(np.float32)
    t = (rng.normal(size=(n, 3)) * spread * 0.4 + np.array(target)).astype(np.float32)
    d = (t - o).astype(np.float32)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    return o, d


@pytest.mark.parametrize("scene_name,nrays", [("monkey", 160), ("cube", 100), ("reference_scene0", 80), ("reference_scene1", 80)])
def test_flattened_scene_closest_hit_matches_oracle(rt, orc, models_dir, scene_name, nrays):
    objs, _ = rt.scenes.CONFIG_SCENES[scene_name]()
    so = rt.SceneObjects(objs)
    flat = FlatScene(so.debug_flatten())
    oracle = orc.Scene(objs, orc.MATH_DET, models_dir)
    origins, dirs = _rays(nrays, 5)
    hits = 0
    for o, d in zip(origins, dirs):
        oh, out = oracle.trace_one(o, d)
        fh, ft, fobj, fn = flat.closest_hit(o, d)
        assert fh == oh
        if oh:
            hits += 1
            assert u32(np.float32(ft)) == u32(out[0])
            assert fobj == int(out[7])
            assert np.array_equal(u32(fn), u32(out[4:7]))
    assert hits > nrays // 4
    assert flat.max_stack <= 10           # RT_STACK_ENTRIES


def test_flattened_bvh_shape(rt, orc, models_dir):
    for name, expect in (("monkey", [0, 241, 241, 0]), ("cube", [0, 4, 4, 0])):
        objs, _ = rt.scenes.CONFIG_SCENES[name]()
        flat = FlatScene(rt.SceneObjects(objs).debug_flatten())
        assert flat.leaf_histogram(0) == expect            # the reference's non-empty leaves
        _, hist = orc.Scene(objs, orc.MATH_DET, models_dir).bvh_info(0, 4)
        assert hist[1:] == expect[1:]


def test_nan_and_axis_parallel_rays(rt, orc, models_dir):
    """Directions with exact zeros give infinite reciprocals and 0*inf = NaN inside the slab
    test; a NaN direction walks the whole tree and hits nothing."""
    objs, _ = rt.scenes.monkey()
    flat = FlatScene(rt.SceneObjects(objs).debug_flatten())
    oracle = orc.Scene(objs, orc.MATH_DET, models_dir)
    cases = [((0, 0, 0), (0, 0, 1)), ((0.1, -0.1, 0), (0, 0, 1)), ((0.1, 3, 1.6), (0, -1, 0)), ((-3, -0.1, 1.6), (1, 0, 0)),
             ((0, 0, 0), (np.nan, np.nan, np.nan))]
    for o, d in cases:
        oh, out = oracle.trace_one(np.array(o, np.float32), np.array(d, np.float32))
        fh, ft, fobj, fn = flat.closest_hit(o, d)
        assert fh == oh
        if oh:
            assert u32(np.float32(ft)) == u32(out[0]) and fobj == int(out[7])


def test_tile_owned_rows(rt):
    assert rt.tile_owned_rows(1080, 8, 0, 1) == 1080
    assert rt.tile_owned_rows(1080, 8, 0, 8) == 17 * 8 and rt.tile_owned_rows(1080, 8, 7, 8) == 16 * 8
    assert sum(rt.tile_owned_rows(1080, 8, r, 8) for r in range(8)) == 1080
    assert rt.tile_owned_rows(203, 8, 1, 2) == 13 * 8        # 26 bands (last one ragged), 13 each
    assert rt.tile_owned_rows(100, 8, 2, 2) == -1             # band_first out of range


def test_obj_loader_fuzz_against_the_oracle(rt, orc, tmp_path):
    """200 random .obj texts - number formats (fixed, exponent, signs, no leading zero), triangle and quad faces as bare
    indices or i/j/k, comments / vn / vt / o / s / usemtl lines, empty lines, a missing final newline - through the
    product's loader (rt_obj_load) and the oracle's restatement of ObjFileMesh (src/obj_read.cu:47-147), then through
    random enlarge / rotate / translate: vertices, face arities and the quad-split triangle list must agree bit for bit."""
    rng = np.random.default_rng(2024)

    def num(x):
        k = rng.integers(0, 5)
        if k == 0:
            return "%.6f" % x
        if k == 1:
            return "%.3e" % x
        if k == 2:
            return repr(float(np.float32(x)))
        if k == 3:
            return ("%.4f" % x).replace("0.", ".", 1) if abs(x) < 1 else "%.2f" % x
        return "%d" % int(x)

    for case in range(200):
        nv = int(rng.integers(3, 40))
        lines = ["# fuzz %d" % case, "mtllib nothing.mtl", "o thing"]
        for _ in range(nv):
            x, y, z = rng.uniform(-3, 3, 3)
            lines.append("v %s %s %s" % (num(x), num(y), num(z)))
            if rng.integers(0, 4) == 0:
                lines.append(rng.choice(["vn 0 1 0", "vt 0.5 0.25", "", "s off", "usemtl m"]))
        for _ in range(int(rng.integers(1, 30))):
            idx = rng.choice(nv, min(nv, int(rng.choice([3, 3, 4]))), replace=False) + 1
            style = rng.integers(0, 3)
            toks = ["%d" % i if style == 0 else ("%d/%d/%d" % (i, rng.integers(1, 9), rng.integers(1, 9)) if style == 1 else "%d//%d" % (i, rng.integers(1, 9))) for i in idx]
            lines.append("f " + " ".join(toks))
        text = "\n".join(lines) + ("\n" if rng.integers(0, 2) else "")
        p = tmp_path / ("f%d.obj" % case)
        p.write_text(text)
        a, b = rt.ObjFileMesh(str(p)), orc.Obj(str(p), orc.MATH_DET)
        steps = []
        for _ in range(int(rng.integers(0, 4))):
            k = rng.integers(0, 3)
            steps.append(("enlarge", float(rng.uniform(0.1, 3))) if k == 0 else
                         ("rotate",) + tuple(float(v) for v in rng.uniform(-7, 7, 3)) if k == 1 else
                         ("translate",) + tuple(float(v) for v in rng.uniform(-5, 5, 3)))
        for s in steps:
            getattr(a, s[0])(*s[1:])
            getattr(b, s[0])(*s[1:])
        assert a.num_vertices == b.num_vertices and a.face_arities() == b.face_arities(), case
        assert np.array_equal(a.vertices().view(np.uint32), b.vertices().view(np.uint32)), (case, steps)
        assert np.array_equal(a.triangles().view(np.uint32), b.triangles().view(np.uint32)), (case, steps)
