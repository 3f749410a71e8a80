"""Committed golden frames (tests/golden/, written by tools/make_golden.py from the oracle's
det mode).  CPU: the oracle still reproduces them.  GPU: the HIP kernel reproduces them bit
for bit through the C ABI, with the oracle not involved at all."""
import hashlib
import os

import numpy as np
import pytest

from conftest import GOLDEN

SCENES = ["three_sphere", "cube", "monkey", "reference_scene0", "reference_scene1", "reference_scene2", "reference_scene3", "reference_scene4"]


@pytest.mark.parametrize("name", SCENES)
def test_oracle_reproduces_golden(orc, rt, models_dir, golden_meta, name):
    g = golden_meta["frames"][name]
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    sc = orc.Scene(objs, orc.MATH_DET, models_dir)
    img = sc.render(orc.camera_default(g["W"], g["H"], orc.MATH_DET), g["W"], g["H"], g["spp"], g["limit"], sky,
                    time_ms=golden_meta["time_ms"], frame_num=0)
    want = np.load(os.path.join(GOLDEN, g["file"]))
    assert np.array_equal(img.view(np.uint32), want.view(np.uint32))
    assert hashlib.sha256(img.tobytes()).hexdigest() == g["sha256"]


def test_oracle_thread_count_invariance(orc, rt, models_dir):
    objs, sky = rt.scenes.monkey()
    sc = orc.Scene(objs, orc.MATH_DET, models_dir)
    cam = orc.camera_default(96, 72, orc.MATH_DET)
    a = sc.render(cam, 96, 72, 4, 8, sky, nthreads=1)
    b = sc.render(cam, 96, 72, 4, 8, sky, nthreads=7)
    assert np.array_equal(a.view(np.uint32), b.view(np.uint32))


def test_oracle_progressive_golden(orc, rt, models_dir):
    objs, sky = rt.scenes.three_sphere()
    sc = orc.Scene(objs, orc.MATH_DET, models_dir)
    cam = orc.camera_default(96, 64, orc.MATH_DET)
    want = np.load(os.path.join(GOLDEN, "fb_three_sphere_96x64_progressive.npy"))
    f0 = sc.render(cam, 96, 64, 4, 4, sky, time_ms=111, frame_num=0)
    f1 = sc.render(cam, 96, 64, 4, 4, sky, time_ms=222, frame_num=1, prev=f0)
    assert np.array_equal(np.stack([f0, f1]).view(np.uint32), want.view(np.uint32))


@pytest.mark.gpu
@pytest.mark.parametrize("name", SCENES)
def test_hip_reproduces_golden(rt, ctx, golden_meta, name):
    g = golden_meta["frames"][name]
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    scene = ctx.commit(rt.SceneObjects(objs))
    data = rt.VariableRenderData(g["W"], g["H"])
    rt.render(ctx, scene, rt.Camera(g["W"], g["H"]), rt.RenderData(g["spp"], g["limit"], True, sky), data, golden_meta["time_ms"])
    want = np.load(os.path.join(GOLDEN, g["file"]))
    assert np.array_equal(data.previous_render.view(np.uint32), want.view(np.uint32))
    assert data.frame_num == 1


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["three_sphere", "cube", "monkey"])
def test_hip_reproduces_config_shape_hashes(rt, ctx, golden_meta, name):
    """256x256, 16 spp: the shape SURVEY.md App. C.2 quotes the reference on"""
    g = golden_meta["sha256_256x256_s16"][name]
    objs, sky = rt.scenes.CONFIG_SCENES[name]()
    scene = ctx.commit(rt.SceneObjects(objs))
    data = rt.VariableRenderData(256, 256)
    rt.render(ctx, scene, rt.Camera(256, 256), rt.RenderData(16, g["limit"], True, sky), data, golden_meta["time_ms"])
    assert hashlib.sha256(data.previous_render.tobytes()).hexdigest() == g["sha256"]


@pytest.mark.gpu
def test_hip_progressive_golden(rt, ctx):
    """VariableRenderData semantics of src/dispatch.cu:111-163: prev is blended and overwritten,
    frame_num counts up"""
    objs, sky = rt.scenes.three_sphere()
    scene = ctx.commit(rt.SceneObjects(objs))
    want = np.load(os.path.join(GOLDEN, "fb_three_sphere_96x64_progressive.npy"))
    data = rt.VariableRenderData(96, 64)
    cam, rd = rt.Camera(96, 64), rt.RenderData(4, 4, True, sky)
    rt.render(ctx, scene, cam, rd, data, 111)
    assert np.array_equal(data.previous_render.view(np.uint32), want[0].view(np.uint32)) and data.frame_num == 1
    rt.render(ctx, scene, cam, rd, data, 222)
    assert np.array_equal(data.previous_render.view(np.uint32), want[1].view(np.uint32)) and data.frame_num == 2
