"""The C-ABI shared library: it loads, exports every symbol include/rt_amd.h declares, and
refuses to work without a GPU (no CPU fallback)."""
import ctypes
import os
import re

import numpy as np

import pytest

from conftest import ROOT


def declared_symbols():
    text = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(rt_[a-z0-9_]+)\s*\(", text)))


def test_header_and_binding_agree(rt):
    assert declared_symbols() == sorted(rt.ABI_SYMBOLS)


def test_library_exports_every_declared_symbol(rt):
    L = rt.lib()
    for name in declared_symbols():
        assert getattr(L, name) is not None, name
    assert b"gfx950" in L.rt_version()


def test_struct_layouts(rt):
    # the ctypes mirrors must match the header's PODs (4-byte fields, no padding)
    assert ctypes.sizeof(rt.rt_material) == 4 * (2 + 3 + 3 + 3 + 1 + 1 + 1 + 3 + 1) + 8 + 8      # ... + img_w, img_h + pointer
    assert ctypes.sizeof(rt.rt_camera) == 4 * 14
    assert ctypes.sizeof(rt.rt_render_settings) == 4 * 6
    assert ctypes.sizeof(rt.rt_tile_spec) == 48       # four ints, three pointers, one int (+ padding)


def test_material_factories(rt):
    m = rt.Material.create_standard((0.7, 0.3, 0.3), 0.25).c
    assert (m.type, m.tex_type, m.need_uv) == (rt.MAT_STANDARD, rt.TEX_COLOUR, 0) and abs(m.smoothness - 0.25) < 1e-7
    e = rt.Material.create_emissive((1, 0.5, 0.25), 6).c
    # src/material.cu:170 emitted = colour * strength; fields the reference leaves unset are 0
    assert list(e.emitted_light) == [6.0, 3.0, 1.5] and e.smoothness == 0.0 and e.need_uv == 0 and e.type == rt.MAT_EMISSIVE
    c = rt.Material.create_checkerboard((1, 1, 1), (0, 0, 0), 8, 0).c
    assert (c.tex_type, c.need_uv, c.num_squares) == (rt.TEX_CHECKERBOARD, 1, 8)
    r = rt.Material.create_refractive((1, 1, 1), 1.5).c
    # src/material.cu:175-185: smoothness forced to 1
    assert (r.type, r.smoothness, r.need_uv) == (rt.MAT_REFRACTIVE, 1.0, 0) and abs(r.refractive_index - 1.5) < 1e-7
    i = rt.Material.create_image(np.zeros((4, 6, 3), np.float32), 0.5).c
    assert (i.tex_type, i.need_uv, i.img_w, i.img_h) == (rt.TEX_IMAGE, 1, 6, 4)


def test_no_gpu_means_failure_not_fallback(rt):
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    with pytest.raises(rt.RayTracerError, match="no CPU fallback"):
        rt.Context(0)


def test_pipelined_entry_points_refuse_a_null_context(rt):
    """no GPU needed: the frames-in-flight entry points check their context before they touch HIP"""
    import ctypes as C
    L = rt.lib()
    assert L.rt_frame_submit(None, None, None, None, 0, None) == rt.RT_ERR_INVALID
    assert L.rt_frame_collect(None, 0, None, None) == rt.RT_ERR_INVALID
    assert L.rt_frame_collect_host(None, C.byref(C.c_int32(0)), None) == rt.RT_ERR_INVALID
    assert L.rt_frame_wait(None) == rt.RT_ERR_INVALID
    assert L.rt_frame_depth(None, 4) == rt.RT_ERR_INVALID
    assert L.rt_frames_pending(None) == 0
    assert (rt.PIPELINE_DEFAULT_DEPTH, rt.PIPELINE_DEPTH) == (4, 8)
    # the header's macros say the same
    import os, re
    hdr = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "include", "rt_amd.h")).read()
    assert int(re.search(r"#define RT_PIPELINE_DEPTH (\d+)", hdr).group(1)) == rt.PIPELINE_DEPTH
    assert int(re.search(r"#define RT_PIPELINE_DEFAULT_DEPTH (\d+)", hdr).group(1)) == rt.PIPELINE_DEFAULT_DEPTH
    assert int(re.search(r"RT_ERR_BUSY = (\d+)", hdr).group(1)) == rt.RT_ERR_BUSY
