"""ray-tracer_amd — host-side Python mirror of the reference's scene / camera / render
interface over the C ABI of libraytracer_amd.so (include/rt_amd.h).

Import with ``importlib.import_module("ray-tracer_amd")`` (the directory name has a hyphen).

The names follow the reference: ``Material.create_standard`` (src/material.cu:157),
``Object``-style factories on :class:`SceneObjects` (src/objects.cu:845-906),
:class:`ObjFileMesh` (src/obj_read.cu:47), :class:`Camera` (src/camera.cu:32),
:class:`RenderData` (src/raytracer.cu:4), :class:`VariableRenderData` + :func:`render`
(src/dispatch.cu:111-163).  All compute happens in the HIP library; there is no CPU path:
creating a :class:`Context` without a GPU raises.
"""
import ctypes as C
import os
import sys

import numpy as np

from . import build as _build
from . import scenes  # noqa: F401  (re-exported)

_HERE = os.path.dirname(os.path.abspath(__file__))

RT_OK, RT_ERR_INVALID, RT_ERR_IO, RT_ERR_UNSUPPORTED, RT_ERR_HIP, RT_ERR_NOMEM, RT_ERR_NO_DEVICE, RT_ERR_BUSY = range(8)
TEX_COLOUR, TEX_GRADIENT, TEX_CHECKERBOARD, TEX_IMAGE = 0, 1, 2, 3
MAT_STANDARD, MAT_EMISSIVE, MAT_REFRACTIVE = 0, 1, 2


class rt_material(C.Structure):
    _fields_ = [("type", C.c_int32), ("tex_type", C.c_int32), ("colour", C.c_float * 3),
                ("light", C.c_float * 3), ("dark", C.c_float * 3), ("num_squares", C.c_int32),
                ("smoothness", C.c_float), ("need_uv", C.c_int32), ("emitted_light", C.c_float * 3),
                ("refractive_index", C.c_float), ("img_w", C.c_int32), ("img_h", C.c_int32),
                ("img_rgb", C.POINTER(C.c_float))]


class rt_camera(C.Structure):
    _fields_ = [("cam_pos", C.c_float * 3), ("tl_pixel_pos", C.c_float * 3), ("delta_u", C.c_float * 3),
                ("delta_v", C.c_float * 3), ("width", C.c_int32), ("height", C.c_int32)]


class rt_render_settings(C.Structure):
    _fields_ = [("rays_per_pixel", C.c_int32), ("reflection_limit", C.c_int32), ("antialias", C.c_int32),
                ("sky_colour", C.c_float * 3)]


class rt_tile_spec(C.Structure):
    _fields_ = [("band_rows", C.c_int32), ("band_first", C.c_int32), ("band_stride", C.c_int32), ("compact", C.c_int32),
                ("tile_list", C.POINTER(C.c_uint32)), ("tile_cost", C.POINTER(C.c_uint32)), ("tile_peak", C.POINTER(C.c_uint32)),
                ("num_tiles", C.c_int32)]


class rt_rank(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("scene", C.c_void_p)]


class rt_scene_info(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("num_objects", "num_triangles", "num_nodes", "lds_bytes", "scene_in_lds", "threads_per_block", "stack_entries", "blocks_per_cu")]


class rt_flat_view(C.Structure):
    _fields_ = [("blob", C.POINTER(C.c_float)), ("blob_f4", C.c_int32), ("off_nodes", C.c_int32),
                ("off_tris", C.c_int32), ("off_objlds", C.c_int32), ("off_meshes", C.c_int32), ("num_meshes", C.c_int32),
                ("stack_entries", C.c_int32), ("objects", C.c_void_p),
                ("num_objects", C.c_int32), ("object_stride", C.c_int32), ("tri_uv", C.POINTER(C.c_float)),
                ("num_triangles", C.c_int32), ("num_nodes", C.c_int32), ("has_mesh", C.c_int32)]


# every symbol include/rt_amd.h declares (tests/test_abi.py checks the .so exports them all)
ABI_SYMBOLS = [
    "rt_material_standard", "rt_material_checkerboard", "rt_material_gradient", "rt_material_emissive",
    "rt_material_refractive", "rt_material_image", "rt_image_texture_load", "rt_image_texture_free",
    "rt_scene_builder_create", "rt_scene_builder_destroy", "rt_scene_builder_error", "rt_scene_add_sphere",
    "rt_scene_add_triangle", "rt_scene_add_triangle_uv", "rt_scene_add_quad", "rt_scene_add_one_way_quad",
    "rt_scene_add_cuboid", "rt_scene_add_mesh", "rt_scene_add_obj_mesh", "rt_scene_builder_num_objects",
    "rt_obj_load", "rt_obj_destroy", "rt_obj_enlarge", "rt_obj_rotate", "rt_obj_translate",
    "rt_obj_num_vertices", "rt_obj_num_faces", "rt_obj_face_arity", "rt_obj_get_face", "rt_obj_from_arrays", "rt_obj_get_vertices",
    "rt_obj_num_triangles", "rt_obj_get_triangles", "rt_camera_default", "rt_camera_make",
    "rt_ctx_create", "rt_ctx_destroy", "rt_last_error", "rt_scene_commit", "rt_scene_destroy",
    "rt_scene_get_info", "rt_render", "rt_render_frames", "rt_render_device", "rt_render_device_batch", "rt_tile_owned_rows", "rt_last_kernel_ms",
    "rt_tile_costs", "rt_partition_tiles", "rt_tiles_copy_device", "rt_max_batch_frames", "rt_peer_access",
    "rt_ctx_synchronize", "rt_render_multi", "rt_render_multi_device", "rt_gather",
    "rt_frame_submit", "rt_frame_collect", "rt_frames_pending", "rt_frame_wait", "rt_frame_depth", "rt_frame_collect_host",
    "rt_to_rgba8_device", "rt_debug_flatten", "rt_debug_read_stats", "rt_debug_eval", "rt_debug_exhaustive", "rt_version",
]

_lib = None


def lib():
    """Load libraytracer_amd.so (building it with hipcc if the sources are newer).  Raises if
    the HIP extension cannot be built or loaded: there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    path = os.environ.get("RT_AMD_LIB") or _build.build()      # RT_AMD_LIB: development builds (tools/)
    # PyTorch-ROCm bundles its own libamdhip64.so.7.  Two HIP runtimes in one process cannot
    # both own the GPU, so when torch is installed it is imported first and this library then
    # binds (by SONAME) to the runtime torch already loaded; device pointers and streams are
    # then shared.  RT_AMD_NO_TORCH=1 skips this (pure ctypes use against /opt/rocm).
    if "torch" not in sys.modules and os.environ.get("RT_AMD_NO_TORCH", "0") != "1":
        try:
            import torch  # noqa: F401
        except ImportError:
            pass
    L = C.CDLL(path)
    fp = C.POINTER(C.c_float)
    vp = C.c_void_p
    pm = C.POINTER(rt_material)
    L.rt_material_standard.argtypes = [pm, fp, C.c_float]
    L.rt_material_checkerboard.argtypes = [pm, fp, fp, C.c_int32, C.c_float]
    L.rt_material_gradient.argtypes = [pm, C.c_float]
    L.rt_material_emissive.argtypes = [pm, fp, C.c_float]
    L.rt_material_refractive.argtypes = [pm, fp, C.c_float]
    L.rt_material_image.argtypes = [pm, C.c_int32, C.c_int32, fp, C.c_float]
    L.rt_image_texture_load.argtypes = [C.c_char_p, C.c_char_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(fp)]
    L.rt_image_texture_free.argtypes = [fp]
    L.rt_image_texture_free.restype = None
    for n in ("rt_material_standard", "rt_material_checkerboard", "rt_material_gradient", "rt_material_emissive",
              "rt_material_refractive", "rt_material_image"):
        getattr(L, n).restype = None
    L.rt_scene_builder_create.argtypes = [C.POINTER(vp)]
    L.rt_scene_builder_destroy.argtypes = [vp]
    L.rt_scene_builder_destroy.restype = None
    L.rt_scene_builder_error.argtypes = [vp]
    L.rt_scene_builder_error.restype = C.c_char_p
    L.rt_scene_add_sphere.argtypes = [vp, fp, C.c_float, pm]
    L.rt_scene_add_triangle.argtypes = [vp, fp, fp, fp, pm]
    L.rt_scene_add_triangle_uv.argtypes = [vp, fp, fp, pm]
    L.rt_scene_add_quad.argtypes = [vp, fp, fp, fp, fp, pm]
    L.rt_scene_add_one_way_quad.argtypes = [vp, fp, fp, fp, fp, C.c_int32, pm]
    L.rt_scene_add_cuboid.argtypes = [vp, fp, C.c_float, C.c_float, C.c_float, pm]
    L.rt_scene_add_mesh.argtypes = [vp, fp, C.c_int32, pm]
    L.rt_scene_add_obj_mesh.argtypes = [vp, vp, pm]
    L.rt_scene_builder_num_objects.argtypes = [vp]
    L.rt_obj_load.argtypes = [C.c_char_p, C.POINTER(vp)]
    L.rt_obj_destroy.argtypes = [vp]
    L.rt_obj_destroy.restype = None
    L.rt_obj_enlarge.argtypes = [vp, C.c_float]
    L.rt_obj_rotate.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    L.rt_obj_translate.argtypes = [vp, C.c_float, C.c_float, C.c_float]
    for n in ("rt_obj_enlarge", "rt_obj_rotate", "rt_obj_translate"):
        getattr(L, n).restype = None
    for n in ("rt_obj_num_vertices", "rt_obj_num_faces", "rt_obj_num_triangles"):
        getattr(L, n).argtypes = [vp]
    L.rt_obj_face_arity.argtypes = [vp, C.c_int32]
    L.rt_obj_get_face.argtypes = [vp, C.c_int32, C.POINTER(C.c_int32)]
    L.rt_obj_get_face.restype = None
    L.rt_obj_from_arrays.argtypes = [fp, C.c_int32, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.c_int32, C.POINTER(vp)]
    L.rt_obj_get_vertices.argtypes = [vp, fp]
    L.rt_obj_get_vertices.restype = None
    L.rt_obj_get_triangles.argtypes = [vp, fp]
    L.rt_camera_default.argtypes = [C.c_int32, C.c_int32, C.POINTER(rt_camera)]
    L.rt_camera_default.restype = None
    L.rt_camera_make.argtypes = [C.c_int32, C.c_int32, fp, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.POINTER(rt_camera)]
    L.rt_camera_make.restype = None
    L.rt_ctx_create.argtypes = [C.c_int32, C.POINTER(vp)]
    L.rt_ctx_destroy.argtypes = [vp]
    L.rt_ctx_destroy.restype = None
    L.rt_last_error.argtypes = [vp]
    L.rt_last_error.restype = C.c_char_p
    L.rt_scene_commit.argtypes = [vp, vp, C.POINTER(vp)]
    L.rt_scene_destroy.argtypes = [vp]
    L.rt_scene_destroy.restype = None
    L.rt_scene_get_info.argtypes = [vp, C.POINTER(rt_scene_info)]
    L.rt_render.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.c_int32, C.POINTER(C.c_int32), fp]
    L.rt_render_frames.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.POINTER(C.c_int32), C.c_int32, C.POINTER(C.c_int32), fp]
    L.rt_render_device.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.c_int32, C.c_int32,
                                   C.POINTER(rt_tile_spec), vp, vp, vp]
    L.rt_render_device_batch.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.POINTER(C.c_int32), C.c_int32, C.c_int32,
                                         C.POINTER(rt_tile_spec), vp, vp]
    L.rt_tile_owned_rows.argtypes = [C.POINTER(rt_tile_spec), C.c_int32]
    if hasattr(L, "rt_frame_submit"):            # (absent from development builds of older revisions)
        L.rt_frame_submit.argtypes = [vp, vp, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.c_int32, C.POINTER(rt_tile_spec)]
        L.rt_frame_collect.argtypes = [vp, C.c_int32, vp, vp]
        L.rt_frames_pending.argtypes = [vp]
        L.rt_frame_wait.argtypes = [vp]
        L.rt_frame_depth.argtypes = [vp, C.c_int32]
        L.rt_frame_collect_host.argtypes = [vp, C.POINTER(C.c_int32), fp]
    u32p = C.POINTER(C.c_uint32)
    L.rt_tile_costs.argtypes = [vp, u32p, u32p, u32p, C.c_int32, C.POINTER(C.c_int32)]
    L.rt_partition_tiles.argtypes = [u32p, C.c_int32, C.c_int32, C.c_int32, C.POINTER(C.c_int32)]
    L.rt_tiles_copy_device.argtypes = [vp, vp, vp, C.c_int32, C.c_int32, u32p, C.c_int32, C.c_int32, vp]
    L.rt_max_batch_frames.argtypes = [vp, C.c_int32, C.c_int32]
    L.rt_peer_access.argtypes = [vp, vp]
    L.rt_last_kernel_ms.argtypes = [vp, C.POINTER(C.c_float)]
    L.rt_ctx_synchronize.argtypes = [vp]
    L.rt_render_multi.argtypes = [C.POINTER(rt_rank), C.c_int32, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.POINTER(C.c_int32), C.c_int32,
                                  C.POINTER(C.c_int32), fp]
    L.rt_render_multi_device.argtypes = [C.POINTER(rt_rank), C.c_int32, C.POINTER(rt_camera), C.POINTER(rt_render_settings), C.POINTER(C.c_int32), C.c_int32,
                                         C.c_int32, C.c_int32, vp, vp]
    L.rt_gather.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp, C.POINTER(rt_tile_spec), vp]
    L.rt_to_rgba8_device.argtypes = [vp, vp, C.c_int32, C.c_int32, vp, vp]
    L.rt_debug_flatten.argtypes = [vp, C.POINTER(rt_flat_view)]
    L.rt_debug_read_stats.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rt_debug_eval.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.c_int32]
    if hasattr(L, "rt_debug_exhaustive"):        # (absent from development builds of older revisions: tools/build_variants.py name@REV)
        L.rt_debug_exhaustive.argtypes = [vp, C.POINTER(C.c_uint64)]
    L.rt_version.restype = C.c_char_p
    _lib = L
    return L


class RayTracerError(RuntimeError):
    """std::runtime_error of the reference (check_cuda_error src/utils.cu:5-10, read_file src/obj_read.cu:10)."""


class PipelineFullError(RayTracerError):
    """rt_frame_submit: RT_PIPELINE_DEPTH frames are in flight (RT_ERR_BUSY)"""


class UnsupportedMeshError(ValueError):
    """std::logic_error("Only triangle or quad meshes are supported.") src/main.cu:141"""


def _fp(a):
    arr = np.ascontiguousarray(a, dtype=np.float32)
    return arr, arr.ctypes.data_as(C.POINTER(C.c_float))


class Material:
    """Material + Texture factories, reference src/material.cu:21-51, :157-185."""

    def __init__(self, c_struct):
        self.c = c_struct

    @staticmethod
    def create_standard(colour, smoothness):
        m = rt_material()
        lib().rt_material_standard(C.byref(m), _fp(colour)[1], C.c_float(smoothness))
        return Material(m)

    @staticmethod
    def create_checkerboard(light, dark, num_squares, smoothness):
        m = rt_material()
        lib().rt_material_checkerboard(C.byref(m), _fp(light)[1], _fp(dark)[1], int(num_squares), C.c_float(smoothness))
        return Material(m)

    @staticmethod
    def create_gradient(smoothness):
        m = rt_material()
        lib().rt_material_gradient(C.byref(m), C.c_float(smoothness))
        return Material(m)

    @staticmethod
    def create_emissive(colour, strength):
        m = rt_material()
        lib().rt_material_emissive(C.byref(m), _fp(colour)[1], C.c_float(strength))
        return Material(m)

    @staticmethod
    def create_refractive(colour, n):
        m = rt_material()
        lib().rt_material_refractive(C.byref(m), _fp(colour)[1], C.c_float(n))
        return Material(m)

    @staticmethod
    def create_image(rgb, smoothness):
        """rgb: [height, width, 3] float32 texels (Texture::create_image src/material.cu:42-51);
        the scene builder copies them when the object is added"""
        arr, p = _fp(np.asarray(rgb, np.float32))
        m = rt_material()
        lib().rt_material_image(C.byref(m), arr.shape[1], arr.shape[0], p, C.c_float(smoothness))
        mat = Material(m)
        mat._keep = arr
        return mat

    @staticmethod
    def from_desc(desc):
        kind = desc[0]
        if kind == "standard":
            return Material.create_standard(desc[1], desc[2])
        if kind == "emissive":
            return Material.create_emissive(desc[1], desc[2])
        if kind == "checkerboard":
            return Material.create_checkerboard(desc[1], desc[2], desc[3], desc[4])
        if kind == "gradient":
            return Material.create_gradient(desc[1])
        if kind == "refractive":
            return Material.create_refractive(desc[1], desc[2])
        if kind == "image":
            return Material.create_image(desc[1], desc[2])
        raise ValueError(kind)


def load_image_texture(parsed_textures_path, name):
    """ImageTexture src/main.cu:40-91: entry `name` of a baked texture file -> [h, w, 3] float32"""
    w, h = C.c_int32(), C.c_int32()
    ptr = C.POINTER(C.c_float)()
    st = lib().rt_image_texture_load(os.fsencode(parsed_textures_path), name.encode(), C.byref(w), C.byref(h), C.byref(ptr))
    if st == RT_ERR_IO:
        raise RayTracerError("Could not find file to open.")
    if st != RT_OK:
        raise RayTracerError("Image file not found.\n")
    try:
        return np.ctypeslib.as_array(ptr, shape=(h.value, w.value, 3)).copy()
    finally:
        lib().rt_image_texture_free(ptr)


class ObjFileMesh:
    """reference src/obj_read.cu:47-147"""

    def __init__(self, filename, _handle=None):
        if _handle is not None:
            self._h = _handle
            return
        h = C.c_void_p()
        st = lib().rt_obj_load(os.fsencode(filename), C.byref(h))
        if st == RT_ERR_IO:
            raise RayTracerError("Could not find file to open.")
        if st != RT_OK:
            raise RayTracerError("could not parse %s" % filename)
        self._h = h

    @staticmethod
    def from_arrays(vertices, faces):
        """vertices [n,3] float32; faces: list of 0-based index lists"""
        v, vp_ = _fp(np.asarray(vertices, np.float32).reshape(-1, 3))
        flat = np.ascontiguousarray([i for f in faces for i in f], dtype=np.int32)
        arity = np.ascontiguousarray([len(f) for f in faces], dtype=np.int32)
        h = C.c_void_p()
        st = lib().rt_obj_from_arrays(vp_, v.shape[0], flat.ctypes.data_as(C.POINTER(C.c_int32)),
                                      arity.ctypes.data_as(C.POINTER(C.c_int32)), len(faces), C.byref(h))
        if st != RT_OK:
            raise ValueError("bad mesh arrays")
        return ObjFileMesh(None, _handle=h)

    def faces(self):
        out = []
        for i, a in enumerate(self.face_arities()):
            buf = (C.c_int32 * a)()
            lib().rt_obj_get_face(self._h, i, buf)
            out.append(list(buf))
        return out

    def enlarge(self, scale_fact):
        lib().rt_obj_enlarge(self._h, C.c_float(scale_fact))

    def rotate(self, x_angle, y_angle, z_angle):
        lib().rt_obj_rotate(self._h, C.c_float(x_angle), C.c_float(y_angle), C.c_float(z_angle))

    def translate(self, offset_x, offset_y, offset_z):
        lib().rt_obj_translate(self._h, C.c_float(offset_x), C.c_float(offset_y), C.c_float(offset_z))

    @property
    def num_vertices(self):
        return lib().rt_obj_num_vertices(self._h)

    @property
    def num_faces(self):
        return lib().rt_obj_num_faces(self._h)

    def face_arities(self):
        return [lib().rt_obj_face_arity(self._h, i) for i in range(self.num_faces)]

    def vertices(self):
        out = np.empty((self.num_vertices, 3), np.float32)
        lib().rt_obj_get_vertices(self._h, out.ctypes.data_as(C.POINTER(C.c_float)))
        return out

    def triangles(self):
        n = lib().rt_obj_num_triangles(self._h)
        if n < 0:
            raise UnsupportedMeshError("Only triangle or quad meshes are supported.\n")
        out = np.empty((n, 9), np.float32)
        st = lib().rt_obj_get_triangles(self._h, out.ctypes.data_as(C.POINTER(C.c_float)))
        if st != RT_OK:
            raise ValueError("face references a missing vertex")
        return out

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().rt_obj_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass


class SceneObjects:
    """The object list of a scene: reference SceneObjects src/main.cu:94-296 with the
    Object::create_* factories of src/objects.cu:845-906 as methods."""

    def __init__(self, description=None, models_dir=None):
        h = C.c_void_p()
        if lib().rt_scene_builder_create(C.byref(h)) != RT_OK:
            raise MemoryError()
        self._h = h
        self.models_dir = models_dir or scenes.models_dir()
        if description:
            self.add_description(description)

    def _check(self, st):
        if st == RT_OK:
            return
        msg = lib().rt_scene_builder_error(self._h).decode()
        if st == RT_ERR_UNSUPPORTED and "triangle or quad" in msg:
            raise UnsupportedMeshError(msg)
        if st == RT_ERR_UNSUPPORTED:
            raise NotImplementedError(msg)
        raise ValueError(msg)

    def create_sphere(self, center, radius, mat):
        self._check(lib().rt_scene_add_sphere(self._h, _fp(center)[1], C.c_float(radius), C.byref(mat.c)))

    def create_triangle(self, p1, p2, p3, mat, uv=None):
        if uv is None:
            self._check(lib().rt_scene_add_triangle(self._h, _fp(p1)[1], _fp(p2)[1], _fp(p3)[1], C.byref(mat.c)))
        else:
            pts = np.concatenate([np.asarray(p, np.float32).reshape(3) for p in (p1, p2, p3)])
            self._check(lib().rt_scene_add_triangle_uv(self._h, _fp(pts)[1], _fp(np.asarray(uv).reshape(6))[1], C.byref(mat.c)))

    def create_quad(self, p1, p2, p3, p4, mat):
        self._check(lib().rt_scene_add_quad(self._h, _fp(p1)[1], _fp(p2)[1], _fp(p3)[1], _fp(p4)[1], C.byref(mat.c)))

    def create_one_way_quad(self, p1, p2, p3, p4, invert_normal, mat):
        self._check(lib().rt_scene_add_one_way_quad(self._h, _fp(p1)[1], _fp(p2)[1], _fp(p3)[1], _fp(p4)[1], int(bool(invert_normal)), C.byref(mat.c)))

    def create_cuboid(self, tl_near_pos, width, height, depth, mat):
        self._check(lib().rt_scene_add_cuboid(self._h, _fp(tl_near_pos)[1], C.c_float(width), C.c_float(height), C.c_float(depth), C.byref(mat.c)))

    def create_mesh(self, mesh, mat):
        """mesh: an ObjFileMesh (src/main.cu:127-148) or an array of triangles [n, 9]"""
        if isinstance(mesh, ObjFileMesh):
            self._check(lib().rt_scene_add_obj_mesh(self._h, mesh._h, C.byref(mat.c)))
        else:
            arr, p = _fp(np.asarray(mesh, np.float32).reshape(-1, 9))
            self._check(lib().rt_scene_add_mesh(self._h, p, arr.shape[0], C.byref(mat.c)))

    def add_description(self, description):
        for o in description:
            kind, mat = o[0], Material.from_desc(o[-1])
            if kind == "sphere":
                self.create_sphere(o[1], o[2], mat)
            elif kind == "triangle":
                self.create_triangle(o[1], o[2], o[3], mat)
            elif kind == "triangle_uv":
                p = np.asarray(o[1], np.float32).reshape(3, 3)
                self.create_triangle(p[0], p[1], p[2], mat, uv=o[2])
            elif kind == "quad":
                self.create_quad(o[1], o[2], o[3], o[4], mat)
            elif kind == "one_way_quad":
                self.create_one_way_quad(o[1], o[2], o[3], o[4], o[5], mat)
            elif kind == "cuboid":
                self.create_cuboid(o[1], o[2], o[3], o[4], mat)
            elif kind == "mesh":
                self.create_mesh(o[1], mat)
            elif kind == "obj":
                path = o[1] if os.path.isabs(o[1]) else os.path.join(self.models_dir, o[1])
                m = ObjFileMesh(path)
                for t in o[2]:
                    getattr(m, t[0])(*t[1:])
                self.create_mesh(m, mat)
            else:
                raise ValueError(kind)

    @property
    def num_objects(self):
        return lib().rt_scene_builder_num_objects(self._h)

    def debug_flatten(self):
        """The flattened device layout as numpy arrays (tests only)."""
        v = rt_flat_view()
        self._check(lib().rt_debug_flatten(self._h, C.byref(v)))
        blob = np.ctypeslib.as_array(v.blob, shape=(v.blob_f4, 4)).copy() if v.blob_f4 else np.zeros((0, 4), np.float32)
        raw = C.string_at(v.objects, v.num_objects * v.object_stride) if v.num_objects else b""
        objs = np.frombuffer(raw, dtype=np.dtype([("type", "<i4"), ("prim_start", "<i4"), ("need_uv", "<i4"), ("root_ref", "<u4"), ("v", "<f4", (8,))]))
        uv = np.ctypeslib.as_array(v.tri_uv, shape=(v.num_triangles, 6)).copy() if v.tri_uv else None
        return {"blob": blob, "off_nodes": v.off_nodes, "off_tris": v.off_tris, "off_objlds": v.off_objlds,
                "off_meshes": v.off_meshes, "num_meshes": v.num_meshes, "stack_entries": v.stack_entries,
                "objects": objs, "tri_uv": uv, "num_triangles": v.num_triangles, "num_nodes": v.num_nodes,
                "has_mesh": bool(v.has_mesh)}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().rt_scene_builder_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass


class Camera:
    """reference Camera src/camera.cu:32-108; ``assign_constant_mem`` becomes :attr:`c` (the
    48-byte DeviceCamData plus the image size), passed to render calls."""

    def __init__(self, width, height, pos=None, fov=None, focal_len=None, rot=(0.0, 0.0, 0.0), floats=None):
        self.c = rt_camera()
        if floats is not None:      # the 12 floats verbatim (fixtures)
            f = np.asarray(floats, np.float32).reshape(12)
            self.c.cam_pos[:] = f[0:3].tolist()
            self.c.tl_pixel_pos[:] = f[3:6].tolist()
            self.c.delta_u[:] = f[6:9].tolist()
            self.c.delta_v[:] = f[9:12].tolist()
            self.c.width, self.c.height = int(width), int(height)
        elif pos is None and fov is None and focal_len is None and tuple(rot) == (0.0, 0.0, 0.0):
            lib().rt_camera_default(int(width), int(height), C.byref(self.c))
        else:
            pi = np.float32(3.141592653589793)
            fov = np.float32(60) * (pi / np.float32(180)) if fov is None else fov
            lib().rt_camera_make(int(width), int(height), _fp(pos or (0, 0, 0))[1], C.c_float(fov),
                                 C.c_float(0.1 if focal_len is None else focal_len),
                                 C.c_float(rot[0]), C.c_float(rot[1]), C.c_float(rot[2]), C.byref(self.c))

    @property
    def width(self):
        return self.c.width

    @property
    def height(self):
        return self.c.height

    def floats(self):
        return np.array(list(self.c.cam_pos) + list(self.c.tl_pixel_pos) + list(self.c.delta_u) + list(self.c.delta_v), np.float32)


class RenderData:
    """reference RenderData src/raytracer.cu:4-12 (defaults of RenderSettings src/main.cu:318-330)"""

    def __init__(self, rays_per_pixel=100, reflection_limit=5, antialias=True, sky_colour=(0.0, 0.0, 0.0)):
        self.c = rt_render_settings(int(rays_per_pixel), int(reflection_limit), int(bool(antialias)), (C.c_float * 3)(*[float(x) for x in sky_colour]))


class VariableRenderData:
    """reference VariableRenderData src/dispatch.cu:111-115"""

    def __init__(self, width, height):
        self.frame_num = 0
        self.previous_render = np.zeros((height, width, 3), np.float32)


class Context:
    """One per GPU.  Raises when there is no GPU (the product has no CPU path)."""

    def __init__(self, device=0):
        h = C.c_void_p()
        st = lib().rt_ctx_create(int(device), C.byref(h))
        if st == RT_ERR_NO_DEVICE:
            raise RayTracerError("Error from HIP (creating context): no usable GPU; ray-tracer_amd has no CPU fallback")
        if st != RT_OK:
            raise RayTracerError("Error from HIP (creating context): status %d" % st)
        self._h = h
        self.device = device

    def _check(self, st):
        if st != RT_OK:
            msg = lib().rt_last_error(self._h).decode()
            if st == RT_ERR_UNSUPPORTED:
                raise NotImplementedError(msg)
            if st == RT_ERR_INVALID:
                raise ValueError(msg)
            if st == RT_ERR_BUSY:
                raise PipelineFullError(msg)
            raise RayTracerError(msg)

    def last_error(self):
        """the context's most recent error message (rt_last_error; empty when there was none)"""
        m = lib().rt_last_error(self._h)
        return m.decode() if m else ""

    def commit(self, scene_objects):
        return Scene(self, scene_objects)

    def last_kernel_ms(self):
        ms = C.c_float()
        self._check(lib().rt_last_kernel_ms(self._h, C.byref(ms)))
        return ms.value

    def synchronize(self):
        """waits for this context's most recent launch"""
        self._check(lib().rt_ctx_synchronize(self._h))

    def tile_costs(self, with_peaks=False):
        """(tile indices in the image, costs[, peak pixel costs]) of the current view's tiles as its first launch measured
        them (rt_tile_costs; waits for that launch)"""
        n = C.c_int32()
        self._check(lib().rt_tile_costs(self._h, None, None, None, 0, C.byref(n)))
        ids, costs, peaks = np.empty(n.value, np.uint32), np.empty(n.value, np.uint32), np.empty(n.value, np.uint32)
        u32p = C.POINTER(C.c_uint32)
        self._check(lib().rt_tile_costs(self._h, ids.ctypes.data_as(u32p), costs.ctypes.data_as(u32p), peaks.ctypes.data_as(u32p), n.value, C.byref(n)))
        return (ids, costs, peaks) if with_peaks else (ids, costs)

    def max_batch_frames(self, width, height):
        """frames the multi-frame entry points put into one launch for this image size (rt_max_batch_frames)"""
        return lib().rt_max_batch_frames(self._h, int(width), int(height))

    def peer_access(self, other):
        """1: copies between the two contexts' GPUs go direct (xGMI), 0: staged by the runtime (rt_peer_access)"""
        return lib().rt_peer_access(self._h, other._h)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().rt_ctx_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass


class Scene:
    """A committed (uploaded) scene: replaces create_gpu_struct src/main.cu:290-295 +
    allocate_constant_mem src/dispatch.cu:104-108."""

    def __init__(self, ctx, scene_objects):
        self.ctx = ctx
        h = C.c_void_p()
        ctx._check(lib().rt_scene_commit(ctx._h, scene_objects._h, C.byref(h)))
        self._h = h

    def info(self):
        i = rt_scene_info()
        self.ctx._check(lib().rt_scene_get_info(self._h, C.byref(i)))
        return {n: getattr(i, n) for n, _ in i._fields_}

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                lib().rt_scene_destroy(self._h)
                self._h = None
        except Exception:      # interpreter shutdown: module globals may already be gone
            pass


def render(ctx, scene, camera, render_data, data, current_time_ms):
    """reference render(VariableRenderData*, int) src/dispatch.cu:156-163: reads
    data.previous_render, overwrites it with the new progressive average, increments
    data.frame_num."""
    fn = C.c_int32(data.frame_num)
    buf = data.previous_render
    assert buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"]
    ctx._check(lib().rt_render(ctx._h, scene._h, C.byref(camera.c), C.byref(render_data.c), int(current_time_ms),
                               C.byref(fn), buf.ctypes.data_as(C.POINTER(C.c_float))))
    data.frame_num = fn.value
    return buf


def render_frames(ctx, scene, camera, render_data, data, times_ms):
    """len(times_ms) consecutive passes of the reference's main loop body (src/main.cu:421-424) in
    one call: the same image as that many render() calls, rendered by multi-frame launches."""
    fn = C.c_int32(data.frame_num)
    buf = data.previous_render
    assert buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"]
    t = (C.c_int32 * len(times_ms))(*[int(x) for x in times_ms])
    ctx._check(lib().rt_render_frames(ctx._h, scene._h, C.byref(camera.c), C.byref(render_data.c), t, len(times_ms),
                                      C.byref(fn), buf.ctypes.data_as(C.POINTER(C.c_float))))
    data.frame_num = fn.value
    return buf


def _tile_spec(band_rows, band_first, band_stride, compact, tile_list, tile_cost, tile_peak=None):
    """rt_tile_spec + the arrays it points at (keep the second value alive for the duration of the call)"""
    ts = rt_tile_spec(int(band_rows), int(band_first), int(band_stride), int(bool(compact)))
    keep = None
    if tile_list is not None:
        ids = np.ascontiguousarray(tile_list, dtype=np.uint32)
        # (a NULL list pointer means "bands": an empty list still needs an address)
        backing = ids if ids.size else np.zeros(1, np.uint32)
        ts.tile_list = backing.ctypes.data_as(C.POINTER(C.c_uint32))
        ts.num_tiles = int(ids.size)
        cost = None
        if tile_cost is not None:
            cost = np.ascontiguousarray(tile_cost, dtype=np.uint32)
            assert cost.size == ids.size
            if cost.size:
                ts.tile_cost = cost.ctypes.data_as(C.POINTER(C.c_uint32))
        peak = None
        if tile_peak is not None and cost is not None:
            peak = np.ascontiguousarray(tile_peak, dtype=np.uint32)
            assert peak.size == ids.size
            if peak.size:
                ts.tile_peak = peak.ctypes.data_as(C.POINTER(C.c_uint32))
        keep = (ids, backing, cost, peak)
    return ts, keep


def render_device(ctx, scene, camera, render_data, time_ms, frame_num, d_out, d_prev=None,
                  band_rows=8, band_first=0, band_stride=1, compact=False, stream=None, tile_list=None, tile_cost=None, tile_peak=None):
    """Device-buffer form: d_out / d_prev are device pointers (ints, e.g. torch.Tensor.data_ptr()).  tile_list: the
    8x8 tiles to render (indices ty * ceil(W / 8) + tx) instead of bands."""
    ts, keep = _tile_spec(band_rows, band_first, band_stride, compact, tile_list, tile_cost, tile_peak)
    ctx._check(lib().rt_render_device(ctx._h, scene._h, C.byref(camera.c), C.byref(render_data.c), int(time_ms), int(frame_num),
                                      C.byref(ts), C.c_void_p(d_prev or 0), C.c_void_p(d_out), C.c_void_p(stream or 0)))
    del keep


def render_device_batch(ctx, scene, camera, render_data, times_ms, frame_num, d_frame,
                        band_rows=8, band_first=0, band_stride=1, compact=False, stream=None, tile_list=None, tile_cost=None, tile_peak=None):
    """len(times_ms) consecutive progressive frames in ONE launch, accumulated in place in the device
    buffer d_frame (bit-identical to that many render_device calls; see rt_render_device_batch)."""
    ts, keep = _tile_spec(band_rows, band_first, band_stride, compact, tile_list, tile_cost, tile_peak)
    t = (C.c_int32 * len(times_ms))(*[int(x) for x in times_ms])
    ctx._check(lib().rt_render_device_batch(ctx._h, scene._h, C.byref(camera.c), C.byref(render_data.c), t, len(times_ms), int(frame_num),
                                            C.byref(ts), C.c_void_p(d_frame), C.c_void_p(stream or 0)))
    del keep


PIPELINE_DEPTH = 8          # RT_PIPELINE_DEPTH (include/rt_amd.h): at most
PIPELINE_DEFAULT_DEPTH = 4


def frame_depth(ctx, depth):
    """how many frames the caller keeps in flight: each is launched on 1 / depth of the CUs (rt_frame_depth)"""
    ctx._check(lib().rt_frame_depth(ctx._h, int(depth)))



def frame_submit(ctx, scene, camera, render_data, time_ms,
                 band_rows=8, band_first=0, band_stride=1, compact=False, tile_list=None, tile_cost=None, tile_peak=None):
    """Queue one frame seeded with time_ms on a stream of the context's own; up to PIPELINE_DEPTH may be submitted and not
    collected (rt_frame_submit).  Frames in flight overlap on the GPU."""
    ts, keep = _tile_spec(band_rows, band_first, band_stride, compact, tile_list, tile_cost, tile_peak)
    ctx._check(lib().rt_frame_submit(ctx._h, scene._h, C.byref(camera.c), C.byref(render_data.c), int(time_ms), C.byref(ts)))
    del keep


def frame_collect(ctx, frame_num, d_frame, stream=None):
    """Fold the oldest submitted frame into the device buffer d_frame as progressive frame frame_num, asynchronously on
    `stream` (rt_frame_collect).  d_frame None: discard the frame."""
    ctx._check(lib().rt_frame_collect(ctx._h, int(frame_num), C.c_void_p(d_frame or 0), C.c_void_p(stream or 0)))


def frame_collect_host(ctx, data):
    """the oldest submitted (whole) frame into data.previous_render, data.frame_num += 1 (rt_frame_collect_host); data None: discard"""
    if data is None:
        ctx._check(lib().rt_frame_collect_host(ctx._h, C.byref(C.c_int32(0)), None))
        return
    fn = C.c_int32(int(data.frame_num))
    buf = data.previous_render
    assert buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"]
    ctx._check(lib().rt_frame_collect_host(ctx._h, C.byref(fn), buf.ctypes.data_as(C.POINTER(C.c_float))))
    data.frame_num = int(fn.value)


def frame_wait(ctx):
    """block until the frame collected last is in its d_frame (rt_frame_wait)"""
    ctx._check(lib().rt_frame_wait(ctx._h))


def frames_pending(ctx):
    return int(lib().rt_frames_pending(ctx._h))


def partition_tiles(width, height, n_ranks, cost=None):
    """owner[ty * tiles_x + tx] = rank (rt_partition_tiles): interleaved without costs, longest-processing-time-first
    with them.  Returns an int32 array over the image's 8x8 tiles."""
    tiles_x, tiles_y = (int(width) + 7) // 8, (int(height) + 7) // 8
    owner = np.empty(tiles_x * tiles_y, np.int32)
    cp = None
    if cost is not None:
        cost = np.ascontiguousarray(cost, dtype=np.uint32)
        assert cost.size == owner.size
        cp = cost.ctypes.data_as(C.POINTER(C.c_uint32))
    st = lib().rt_partition_tiles(cp, tiles_x, tiles_y, int(n_ranks), owner.ctypes.data_as(C.POINTER(C.c_int32)))
    if st != RT_OK:
        raise ValueError("rt_partition_tiles: bad argument")
    return owner


def tiles_copy_device(ctx, d_compact, d_frame, width, height, tile_list, to_frame=True, stream=None):
    """compact tile-list image <-> full frame on ctx's GPU (rt_tiles_copy_device)"""
    ids = np.ascontiguousarray(tile_list, dtype=np.uint32)
    backing = ids if ids.size else np.zeros(1, np.uint32)
    ctx._check(lib().rt_tiles_copy_device(ctx._h, C.c_void_p(d_compact), C.c_void_p(d_frame), int(width), int(height),
                                          backing.ctypes.data_as(C.POINTER(C.c_uint32)), int(ids.size), int(bool(to_frame)), C.c_void_p(stream or 0)))


def _ranks(ctxs, scenes):
    arr = (rt_rank * len(ctxs))()
    for i, (c, s) in enumerate(zip(ctxs, scenes)):
        arr[i].ctx, arr[i].scene = c._h, s._h
    return arr


def render_multi(ctxs, scenes, camera, render_data, data, times_ms):
    """render() for a node (rt_render_multi): rank i = (ctxs[i], scenes[i]) renders the bands b % n == i on its
    own GPU; the image lands in data.previous_render like render_frames on one GPU."""
    fn = C.c_int32(data.frame_num)
    buf = data.previous_render
    assert buf.dtype == np.float32 and buf.flags["C_CONTIGUOUS"]
    t = (C.c_int32 * len(times_ms))(*[int(x) for x in times_ms])
    ctxs[0]._check(lib().rt_render_multi(_ranks(ctxs, scenes), len(ctxs), C.byref(camera.c), C.byref(render_data.c), t, len(times_ms),
                                         C.byref(fn), buf.ctypes.data_as(C.POINTER(C.c_float))))
    data.frame_num = fn.value
    return buf


def render_multi_device(ctxs, scenes, camera, render_data, times_ms, frame_num, d_frame, band_rows=0, stream=None):
    """device-buffer form (rt_render_multi_device): d_frame is a full frame on ctxs[0]'s GPU, updated in place.
    band_rows = 0: cost-balanced tile lists (the first call of a view measures the tiles); > 0: static bands"""
    t = (C.c_int32 * len(times_ms))(*[int(x) for x in times_ms])
    ctxs[0]._check(lib().rt_render_multi_device(_ranks(ctxs, scenes), len(ctxs), C.byref(camera.c), C.byref(render_data.c), t, len(times_ms),
                                                int(frame_num), int(band_rows), C.c_void_p(d_frame), C.c_void_p(stream or 0)))


def gather(root, d_frame, width, height, src, d_bands, band_rows=8, band_first=0, band_stride=1, stream=None, tile_list=None):
    """the exchange step alone (rt_gather): src's compact buffer (bands, or the tiles of tile_list) -> the full frame on root's GPU"""
    ts, keep = _tile_spec(band_rows, band_first, band_stride, True, tile_list, None)
    root._check(lib().rt_gather(root._h, C.c_void_p(d_frame), int(width), int(height), src._h, C.c_void_p(d_bands), C.byref(ts), C.c_void_p(stream or 0)))
    del keep


def debug_eval(ctx, op, bits):
    """device-side evaluation of a math / RNG header function on uint32 bit patterns (tests)"""
    a = np.ascontiguousarray(bits, dtype=np.uint32)
    out = np.empty_like(a)
    ctx._check(lib().rt_debug_eval(ctx._h, int(op), a.ctypes.data_as(C.POINTER(C.c_uint32)), out.ctypes.data_as(C.POINTER(C.c_uint32)), a.size))
    return out


def debug_exhaustive(ctx):
    """(differing, in range) for the device code's short reciprocal, then for its short square root, over all 2^32 inputs"""
    out = (C.c_uint64 * 4)()
    ctx._check(lib().rt_debug_exhaustive(ctx._h, out))
    return tuple(int(x) for x in out)


def tile_owned_rows(height, band_rows=8, band_first=0, band_stride=1):
    ts = rt_tile_spec(int(band_rows), int(band_first), int(band_stride), 0)
    return lib().rt_tile_owned_rows(C.byref(ts), int(height))


def save_png(path, image):
    """Writes a frame as an 8-bit RGB PNG (the format of the reference's images/*.png).  `image`
    is [H, W, 3|4] uint8, or a float frame, which is converted like the reference's display path
    (src/main.cu:343-371: int(px * 255), clamped)."""
    import struct
    import zlib
    img = np.asarray(image)
    if img.dtype != np.uint8:
        img = np.clip((img.astype(np.float32) * np.float32(255)).astype(np.int64), 0, 255).astype(np.uint8)
    img = np.ascontiguousarray(img[:, :, :3])
    h, w = img.shape[:2]
    raw = np.concatenate([np.zeros((h, 1), np.uint8), img.reshape(h, w * 3)], axis=1).tobytes()

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0)) + chunk(b"IDAT", zlib.compress(raw, 6)) + chunk(b"IEND", b""))


def to_rgba8_device(ctx, d_rgb, width, height, d_rgba, stream=None):
    """float -> RGBA8 of src/main.cu:343-371 on the device."""
    ctx._check(lib().rt_to_rgba8_device(ctx._h, C.c_void_p(d_rgb), int(width), int(height), C.c_void_p(d_rgba), C.c_void_p(stream or 0)))
