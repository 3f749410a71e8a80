/*
 * rt_host.h — host-side scene model shared by rt_host.cpp (builders, loaders, flattening) and
 * rt_capi.cpp (upload + launch).  Internal to libraytracer_amd.so.
 */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <string>
#include <vector>

#include "../../include/rt_amd.h"
#include "rt_device_scene.h"

struct HostTri {
    float p0[3], s1[3], s2[3], n[3];
    float uv[6];
};

struct HostObject {
    int type = 0;
    rt_material mat{};
    float v[8] = {0, 0, 0, 0, 0, 0, 0, 0};   /* see rt_object */
    std::vector<HostTri> tris;           /* top-level shapes: 1 / 2 / 12; meshes: leaf order */
    std::vector<rt_node> nodes;          /* meshes: child refs relative to this object */
    std::vector<float> texels;           /* IMAGE texture: a copy of the caller's rgb data */
    uint32_t root_ref = RT_REF_EMPTY_LEAF;
    int stack_need = 0;
};

struct FlatScene;

struct rt_scene_builder {
    std::vector<HostObject> objs;
    std::string err;
    FlatScene *debug_flat = nullptr;     /* owned; rt_debug_flatten */
    ~rt_scene_builder();
};

struct rt_obj {
    std::vector<float> vx, vy, vz;               /* vertex_mat rows (reference src/obj_read.cu:49) */
    std::vector<std::vector<int>> faces;         /* 0-based vertex indices per face */
};

/* flattened scene, ready to upload */
struct FlatScene {
    std::vector<rt_f4> blob;
    int off_nodes = 0, off_tris = 0, off_objlds = 0, off_meshes = 0, off_objtab = 0;
    int num_meshes = 0;
    int stack_entries = 1;                       /* per-lane traversal stack depth this scene needs */
    std::vector<rt_object> objects;
    std::vector<float> tri_uv;                   /* empty unless some material needs UVs */
    std::vector<float> tex_data;                 /* IMAGE texels of all objects, back to back */
    int num_tris = 0, num_nodes = 0;
    bool has_mesh = false;
};

/* returns "" or an error message */
std::string rt_flatten(const rt_scene_builder &b, FlatScene &out);

#endif
