"""Turns the reference's model assets into array fixtures (run once, in the build container).

Reads /root/reference/models/*.obj through the product's own loader (rt_obj_load, the
restatement of reference src/obj_read.cu:47-147) and stores what the loader extracts —
float32 vertices and 0-based face index lists — as ray-tracer_amd/models/<name>.npz.
Only this derived data travels with the repo; `ray-tracer_amd.scenes.models_dir()` writes
.obj text from it at run time so scene descriptions can keep naming `cube.obj` /
`low_poly_monkey.obj` and the .obj loader stays on the path.
"""
import importlib
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
rt = importlib.import_module("ray-tracer_amd")

SRC = "/root/reference/models"
DST = os.path.join(ROOT, "ray-tracer_amd", "models")

for name in ("cube", "low_poly_monkey"):
    m = rt.ObjFileMesh(os.path.join(SRC, name + ".obj"))
    faces = m.faces()
    np.savez_compressed(os.path.join(DST, name + ".npz"),
                        vertices=m.vertices(),
                        face_indices=np.array([i for f in faces for i in f], np.int32),
                        face_arity=np.array([len(f) for f in faces], np.int32))
    print(name, m.num_vertices, "vertices", m.num_faces, "faces")
