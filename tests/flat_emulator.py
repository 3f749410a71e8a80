"""A float32 Python emulation of the HIP kernel's closest-hit search over the FLATTENED scene
layout (ray-tracer_amd/csrc/rt_device_scene.h), used by the CPU-only tests to validate the
host flattener and the traversal design (current node in a register, deferred sibling +
entry distance on a stack, boxes stored in the parent) without a GPU.  Test code only.
"""
import numpy as np

f32 = np.float32
INF = f32(1073741824.0)
EPS = f32(0.000001)
LEAF = 0x80000000
CHAIN = 0x40000000


def _fmin(a, b):
    if np.isnan(b):
        return a
    if np.isnan(a):
        return b
    return a if a < b else b


def _fmax(a, b):
    if np.isnan(b):
        return a
    if np.isnan(a):
        return b
    return a if a > b else b


def box_test(lo, hi, o, inv):
    tmin, tmax = f32(0), INF
    with np.errstate(all="ignore"):
        for k in range(3):
            t1 = f32(f32(lo[k] - o[k]) * inv[k])
            t2 = f32(f32(hi[k] - o[k]) * inv[k])
            tmin = _fmax(tmin, _fmin(t1, t2))
            tmax = _fmin(tmax, _fmax(t1, t2))
    return bool(tmin < tmax and tmax > 0), tmin


def _dot(a, b):
    return f32(f32(f32(a[0] * b[0]) + f32(a[1] * b[1])) + f32(a[2] * b[2]))


def _cross(a, b):
    return np.array([f32(f32(a[1] * b[2]) - f32(a[2] * b[1])), f32(f32(a[2] * b[0]) - f32(a[0] * b[2])), f32(f32(a[0] * b[1]) - f32(a[1] * b[0]))], f32)


def tri_test(tris, idx, o, d):
    q = tris[3 * idx:3 * idx + 3].reshape(12)
    p0, s1, s2 = q[0:3], q[3:6], q[6:9]
    with np.errstate(all="ignore"):
        p = _cross(d, s2)
        det = _dot(s1, p)
        inv_det = f32(f32(1) / det)
        t = (o - p0).astype(f32)
        u = f32(_dot(t, p) * inv_det)
        qv = _cross(t, s1)
        v = f32(_dot(d, qv) * inv_det)
        w = f32(f32(f32(1) - u) - v)
        dist = f32(_dot(s2, qv) * inv_det)
    return bool(dist > EPS and u >= 0 and v >= 0 and w >= 0), dist


class FlatScene:
    def __init__(self, flat):
        self.blob = flat["blob"]
        # blob order: nodes, objlds, meshes, objtab, tris (the triangles last: rt_device_scene.h)
        self.nodes = self.blob[flat["off_nodes"]:flat["off_objlds"]]
        self.tris = self.blob[flat["off_tris"]:flat["off_tris"] + 3 * flat["num_triangles"]]
        self.objlds = self.blob[flat["off_objlds"]:flat["off_meshes"]]
        self.objects = flat["objects"]
        self.max_stack = 0

    def mesh(self, ob, o, d, inv):
        best, best_prim = INF, -1
        if np.isnan(d).any():
            return False, best, -1          # a NaN direction fails every triangle test
        hit, rd = box_test(ob["v"][0:3], ob["v"][3:6], o, inv)
        cur = int(ob["root_ref"])
        if not hit or rd > best or ((cur & CHAIN) and not rd < best):
            return False, best, -1
        stack = []
        while True:
            descended = False
            if cur & LEAF:
                start, count = cur & 0xFFFFF, (cur >> 20) & 1023
                for k in range(count):
                    h, t = tri_test(self.tris, start + k, o, d)
                    if h and t < best:
                        best, best_prim = t, start + k
            else:
                ni = cur & 0x3FFFFFFF
                n = self.nodes[4 * ni:4 * ni + 4].reshape(16)
                lh, ld = box_test(n[0:3], n[3:6], o, inv)
                rh, rdist = box_test(n[6:9], n[9:12], o, inv)
                lref, rref = int(n[12:13].view(np.uint32)[0]), int(n[13:14].view(np.uint32)[0])
                l_push, r_push = lh and ld < best, rh and rdist < best
                l_first = bool(ld < rdist)
                first = (lref, ld, l_push) if l_first else (rref, rdist, r_push)
                second = (rref, rdist, r_push) if l_first else (lref, ld, l_push)
                if second[2]:
                    if first[2]:
                        stack.append((first[0], first[1]))
                        self.max_stack = max(self.max_stack, len(stack))
                    cur = second[0]
                    descended = True
                elif first[2]:
                    cur = first[0]
                    descended = True
            if descended:
                continue
            found = False
            while stack:
                ref, dd = stack.pop()
                if (dd < best) if (ref & CHAIN) else (not dd > best):
                    cur, found = ref, True
                    break
            if not found:
                break
        return best_prim >= 0, best, best_prim

    def closest_hit(self, origin, direction):
        """returns (hit, dist, object index, normal) like the kernel's collision step"""
        o = np.asarray(origin, f32)
        d = np.asarray(direction, f32)
        with np.errstate(all="ignore"):
            inv = (f32(1) / d).astype(f32)
        best_t, best_obj, best_prim = INF, -1, -1
        for i, ob in enumerate(self.objects):
            ty, hit, t, prim = int(ob["type"]), False, INF, -1
            if ty == 0:
                c, r = ob["v"][0:3], ob["v"][3]
                cq = (c - o).astype(f32)
                qa = _dot(d, d)
                qb = f32(_dot(d, cq) * f32(-2))
                qc = f32(_dot(cq, cq) - f32(r * r))
                disc = f32(f32(qb * qb) - f32(f32(f32(4) * qa) * qc))
                if disc >= 0:
                    dist = f32(f32(-qb - np.sqrt(disc, dtype=f32)) / f32(f32(2) * qa))
                    if dist > EPS:
                        hit, t = True, dist
            elif ty == 1:
                hit, t = tri_test(self.tris, int(ob["prim_start"]), o, d)
                prim = int(ob["prim_start"])
            elif ty in (2, 3):
                if ty == 3 and _dot(d, ob["v"][0:3]) < 0:
                    pass
                else:
                    hit, t, prim = self.quad(int(ob["prim_start"]), o, d)
            elif ty == 4:
                cb = INF
                for f in range(6):
                    fh, ft, fp = self.quad(int(ob["prim_start"]) + 2 * f, o, d)
                    if fh and ft < cb:
                        cb, prim, hit = ft, fp, True
                t = cb
            elif ty == 5:
                hit, t, prim = self.mesh(ob, o, d, inv)
            if hit and t <= best_t:
                best_t, best_obj, best_prim = t, i, prim
        if best_obj < 0:
            return False, INF, -1, None
        rec = self.objlds[4 * best_obj:4 * best_obj + 4].reshape(16)     # rt_objlds: a, b, c, d
        packed = int(rec[7:8].view(np.uint32)[0])
        P = (d * best_t + o).astype(f32)
        if packed & 32:
            v = (P - rec[8:11]).astype(f32)
            m = f32(f32(f32(v[0] * v[0]) + f32(v[1] * v[1])) + f32(v[2] * v[2]))
            N = (v * f32(f32(1) / np.sqrt(m, dtype=f32))).astype(f32)
        else:
            n = self.tris[3 * best_prim + 2][1:4]
            N = (-n if _dot(n, d) > 0 else n).astype(f32)
        return True, best_t, best_obj, N

    def quad(self, first, o, d):
        h1, t1 = tri_test(self.tris, first, o, d)
        h2, t2 = tri_test(self.tris, first + 1, o, d)
        return (h1 or h2), (t1 if h1 else t2), (first if h1 else first + 1)

    def leaf_histogram(self, ob_index, hist_len=4):
        """leaf sizes reachable from a mesh object's root (empty leaves excluded)"""
        hist = [0] * hist_len
        todo = [int(self.objects[ob_index]["root_ref"])]
        while todo:
            ref = todo.pop()
            if ref & LEAF:
                c = (ref >> 20) & 1023
                if c:
                    hist[min(c, hist_len - 1)] += 1
            else:
                ni = ref & 0x3FFFFFFF
                n = self.nodes[4 * ni:4 * ni + 4].reshape(16)
                todo.append(int(n[12:13].view(np.uint32)[0]))
                todo.append(int(n[13:14].view(np.uint32)[0]))
        return hist
