"""The two forms of the slab test are the same function where the kernel uses the short one (CPU property test).

rt_pixel.h box_enter is the reference's BoundingBox::ray_hits (src/objects.cu:404-434) folded with the traversal's
`entry distance < best`; box_enter_med3 (round 4) is two chains of three v_med3_f32 - clamp 0 and `best` through the three
slabs' intervals, enter iff the first is below the second.  rt_pixel.h argues that both make the same decision and, where
the box is entered, report the same entry distance, for every ray whose slab distances are numbers (no 0 * inf: no direction
component of exactly 0 - those rays take the min / max copy of the loop).  Here both forms are evaluated in binary32 (numpy)
on millions of random and adversarial cases: boxes in front, behind, around the origin, flat boxes, touching slabs, origins ON
box planes, huge and tiny directions, infinite reciprocals of non-zero... and `best` below, inside and beyond the box."""
import numpy as np

F = np.float32
INF_F = F(1073741824.0)


def box_enter(b, o, inv, best):
    """(enter, tmin) as rt_pixel.h box_enter: fminf / fmaxf drop NaNs"""
    tmin = np.zeros(o.shape[0], F)
    tmax = np.full(o.shape[0], INF_F, F)
    with np.errstate(all="ignore"):
        for k in range(3):
            t1 = (b[:, k] - o[:, k]) * inv[:, k]
            t2 = (b[:, 3 + k] - o[:, k]) * inv[:, k]
            tmin = np.fmax(tmin, np.fmin(t1, t2))
            tmax = np.fmin(tmax, np.fmax(t1, t2))
        return tmin < np.fmin(tmax, best), tmin


def med3(a, b, c):
    """median of three numbers (no NaN among them: the caller guarantees it)"""
    return np.maximum(np.minimum(a, b), np.minimum(np.maximum(a, b), c))


def box_enter_med3(b, o, inv, best):
    with np.errstate(all="ignore"):
        lo = np.zeros(o.shape[0], F)
        hi = best.copy()
        for k in range(3):
            t1 = (b[:, k] - o[:, k]) * inv[:, k]
            t2 = (b[:, 3 + k] - o[:, k]) * inv[:, k]
            lo = med3(t1, t2, lo)
            hi = med3(t1, t2, hi)
        return lo < hi, lo


def cases(rng, n):
    lo = rng.uniform(-3, 3, (n, 3)).astype(F)
    size = rng.uniform(0, 3, (n, 3)).astype(F)
    size[rng.random((n, 3)) < 0.15] = 0                       # flat boxes (the reference drops them: strict tmin < tmax)
    b = np.concatenate([lo, lo + size], axis=1).astype(F)      # min <= max, as the BVH builder stores them
    o = rng.uniform(-4, 4, (n, 3)).astype(F)
    on_plane = rng.random((n, 3)) < 0.15                       # origins exactly ON a box plane (a bounce off a triangle that defines it)
    which = rng.integers(0, 2, (n, 3))
    o = np.where(on_plane, np.where(which == 0, b[:, :3], b[:, 3:]), o).astype(F)
    d = rng.standard_normal((n, 3)).astype(F)
    d *= (10.0 ** rng.uniform(-30, 3, (n, 3))).astype(F)       # components from 1e-30 (reciprocals near the top of the range) to 1e3
    aimed = rng.random(n) < 0.6                                # most rays are aimed at a point of their box (otherwise few would enter)
    target = (b[:, :3] + (b[:, 3:] - b[:, :3]) * rng.random((n, 3)).astype(F)).astype(F)
    d = np.where(aimed[:, None], (target - o) * (10.0 ** rng.uniform(-3, 3, (n, 1))).astype(F), d).astype(F)
    d[d == 0] = F(1e-20)                                       # the short form is never used with a zero component
    with np.errstate(all="ignore"):
        inv = (F(1.0) / d).astype(F)
    best = np.where(rng.random(n) < 0.3, INF_F, (10.0 ** rng.uniform(-3, 3, n))).astype(F)
    return b, o, inv, best


def test_med3_form_makes_the_reference_decision_and_reports_its_entry_distance():
    rng = np.random.default_rng(42)
    entered = 0
    for _ in range(8):
        b, o, inv, best = cases(rng, 500_000)
        e0, t0 = box_enter(b, o, inv, best)
        e1, t1 = box_enter_med3(b, o, inv, best)
        with np.errstate(all="ignore"):
            t = np.concatenate([(b[:, :3] - o) * inv, (b[:, 3:] - o) * inv], axis=1)
        ok = ~np.isnan(t).any(axis=1)                          # inf * 0 cannot happen with non-zero components, (b - o) * inf = inf can: kept
        assert ok.mean() > 0.99
        assert np.array_equal(e0[ok], e1[ok])
        both = ok & e0
        # the same NUMBER (a zero may come out as +0 from one form and -0 from the other: the distance only ever feeds <, == and
        # the stack, where the two zeros are one value)
        assert np.array_equal(t0[both], t1[both]) and not np.isnan(t0[both]).any()
        entered += int(both.sum())
    assert entered > 200_000                                   # (the comparison of distances is not vacuous)


def test_a_zero_direction_component_is_why_the_kernel_keeps_the_min_max_form():
    """origin on a box plane + direction component 0: (b - o) * (1 / 0) = 0 * inf = NaN.  fminf / fmaxf drop the NaN and the slab
    collapses to [inf, inf] (or [-inf, -inf]): the reference's answer is a miss, whichever plane the origin lies on.  A median of
    three with a NaN operand is whatever the hardware's NaN rule says (v_med3_f32 returns a min3 or the NaN depending on the
    mode), so the kernel never lets the short form see one: such rays take the min / max copy of the loop
    (tests/test_gpu_parity.py::test_rays_with_a_zero_direction_component renders them on the device)"""
    b = np.array([[0, 0, 0, 1, 1, 1]] * 2, F)
    o = np.array([[0.0, 0.5, -1.0], [1.0, 0.5, -1.0]], F)       # on the plane x == 0, on the plane x == 1
    d = np.array([[0.0, 0.0, 1.0]] * 2, F)
    with np.errstate(all="ignore"):
        inv = (F(1.0) / d).astype(F)
        t = np.concatenate([(b[:, :3] - o) * inv, (b[:, 3:] - o) * inv], axis=1)
    assert np.isnan(t[0, 0]) and np.isnan(t[1, 3])
    e0, _ = box_enter(b, o, inv, np.array([INF_F, INF_F], F))
    assert not e0.any()
    # the same ray a hair inside the box is a hit at distance 1 in both forms (no NaN: 0.25 * inf = inf)
    o[:, 0] = F(0.25)
    e0, t0 = box_enter(b, o, inv, np.array([INF_F, INF_F], F))
    e1, t1 = box_enter_med3(b, o, inv, np.array([INF_F, INF_F], F))
    assert e0.all() and e1.all() and (t0 == F(1.0)).all() and (t1 == F(1.0)).all()
