"""GPU parity for long runs (the box tree and the list-order
rejection scan) against the CPU oracle, on inputs chosen to break a tree that
is not exactly equivalent to the reference's insertion-ordered scan
(HittableList::hit, src/geometry/mod.rs:240-258): exact ties, extreme size
ratios, slivers, several long runs in one list, NaN/zero-direction rays, NaN
primitives, cameras inside dense clusters.  Tolerance: 0 ulp, ids bit-exact."""
import os

import numpy as np
import pytest

from raytracing_amd import (Camera, Cube, Dielectric, HittableList, Lambertian, Metal, Plane, Sphere, Triangle, _lib,
                            flat_scene_from_arrays, make_params, workloads)
from tests.test_gpu_parity import _bits, assert_same, check, render_gpu, render_oracle

pytestmark = pytest.mark.gpu
f32 = np.float32


def _mat(rng):
    k = rng.integers(0, 3)
    alb = tuple(rng.uniform(0.1, 1.0, 3))
    return (Metal(alb, float(rng.uniform(0, 0.5))), Lambertian(alb), Dielectric(alb, 1.5))[k]


def _cam(origin=(0.0, 1.5, 7.0), look_at=(0.0, 0.0, 0.0), fov=40.0, aperture=0.0):
    focus = float(np.linalg.norm(np.array(origin) - np.array(look_at)))
    return Camera.new(fov, 1.5, aperture, focus, origin, look_at, (0.0, 1.0, 0.0))


def test_exact_duplicates_in_a_tree_run(renderer, oracle, tables):
    """coincident spheres and triangles: equal t is accepted by `t_max < root`
    (src/geometry/mod.rs:74-79), so the LATER list entry must win, wherever the
    tree happens to put the two copies"""
    rng = np.random.default_rng(5)
    s = HittableList()
    base = [(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.2, 0.7))) for _ in range(60)]
    for c, r in base:
        s.add(Sphere(c, r, _mat(rng)))
    for k in rng.permutation(60):                       # the same spheres again, shuffled, other materials
        s.add(Sphere(base[k][0], base[k][1], _mat(rng)))
    tri = []
    for _ in range(50):
        a = rng.uniform(-3, 3, 3)
        tri.append((tuple(a), tuple(a + rng.uniform(-1, 1, 3)), tuple(a + rng.uniform(-1, 1, 3))))
    for t in tri:
        s.add(Triangle(*t, _mat(rng)))
    for k in rng.permutation(50):
        s.add(Triangle(*tri[k], _mat(rng)))
    flat = s.flatten()
    gpu = render_gpu(renderer, flat, _cam(), [make_params(120, 80, spp=3)])
    orc = render_oracle(oracle, tables, flat, _cam(), [make_params(120, 80, spp=3)])
    assert_same(gpu, orc, oracle)
    ids = gpu[1]
    assert not np.isin(ids, np.arange(0, 60)).any(), "an earlier duplicate sphere won a tie"
    assert not np.isin(ids, np.arange(120, 170)).any(), "an earlier duplicate triangle won a tie"


def test_extreme_size_ratios(renderer, oracle, tables):
    """radii from 1e-4 to 50 in one run: the upper tree levels must stay conservative
    for the smallest sphere under them"""
    rng = np.random.default_rng(6)
    s = HittableList()
    for k in range(300):
        r = float(10.0 ** rng.uniform(-4, 0))
        if k % 50 == 0:
            r = float(rng.uniform(5, 50))
        c = tuple(rng.uniform(-4, 4, 3)) if r < 5 else tuple(rng.uniform(-100, 100, 3))
        if k % 7 == 0:
            r = -r
        s.add(Sphere(c, r, _mat(rng)))
    # needles right in front of the camera so the tiny ones are actually hit
    for k in range(40):
        s.add(Sphere((float(rng.uniform(-0.5, 0.5)), float(rng.uniform(1.0, 2.0)), 6.0), 0.002, Lambertian((1.0, 0.2, 0.2))))
    check(renderer, oracle, tables, s.flatten(), _cam(), [make_params(128, 84, spp=3)])


def test_sliver_and_degenerate_triangles(renderer, oracle, tables):
    rng = np.random.default_rng(7)
    s = HittableList()
    for k in range(400):
        a = rng.uniform(-3, 3, 3)
        kind = k % 5
        if kind == 0:       # needle: 10 units long, 1e-4 wide
            d = rng.normal(size=3); d /= np.linalg.norm(d)
            b, c = a + 10 * d, a + 10 * d + 1e-4 * rng.normal(size=3)
        elif kind == 1:     # zero area (repeated vertex / collinear)
            b, c = a.copy(), a + rng.uniform(-1, 1, 3)
        elif kind == 2:     # huge
            b, c = a + rng.uniform(-200, 200, 3), a + rng.uniform(-200, 200, 3)
        elif kind == 3:     # tiny
            b, c = a + 1e-3 * rng.normal(size=3), a + 1e-3 * rng.normal(size=3)
        else:
            b, c = a + rng.uniform(-1, 1, 3), a + rng.uniform(-1, 1, 3)
        s.add(Triangle(tuple(a), tuple(b), tuple(c), _mat(rng)))
    check(renderer, oracle, tables, s.flatten(), _cam(), [make_params(128, 84, spp=3)])


def test_shared_edges_and_vertices(renderer, oracle, tables):
    """a closed mesh seen from outside and from inside: rays through shared edges hit two
    triangles at (nearly) the same t"""
    tris = workloads.uv_sphere_mesh(center=(0.0, 0.0, 0.0), radius=1.5, n_lat=10, n_lon=20)
    s = HittableList()
    for k, (a, b, c) in enumerate(tris):
        s.add(Triangle(a, b, c, Dielectric((0.9, 0.95, 1.0), 1.4) if k % 3 else Lambertian((0.8, 0.5, 0.2))))
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -1.5, 0.0), Lambertian((0.4, 0.4, 0.4))))
    flat = s.flatten()
    check(renderer, oracle, tables, flat, _cam((0.0, 0.5, 5.0)), [make_params(128, 84, spp=3)])
    check(renderer, oracle, tables, flat, _cam((0.1, 0.2, 0.3), look_at=(1.0, 0.3, -1.0), fov=90.0), [make_params(96, 64, spp=3)])


def test_several_long_runs_in_one_list(renderer, oracle, tables):
    """spheres / plane / spheres / triangles / cube / triangles / short sphere run: every run
    keeps its own place in the order, hits across runs shrink t_max for the later ones"""
    rng = np.random.default_rng(8)
    s = HittableList()

    def spheres(n, lo, hi):
        for _ in range(n):
            s.add(Sphere(tuple(rng.uniform(lo, hi, 3)), float(rng.uniform(0.1, 0.5)), _mat(rng)))

    def tris(n, lo, hi):
        for _ in range(n):
            a = rng.uniform(lo, hi, 3)
            s.add(Triangle(tuple(a), tuple(a + rng.uniform(-0.8, 0.8, 3)), tuple(a + rng.uniform(-0.8, 0.8, 3)), _mat(rng)))

    spheres(100, -3, 3)
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -3.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
    spheres(200, -4, 4)
    tris(300, -3, 3)
    s.add(Cube((0.0, 0.0, 0.0), 1.0, 1.0, 1.0, Dielectric((0.9, 0.9, 0.9), 1.5)))
    tris(100, -2, 2)
    spheres(40, -2, 2)          # a rejection-scan run (32..63)
    spheres(0, 0, 0)
    s.add(Cube((2.0, 0.0, 0.0), 0.5, 2.0, 0.5, Metal((0.9, 0.9, 0.9), 0.0)))
    spheres(10, -2, 2)          # a short exact run
    check(renderer, oracle, tables, s.flatten(), _cam(aperture=0.2), [make_params(128, 84, spp=3)])


def test_camera_inside_a_dense_cluster(renderer, oracle, tables):
    rng = np.random.default_rng(9)
    s = HittableList()
    for _ in range(500):
        s.add(Sphere(tuple(rng.uniform(-1.5, 1.5, 3)), float(rng.uniform(0.2, 0.6)), _mat(rng)))
    check(renderer, oracle, tables, s.flatten(), _cam((0.05, 0.02, 0.1), look_at=(1.0, 0.2, -0.5), fov=100.0),
          [make_params(96, 64, spp=3)])


@pytest.mark.parametrize("t_min", [-4.0, -0.5, 0.0, 0.5, 3.0])
def test_t_min_other_than_the_reference_value(renderer, oracle, tables, t_min):
    """Hittable::hit takes t_min from its caller (0.001 in ray_colour, src/raytracer.rs:75); with a
    negative one, hits behind the ray origin count (src/geometry/mod.rs:74-79), so the boxes behind
    the origin must not be skipped; with a large one, near hits are passed over for their second
    root or for farther primitives"""
    rng = np.random.default_rng(15)
    s = HittableList()
    for _ in range(300):
        s.add(Sphere(tuple(rng.uniform(-4, 4, 3)), float(rng.uniform(0.2, 0.7)), _mat(rng)))
    for _ in range(150):
        a = rng.uniform(-4, 4, 3)
        s.add(Triangle(tuple(a), tuple(a + rng.uniform(-1, 1, 3)), tuple(a + rng.uniform(-1, 1, 3)), _mat(rng)))
    flat = s.flatten()
    cam = _cam((0.3, 0.2, 0.5), look_at=(1.0, 0.4, -2.0), fov=70.0)      # inside the cluster
    p = [make_params(72, 48, spp=3, t_min=t_min)]
    orc = render_oracle(oracle, tables, flat, cam, p)
    assert_same(render_gpu(renderer, flat, cam, p), orc, oracle)
    os.environ["RT_HIP_FLAT_SCAN"] = "1"
    try:
        assert_same(render_gpu(renderer, flat, cam, p), orc, oracle)
    finally:
        del os.environ["RT_HIP_FLAT_SCAN"]


def test_ties_at_t_min_zero_in_a_coplanar_sheet(renderer, oracle, tables):
    """camera origin inside a sheet of coplanar triangles, t_min = 0: every ray hits many triangles at
    t = +-0 exactly and the LAST of them in list order keeps the hit (src/geometry/mod.rs:74-79 accepts
    equality); the tree must still reach every one of them although closest_so_far equals t_min"""
    rng = np.random.default_rng(16)
    s = HittableList()
    for _ in range(200):
        s.add(Sphere(tuple(rng.uniform(-5, 5, 3)), float(rng.uniform(0.2, 0.6)), _mat(rng)))
    for _ in range(300):
        a = np.array((rng.uniform(-6, 6), 1.0, rng.uniform(-6, 6)))
        b = a + np.array((rng.uniform(-3, 3), 0.0, rng.uniform(-3, 3)))
        c = a + np.array((rng.uniform(-3, 3), 0.0, rng.uniform(-3, 3)))
        s.add(Triangle(tuple(a), tuple(b), tuple(c), _mat(rng)))
    flat = s.flatten()
    cam = _cam((0.25, 1.0, 0.5), look_at=(2.0, 0.5, -3.0), fov=60.0)
    for t_min in (0.0, -0.25):
        p = [make_params(64, 40, spp=2, t_min=t_min)]
        orc = render_oracle(oracle, tables, flat, cam, p)
        gpu = render_gpu(renderer, flat, cam, p)
        assert_same(gpu, orc, oracle)
        assert (gpu[1] >= 200).all(), "every ray starts inside some triangle of the sheet"


def _null_camera(direction=(0.0, 0.0, 0.0)):
    """every pixel gets the same ray origin (0,0,5) and `direction`"""
    cam = _lib.rt_camera()
    cam.origin[:] = (0.0, 0.0, 5.0)
    cam.lower_left_corner[:] = tuple(o + d for o, d in zip((0.0, 0.0, 5.0), direction))
    cam.horizontal[:] = (0.0, 0.0, 0.0)
    cam.vertical[:] = (0.0, 0.0, 0.0)
    cam.u[:] = (1.0, 0.0, 0.0)
    cam.v[:] = (0.0, 1.0, 0.0)
    cam.lens_radius = 0.0
    return cam


def _same_with_nans(gpu, orc, oracle):
    a, b = gpu[0], orc[0][..., :3]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.array_equal(_bits(a)[ok], _bits(b)[ok])
    assert np.array_equal(gpu[1], orc[1]), "%d primary ids differ" % (gpu[1] != orc[1]).sum()
    assert gpu[3] == orc[2], "segments %d != %d" % (gpu[3], orc[2])
    assert np.array_equal(gpu[2], oracle.pack(orc[0]))


def test_zero_direction_rays_through_tree_runs(renderer, oracle, tables):
    """direction (0,0,0): a = 0, half_b = 0, the discriminant is 0*c and the roots are 0/0 -
    whatever the reference's arithmetic yields for each sphere in list order is the answer"""
    rng = np.random.default_rng(10)
    s = HittableList()
    for _ in range(150):
        s.add(Sphere(tuple(rng.uniform(-6, 6, 3)), float(rng.uniform(0.3, 2.0)), _mat(rng)))
    for _ in range(100):
        a = rng.uniform(-3, 3, 3)
        s.add(Triangle(tuple(a), tuple(a + rng.uniform(-2, 2, 3)), tuple(a + rng.uniform(-2, 2, 3)), _mat(rng)))
    flat = s.flatten()
    p = [make_params(24, 16, spp=2)]
    _same_with_nans(render_gpu(renderer, flat, _null_camera(), p), render_oracle(oracle, tables, flat, _null_camera(), p), oracle)
    # and an ordinary direction shared by all pixels (fully coherent wave)
    cam = _null_camera((0.05, 0.02, -1.0))
    _same_with_nans(render_gpu(renderer, flat, cam, p), render_oracle(oracle, tables, flat, cam, p), oracle)


def test_nan_closest_hit_entering_tree_runs(renderer, oracle, tables):
    """a plane hit with t = 0/0 = NaN (src/geometry/mod.rs:114-118) makes t_max NaN for every
    later run; `t_max < root` is then false for all roots, so later hits keep replacing it"""
    rng = np.random.default_rng(11)
    s = HittableList()
    s.add(Plane((0.0, 1.0, 0.0), (0.0, 0.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
    for _ in range(120):
        s.add(Sphere((float(rng.uniform(-4, 4)), float(rng.uniform(-0.5, 0.5)), float(rng.uniform(-8, -2))),
                     float(rng.uniform(0.2, 0.6)), _mat(rng)))
    for _ in range(80):
        a = np.array((rng.uniform(-4, 4), rng.uniform(-0.5, 0.5), rng.uniform(-8, -2)))
        s.add(Triangle(tuple(a), tuple(a + rng.uniform(-1, 1, 3)), tuple(a + rng.uniform(-1, 1, 3)), _mat(rng)))
    cam = _lib.rt_camera()
    cam.origin[:] = (0.0, 0.0, 0.0)
    cam.lower_left_corner[:] = (-2.0, -1.0, -2.0)
    cam.horizontal[:] = (4.0, 0.0, 0.0)
    cam.vertical[:] = (0.0, 2.0, 0.0)
    cam.u[:] = (1.0, 0.0, 0.0)
    cam.v[:] = (0.0, 1.0, 0.0)
    cam.lens_radius = 0.0
    tb = dict(tables)
    tb["disc"] = np.zeros(128, dtype=f32)        # no jitter: the middle row lies in the plane
    renderer.set_tables(tb)
    try:
        p = [make_params(65, 33, spp=2)]
        flat = s.flatten()
        _same_with_nans(render_gpu(renderer, flat, cam, p), render_oracle(oracle, tb, flat, cam, p), oracle)
    finally:
        renderer.set_tables(tables)


def test_nan_and_infinite_primitives_in_tree_runs(renderer, oracle, tables):
    """NaN centres (hit by every ray with t = NaN), infinite radii, coordinates of 1e19 and
    1e30 inside long runs: none of them may poison the bounding volumes of their neighbours"""
    rng = np.random.default_rng(12)
    n = 200
    c = rng.uniform(-3, 3, (n, 3)).astype(f32)
    r = rng.uniform(0.2, 0.6, n).astype(f32)
    c[17] = (np.nan, 0.0, 0.0)
    r[40] = np.inf
    c[77] = (1e19, 0.0, 0.0)
    c[78] = (0.0, -1e30, 0.0); r[78] = 1e30
    r[120] = np.nan
    nt = 100
    v0 = rng.uniform(-3, 3, (nt, 3)).astype(f32)
    v1 = (v0 + rng.uniform(-1, 1, (nt, 3))).astype(f32)
    v2 = (v0 + rng.uniform(-1, 1, (nt, 3))).astype(f32)
    v1[5] = (np.nan, 1.0, 1.0)
    v2[50] = (np.inf, 0.0, 0.0)
    v0[60] = (1e25, 1e25, 1e25)
    kinds = rng.integers(0, 3, n + nt).astype(np.int32)
    flat = flat_scene_from_arrays(
        sph_cx=c[:, 0], sph_cy=c[:, 1], sph_cz=c[:, 2], sph_r=r, sph_mat=np.arange(n), sph_id=np.arange(n),
        tri_v0x=v0[:, 0], tri_v0y=v0[:, 1], tri_v0z=v0[:, 2], tri_v1x=v1[:, 0], tri_v1y=v1[:, 1], tri_v1z=v1[:, 2],
        tri_v2x=v2[:, 0], tri_v2y=v2[:, 1], tri_v2z=v2[:, 2], tri_mat=np.arange(n, n + nt), tri_id=np.arange(n, n + nt),
        mat_kind=kinds, mat_r=rng.uniform(0.2, 1, n + nt), mat_g=rng.uniform(0.2, 1, n + nt),
        mat_b=rng.uniform(0.2, 1, n + nt), mat_param=np.where(kinds == 2, 1.5, 0.2))
    p = [make_params(64, 40, spp=2)]
    _same_with_nans(render_gpu(renderer, flat, _cam(), p), render_oracle(oracle, tables, flat, _cam(), p), oracle)


def test_far_away_camera_and_scene(renderer, oracle, tables):
    """coordinates around 1e6 (f32 spacing 0.06): margins scale with |o|, |c|"""
    rng = np.random.default_rng(13)
    off = np.array((1.0e6, -2.0e6, 5.0e5))
    s = HittableList()
    for _ in range(300):
        s.add(Sphere(tuple(off + rng.uniform(-30, 30, 3)), float(rng.uniform(1.0, 5.0)), _mat(rng)))
    cam = _cam(tuple(off + np.array((0.0, 10.0, 90.0))), look_at=tuple(off))
    check(renderer, oracle, tables, s.flatten(), cam, [make_params(96, 64, spp=3)])


def _camera_looking_at_world_origin(origin, scale):
    """pinhole camera at `origin` whose ray directions are about `scale` long"""
    o = np.array(origin, dtype=np.float64)
    w = -o / np.linalg.norm(o)
    u = np.cross((0.0, 1.0, 0.0), w) if abs(w[1]) < 0.9 else np.cross((1.0, 0.0, 0.0), w)
    u /= np.linalg.norm(u)
    v = np.cross(w, u)
    cam = _lib.rt_camera()
    cam.origin[:] = tuple(o)
    hor, ver = 0.6 * scale * u, 0.4 * scale * v
    cam.lower_left_corner[:] = tuple(o - hor / 2 - ver / 2 + scale * w)
    cam.horizontal[:] = tuple(hor)
    cam.vertical[:] = tuple(ver)
    cam.u[:] = tuple(u)
    cam.v[:] = tuple(v)
    cam.lens_radius = 0.0
    return cam


def _tree_and_flat_scan_against_oracle(renderer, oracle, tables, flat, cam, p):
    orc = render_oracle(oracle, tables, flat, cam, p)
    _same_with_nans(render_gpu(renderer, flat, cam, p), orc, oracle)
    os.environ["RT_HIP_FLAT_SCAN"] = "1"
    try:
        _same_with_nans(render_gpu(renderer, flat, cam, p), orc, oracle)
    finally:
        del os.environ["RT_HIP_FLAT_SCAN"]


@pytest.mark.parametrize("n_spheres", [40, 200])
def test_rays_with_extreme_origins_and_directions(renderer, oracle, tables, n_spheres):
    """origins from 1e17 to 1e30 (|o|^2 overflows f32 from 1.8e19 on) and directions of length 1e-25
    to 1e20: Sphere::hit then works with infinities and accepts NaN roots
    (src/geometry/mod.rs:62-79), so no conservative test may reject anything for such rays - in the
    rejection scan of 32-63 spheres, in the list-order scan of longer runs, and in the tree"""
    rng = np.random.default_rng(14)
    s = HittableList()
    for _ in range(n_spheres):
        s.add(Sphere(tuple(rng.uniform(-3, 3, 3)), float(rng.uniform(0.2, 0.7)), _mat(rng)))
    flat = s.flatten()
    p = [make_params(48, 32, spp=2)]
    cases = [((0.0, 0.0, 1e17), 1.0), ((0.0, 0.0, 1e19), 1.0), ((3e19, -2e19, 1e19), 1.0), ((5e21, 2e20, -3e21), 1.0),
             ((0.0, 1e30, 0.0), 1.0), ((0.0, 0.0, 8.0), 1e7), ((0.0, 0.0, 8.0), 1e20), ((0.0, 0.0, 8.0), 1e-8),
             ((0.0, 0.0, 8.0), 1e-25), ((1e10, 0.0, 1e10), 1e15)]
    for origin, scale in cases:
        _tree_and_flat_scan_against_oracle(renderer, oracle, tables, flat,
                                           _camera_looking_at_world_origin(origin, scale), p)


@pytest.mark.parametrize("n_spheres", [40, 200])
@pytest.mark.parametrize("far,length", [(8e17, 1.0), (8e17, 3.0), (8e17, 5.0), (8e17, 30.0), (1e14, 3e5), (1e16, 1000.0),
                                        (1e16, 3000.0), (3e12, 9e5), (3e13, 9e5), (2e10, 9.9e5)])
def test_rays_whose_discriminant_overflows_for_ordinary_spheres(renderer, oracle, tables, n_spheres, far, length):
    """origins and direction lengths that each stay in the ordinary range but whose product does
    not: half_b^2 and a*c (src/geometry/mod.rs:63-66) pass 3.4e38 from |d| |o - c| = 1.8e19 on, the
    discriminant is inf - inf and every such sphere is accepted with a NaN root, also spheres the
    ray passes far from.  A camera cannot make such rays (its directions are differences of
    points near the origin); rays scattered by a plane at height `far` whose normal is `length`
    long can (Plane::hit does not normalise, src/geometry/mod.rs:112-130).  Products on both
    sides of the limit, in the rejection scan (40 spheres) and in the tree (200)."""
    rng = np.random.default_rng(15)
    s = HittableList()
    for _ in range(n_spheres):
        s.add(Sphere(tuple(rng.uniform(-30, 30, 3) + np.array((0.0, -60.0, 0.0))), float(rng.uniform(0.2, 3.0)), _mat(rng)))
    s.add(Plane((0.0, length, 0.0), (0.0, far, 0.0), Lambertian((0.8, 0.6, 0.4)) if n_spheres == 40 else Metal((0.8, 0.6, 0.4), 0.3)))
    cam = _cam((0.0, 0.0, 0.0), look_at=(0.1, 1.0, 0.2), fov=50.0)
    _tree_and_flat_scan_against_oracle(renderer, oracle, tables, s.flatten(), cam, [make_params(40, 24, spp=2, max_bounces=4)])


def test_bounce_from_a_far_plane_back_towards_a_sphere_run(renderer, oracle, tables):
    """a plane at 1e19 is hit at t = 1e18; the scattered rays start there with directions of length
    ten and more, and the discriminants of all spheres of the run overflow (found by the fuzzer)"""
    rng = np.random.default_rng(16)
    s = HittableList()
    for _ in range(300):
        s.add(Sphere(tuple(np.array((470.0, 2260.0, 1790.0)) + rng.uniform(-60, 60, 3)), float(rng.uniform(0.3, 7.0)), _mat(rng)))
    s.add(Plane((11.2, 2.0, 14.2), (500.0, 1e19, 1800.0), Lambertian((0.88, 0.40, 0.29))))
    cam = _cam((465.0, 2311.0, 1745.0), look_at=(470.0, 2260.0, 1790.0), fov=80.0)
    _tree_and_flat_scan_against_oracle(renderer, oracle, tables, s.flatten(), cam, [make_params(96, 48, spp=3, max_bounces=4)])


def test_list_order_scan_equals_tree_on_the_gpu(renderer):
    """RT_HIP_FLAT_SCAN=1 uploads the same scene without a tree (every sphere tested in list
    order); sums, ids and segment counts must be bit-identical at a size the oracle cannot reach"""
    wl = workloads.workload("c4")
    p = [make_params(640, 360, spp=4, seed=3)]
    tree = render_gpu(renderer, wl["flat"], wl["camera"], p)
    os.environ["RT_HIP_FLAT_SCAN"] = "1"
    try:
        flat = render_gpu(renderer, wl["flat"], wl["camera"], p)
    finally:
        del os.environ["RT_HIP_FLAT_SCAN"]
    assert np.array_equal(_bits(tree[0]), _bits(flat[0]))
    assert np.array_equal(tree[1], flat[1]) and np.array_equal(tree[2], flat[2]) and tree[3] == flat[3]


def test_fuzz_tree_against_list_order_scan():
    """140 random scenes (a fixed number, whatever the load of the machine: slivers, lattices of touching spheres, coplanar sheets,
    extents from 0.1 to 1000, cameras inside and far outside) at 1280x720: the tree scan
    and the list-order scan of the same library must agree bit for bit"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_tree_vs_list.py"), "90n", "100000"],
                         capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "mismatches 0" in out.stdout.splitlines()[-1], out.stdout[-2000:] + out.stderr[-2000:]
    # and with NaN, infinite, huge or tiny numbers in every scene
    out = subprocess.run([sys.executable, os.path.join(root, "tools", "stress_tree_vs_list.py"), "50n", "5100000", "1.0"],
                         capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "mismatches 0" in out.stdout.splitlines()[-1], out.stdout[-2000:] + out.stderr[-2000:]


def test_fuzz_gpu_against_oracle():
    """the same random scenes at 96x54 against the CPU oracle (tests/stress_gpu_vs_oracle.py), with a
    random t_min now and then: sums, ids, packed pixels and segment counts identical"""
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_gpu_vs_oracle.py"), "160n", "700000"],
                         capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "mismatches 0" in out.stdout.splitlines()[-1], out.stdout[-2000:] + out.stderr[-2000:]
    # and with NaN, infinite or huge coordinates in every scene
    out = subprocess.run([sys.executable, os.path.join(root, "tests", "stress_gpu_vs_oracle.py"), "120n", "4900000", "1.0"],
                         capture_output=True, text=True, timeout=200)
    assert out.returncode == 0 and "mismatches 0" in out.stdout.splitlines()[-1], out.stdout[-2000:] + out.stderr[-2000:]


def test_nan_sphere_behind_a_cube_hit_with_negative_t(renderer, oracle, tables):
    """the camera sits inside a cube that comes first in the list: Cube::hit reports its negative
    entry distance (src/geometry/mod.rs:158-223), after which no finite root can be accepted
    (t_min > closest_so_far) - but a sphere with a huge centre still is, with a NaN distance, and then
    every later sphere the ray's line meets; the tree must not cull on the negative distance"""
    rng = np.random.default_rng(17)
    s = HittableList()
    s.add(Cube((0.0, 0.0, 0.0), 30.0, 30.0, 30.0, Lambertian((0.4, 0.5, 0.6))))
    c = rng.uniform(-6, 6, (400, 3)).astype(f32)
    r = rng.uniform(0.2, 0.8, 400).astype(f32)
    for k in range(400):
        s.add(Sphere(tuple(c[k]), float(r[k]), _mat(rng)))
    flat = s.flatten()
    a = flat.arrays
    a["sph_cx"][250] = 1e38           # |o - c|^2 overflows: discriminant inf - inf
    a["sph_r"][120] = np.nan
    for cam in (_cam((1.0, 2.0, 9.0)), _cam((0.2, 0.1, 0.3), look_at=(3.0, 1.0, -2.0), fov=80.0)):
        p = [make_params(72, 48, spp=2)]
        _same_with_nans(render_gpu(renderer, flat, cam, p), render_oracle(oracle, tables, flat, cam, p), oracle)
