"""GPU parity: librt_hip.so (through its C ABI) against the CPU oracle on the
same inputs.  Integer outputs (primary ids, packed pixels, segment and sample
counts) must be bit-identical; so must the f32 sums, because both sides
evaluate the same binary32 operations in the same order - the tolerance is 0 ulp.
Sizes are kept to what the oracle finishes in seconds; the full benchmark sizes
are covered by size-independent properties in test_gpu_properties.py."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import (Camera, Cube, Dielectric, HittableList, Lambertian, Metal, Plane, Scene,
                            Sphere, Triangle, _lib, image_height, make_params, workloads)
from tests.test_multi_gpu_gloo import block_from_packed

pytestmark = pytest.mark.gpu
f32 = np.float32


def _bits(a):
    return np.ascontiguousarray(a, dtype=np.float32).view(np.uint32)


def render_gpu(renderer, flat, cam, params_list):
    renderer.set_scene(flat)
    renderer.set_camera(cam)
    renderer.begin_frame(params_list[0].width, params_list[0].height)
    for p in params_list:
        renderer.render(p)
    return (renderer.read_accum(), renderer.read_primary_ids(), renderer.read_packed(),
            renderer.ray_count(), renderer.read_sample_counts())


def render_oracle(oracle, tables, flat, cam, params_list, threads=8):
    acc = None
    ids = None
    seg = 0
    for p in params_list:
        acc, i, s = oracle.render(flat, cam, tables, p, threads=threads, accum=acc)
        if p.sample_begin == 0:
            ids = i if ids is None else np.maximum(i, ids)     # unrendered bands hold -1
        seg += s
    return acc, ids, seg


def assert_same(gpu, orc_out, oracle, ids_mask=None):
    acc, ids, packed, seg, counts = gpu
    o_acc, o_ids, o_seg = orc_out
    if o_ids is None:
        assert (ids == -1).all()          # no call rendered sample 0: nothing recorded
    elif ids_mask is None:
        assert np.array_equal(ids, o_ids), "%d primary ids differ" % (ids != o_ids).sum()
    else:
        assert np.array_equal(ids[ids_mask], o_ids[ids_mask])
    diff = _bits(acc) != _bits(o_acc[..., :3])
    assert not diff.any(), "%d of %d sums differ" % (diff.sum(), diff.size)
    assert np.array_equal(counts, o_acc[..., 3].astype(np.uint32))
    done = o_acc[..., 3] > 0
    assert np.array_equal(packed[done], oracle.pack(o_acc)[done])
    assert seg == o_seg


def check(renderer, oracle, tables, flat, cam, params_list):
    assert_same(render_gpu(renderer, flat, cam, params_list),
                render_oracle(oracle, tables, flat, cam, params_list), oracle)


# ---------------------------------------------------------------- reference scene
def test_default_scene_default_size(renderer, oracle, tables):
    """configs[0]: the reference's own run, 600x337 at 1 spp (src/raytracer.rs:118-120)"""
    w = 600
    h = image_height(w)
    assert h == 337
    check(renderer, oracle, tables, Scene.get_scene().flatten(), Scene.get_camera(), [make_params(w, h, spp=1)])


def test_default_scene_depth_of_field_multi_sample(renderer, oracle, tables):
    """configs[1] at reduced size: aperture 0.1 thin lens, several samples per pixel"""
    check(renderer, oracle, tables, Scene.get_scene().flatten(), Scene.get_camera(aperture=0.1),
          [make_params(320, 180, spp=8)])


def test_dolly_positions(renderer, oracle, tables):
    """the camera of frames 1, 37 and 99 of the reference's dolly (src/scene/mod.rs:61-63,78-83)"""
    flat = Scene.get_scene().flatten()
    for counter in (1, 37, 99):
        cam = oracle.scene_camera(counter, 0.0)
        check(renderer, oracle, tables, flat, cam, [make_params(200, 112, spp=2)])


@pytest.mark.parametrize("w,h", [(2, 2), (7, 5), (8, 8), (33, 9), (61, 37), (64, 64), (257, 31)])
def test_ragged_sizes(renderer, oracle, tables, w, h):
    """widths and heights that do not fill 8x8 tiles or 32-wide workgroups"""
    check(renderer, oracle, tables, Scene.get_scene().flatten(), Scene.get_camera(aperture=0.05),
          [make_params(w, h, spp=3)])


def test_progressive_chunks_equal_one_shot(renderer, oracle, tables):
    flat = Scene.get_scene().flatten()
    cam = Scene.get_camera(aperture=0.1)
    chunks = [make_params(160, 90, spp=7, sample_begin=b, sample_count=c) for b, c in ((0, 3), (3, 1), (4, 3))]
    one = render_gpu(renderer, flat, cam, [make_params(160, 90, spp=7)])
    many = render_gpu(renderer, flat, cam, chunks)
    assert np.array_equal(_bits(one[0]), _bits(many[0]))
    assert all(np.array_equal(a, b) for a, b in zip(one[1:], many[1:]))
    assert_same(many, render_oracle(oracle, tables, flat, cam, chunks), oracle)


def test_sample_ranges_longer_than_one_launch(renderer, oracle, tables):
    """a launch takes up to 512 samples per pixel (its work units then are single pixels) and
    rt_render cuts a longer range into several; 150 samples in one call (one launch, units of two
    pixels) and in 70 + 80, and 600 samples (launches of 512 and 88) are the oracle's sums"""
    flat = Scene.get_scene().flatten()
    cam = Scene.get_camera(aperture=0.1)
    one = render_gpu(renderer, flat, cam, [make_params(40, 24, spp=150)])
    two = render_gpu(renderer, flat, cam, [make_params(40, 24, spp=150, sample_begin=0, sample_count=70),
                                           make_params(40, 24, spp=150, sample_begin=70, sample_count=80)])
    assert np.array_equal(_bits(one[0]), _bits(two[0])) and one[3] == two[3]
    assert_same(one, render_oracle(oracle, tables, flat, cam, [make_params(40, 24, spp=150)]), oracle)
    long = render_gpu(renderer, flat, cam, [make_params(21, 13, spp=600)])
    assert_same(long, render_oracle(oracle, tables, flat, cam, [make_params(21, 13, spp=600)]), oracle)


def test_kernel_info_names_the_launched_instantiation(renderer, monkeypatch):
    """rt_kernel_info: the reference's scene runs the kernel without long runs; a scene with a run of 32
    spheres and more (or 64 triangles and more) a tree kernel, for up to 10 bounces or up to 16; the
    list-order rejection scan only when asked for (RT_HIP_FLAT_SCAN=1, a measurement mode)"""
    from raytracing_amd import HittableList, Lambertian, Sphere, workloads, _lib
    renderer.set_scene(Scene.get_scene().flatten())
    assert renderer.kernel_variant() == "render_tiles<0, 16>" and renderer.kernel_variant(16) == "render_tiles<0, 16>"
    some = HittableList()
    for k in range(40):
        some.add(Sphere((0.3 * k, 0.0, -1.0), 0.1, Lambertian((0.5, 0.5, 0.5))))
    renderer.set_scene(some.flatten())
    assert renderer.kernel_variant(10) == "render_tiles<2, 10>" and renderer.kernel_variant(11) == "render_tiles<2, 16>"
    monkeypatch.setenv("RT_HIP_FLAT_SCAN", "1")
    renderer.set_scene(some.flatten())
    assert renderer.kernel_variant() == "render_tiles<1, 16>"
    monkeypatch.delenv("RT_HIP_FLAT_SCAN")
    renderer.set_scene(workloads.mesh_scene())
    assert renderer.kernel_variant() == "render_tiles<2, 10>"
    with pytest.raises(_lib.RtError):
        renderer.kernel_variant(17)
    assert len(_lib.load().rt_build_id()) == 16


def _frames_on_gpu(renderer, flat, cams, calls):
    """rt_begin_frames + rt_render_frames (one call per entry of `calls`), then every frame's outputs"""
    renderer.set_scene(flat)
    renderer.begin_frames(calls[0].width, calls[0].height, len(cams))
    for p in calls:
        renderer.render_frames(p, cams)
    out = []
    for f in range(len(cams)):
        renderer.select_frame(f)
        out.append((renderer.read_accum(), renderer.read_primary_ids(), renderer.read_packed(),
                    renderer.read_sample_counts()))
    return out, renderer.ray_count()


@pytest.mark.parametrize("w,h,spp,n_frames", [(600, 337, 1, 10), (61, 43, 3, 4), (120, 67, 70, 2), (40, 24, 2, 1)])
def test_several_frames_in_one_launch(renderer, oracle, tables, w, h, spp, n_frames):
    """rt_render_frames: the reference's run of ten frames (src/raytracer.rs:32,165; the camera one dolly
    step further per frame, src/scene/mod.rs:78-83) in ONE launch at its own size and sample count, and
    smaller cases with partial tiles and units of a few pixels: every frame's sums, ids, packed pixels
    and sample counts are the oracle's for that frame's camera, the segment count is the total"""
    flat = Scene.get_scene().flatten()
    cams = [oracle.scene_camera(f + 5, 0.1 if spp > 1 else 0.0) for f in range(n_frames)]
    renderer.set_camera(oracle.scene_camera(0, 0.0))     # not the camera of any frame: the call must not use it
    p = make_params(w, h, spp=spp)
    got, seg = _frames_on_gpu(renderer, flat, cams, [p])
    total = 0
    for f, cam in enumerate(cams):
        acc, ids, s = oracle.render(flat, cam, tables, p, threads=8)
        total += s
        assert np.array_equal(_bits(got[f][0]), _bits(np.ascontiguousarray(acc[..., :3]))), f
        assert np.array_equal(got[f][1], ids) and np.array_equal(got[f][2], oracle.pack(acc)), f
        assert np.array_equal(got[f][3], acc[..., 3].astype(np.uint32)), f
    assert seg == total


def test_frames_in_one_launch_equal_frames_one_by_one(renderer, oracle, tables):
    """the same through the library alone: progressive sample ranges and a band share rendered for five
    frames at once equal rt_select_frame + rt_set_camera + rt_render per frame; a scene with a box tree
    (its frames are rendered one after the other inside the call) equals the oracle"""
    flat = Scene.get_scene().flatten()
    cams = [oracle.scene_camera(3 * f, 0.1) for f in range(5)]
    w, h = 200, 112
    calls = [make_params(w, h, spp=5, sample_begin=0, sample_count=2, band_begin=1, band_stride=2),
             make_params(w, h, spp=5, sample_begin=2, sample_count=3, band_begin=1, band_stride=2)]
    got, seg = _frames_on_gpu(renderer, flat, cams, calls)
    renderer.begin_frames(w, h, len(cams))
    for f, cam in enumerate(cams):
        renderer.select_frame(f)
        renderer.set_camera(cam)
        for p in calls:
            renderer.render(p)
    for f in range(len(cams)):
        renderer.select_frame(f)
        assert np.array_equal(_bits(got[f][0]), _bits(renderer.read_accum())), f
        assert np.array_equal(got[f][1], renderer.read_primary_ids()), f
        assert np.array_equal(got[f][3], renderer.read_sample_counts()), f
    assert seg == renderer.ray_count()
    from raytracing_amd import workloads
    mesh = workloads.mesh_scene()
    p = make_params(96, 54, spp=2)
    got, seg = _frames_on_gpu(renderer, mesh, cams[:3], [p])
    total = 0
    for f, cam in enumerate(cams[:3]):
        acc, ids, s = oracle.render(mesh, cam, tables, p, threads=8)
        total += s
        assert np.array_equal(_bits(got[f][0]), _bits(np.ascontiguousarray(acc[..., :3]))), f
        assert np.array_equal(got[f][1], ids), f
    assert seg == total
    from raytracing_amd import _lib
    with pytest.raises(_lib.RtError):
        renderer.render_frames(p, cams)            # more cameras than frames
    with pytest.raises(_lib.RtError):
        renderer.select_frame(3)
    # fewer cameras than frames: the first two frames are rendered again (added to), the third is not
    renderer.select_frame(2)
    third = renderer.read_sample_counts()
    renderer.render_frames(p, cams[:2])
    renderer.select_frame(0)
    assert int(renderer.read_sample_counts().max()) == 4
    renderer.select_frame(2)
    assert np.array_equal(renderer.read_sample_counts(), third)
    for w_, h_, n_ in ((8, 8, 0), (8, 8, 4097), (65536, 65536, 1), (40000, 40000, 3)):
        with pytest.raises(_lib.RtError) as e:
            renderer.begin_frames(w_, h_, n_)
        assert e.value.code == _lib.RT_E_BADARG
    renderer.begin_frame(8, 8)                     # back to one frame for the tests that follow


@pytest.mark.parametrize("unit_px", [1, 2, 4, 8, 16, 32, 64])
def test_every_work_unit_size_gives_the_same_picture(renderer, oracle, tables, unit_px):
    """the launch hands out parts of 8x8 tiles (1 to 64 pixels x the samples of the launch) to
    the waves; whatever the size, and with a frame whose right and top tiles are partial
    (61 x 43), the sums are those of the oracle: 3 samples (units of a few items, several in
    flight per wave), 64 and 70 samples (a second launch of 6), two bands out of three, and a
    depth budget of 0 (units with nothing to trace)"""
    import os
    flat = Scene.get_scene().flatten()
    cam = Scene.get_camera(aperture=0.1)
    w, h = 61, 43
    cases = [[make_params(w, h, spp=3)], [make_params(w, h, spp=64)], [make_params(w, h, spp=70)],
             [make_params(w, h, spp=4, band_begin=g, band_stride=3) for g in (0, 2)],
             [make_params(w, h, spp=2, max_bounces=0)]]
    want = [render_oracle(oracle, tables, flat, cam, c) for c in cases]
    os.environ["RT_HIP_UNIT_PX"] = str(unit_px)
    try:
        got = [render_gpu(renderer, flat, cam, c) for c in cases]
    finally:
        del os.environ["RT_HIP_UNIT_PX"]
    for g, o in zip(got, want):
        assert_same(g, o, oracle)


@pytest.mark.parametrize("staged", [False, True])
@pytest.mark.parametrize("n_ctx,w,h", [(1, 61, 43), (2, 120, 67), (3, 61, 43), (5, 64, 64), (8, 33, 9)])
def test_gather_of_several_contexts(renderer, oracle, tables, n_ctx, w, h, staged, monkeypatch):
    """rt_gather_packed: n contexts (here all on GPU 0) render the bands b % n == g of one frame; the
    library quantises each share, collects the blocks on the first context's GPU and returns the
    picture in display order.  More contexts than bands (8 on a 9-row frame) leave some with nothing."""
    from raytracing_amd import Renderer, gather_packed
    if staged:
        # the blocks of the other contexts go through a staging block and a peer copy, as between GPUs
        monkeypatch.setenv("RT_HIP_GATHER_STAGED", "1")
    flat = Scene.get_scene().flatten()
    cam = Scene.get_camera(aperture=0.1)
    want = render_oracle(oracle, tables, flat, cam, [make_params(w, h, spp=2)])
    group = [Renderer(0, tables) for _ in range(n_ctx)]
    try:
        for rep in range(2):            # the second gather reuses the buffers of the first
            for g, r in enumerate(group):
                r.set_scene(flat); r.set_camera(cam); r.begin_frame(w, h)
                r.render(make_params(w, h, spp=2, band_begin=g, band_stride=n_ctx))
            got = gather_packed(group)
            assert np.array_equal(got, oracle.pack(want[0])), rep
        assert sum(r.ray_count() for r in group) == want[2]
        with pytest.raises(_lib.RtError):
            gather_packed(group + [group[0]])        # a context listed twice
    finally:
        for r in group:
            r.close()


def test_band_partition(renderer, oracle, tables):
    """the multi-GPU split: bands b % 3 == g rendered separately add up to the frame, and
    rt_export_packed_bands lays each share out as the gather expects"""
    flat = Scene.get_scene().flatten()
    cam = Scene.get_camera()
    w, h = 120, 67
    full = render_gpu(renderer, flat, cam, [make_params(w, h, spp=2)])
    parts = [make_params(w, h, spp=2, band_begin=g, band_stride=3) for g in range(3)]
    split = render_gpu(renderer, flat, cam, parts)
    assert np.array_equal(_bits(full[0]), _bits(split[0])) and np.array_equal(full[1], split[1])
    assert np.array_equal(full[2], split[2]) and full[3] == split[3]
    import torch
    from raytracing_amd import multi_gpu
    rows = multi_gpu.block_rows(3, h)
    blocks = []
    for g in range(3):
        blk = torch.full((rows, w), -1, dtype=torch.int32, device="cuda")
        renderer.export_packed_bands(blk.data_ptr(), blk.numel(), g, 3)
        renderer.sync()
        got = blk.cpu().numpy().view(np.uint32)
        assert np.array_equal(got, block_from_packed(full[2], g, 3, h))
        blocks.append(blk)
    frame = multi_gpu.compose(torch.stack(blocks), h).cpu().numpy().view(np.uint32)
    assert np.array_equal(frame, full[2])
    # one band alone on the oracle
    one = make_params(w, h, spp=2, band_begin=5, band_stride=9)
    assert_same(render_gpu(renderer, flat, cam, [one]), render_oracle(oracle, tables, flat, cam, [one]), oracle,
                ids_mask=np.repeat((np.arange(h)[::-1] // 8 == 5)[:, None], w, 1))


@pytest.mark.parametrize("kw", [dict(max_bounces=0), dict(max_bounces=1), dict(max_bounces=2),
                                dict(max_bounces=16), dict(gamma=False), dict(t_min=0.0), dict(t_min=0.25),
                                dict(seed=12345), dict(seed=0xFFFFFFFF)])
def test_parameter_variants(renderer, oracle, tables, kw):
    check(renderer, oracle, tables, Scene.get_scene().flatten(), Scene.get_camera(aperture=0.1),
          [make_params(128, 72, spp=3, **kw)])


def test_large_sample_numbers_wrap_like_the_counter(renderer, oracle, tables):
    """sample index n = pixel*spp + s exceeds 2^32 for late pixels when spp is huge; only
    n mod 32 matters because the u32 counter wraps at a multiple of 64 (src/denoising/mod.rs:13-17)"""
    p = make_params(96, 54, spp=4000000, sample_begin=3999990, sample_count=3)
    check(renderer, oracle, tables, Scene.get_scene().flatten(), Scene.get_camera(), [p])


# ---------------------------------------------------------------- random scenes
def random_scene(rng, n_prims, kinds=("sphere", "plane", "cube", "tri")):
    s = HittableList()

    def material():
        k = rng.integers(0, 3)
        alb = tuple(rng.uniform(0.05, 1.0, 3))
        if k == 0:
            return Metal(alb, float(rng.uniform(0, 0.6)))
        if k == 1:
            return Lambertian(alb)
        return Dielectric(alb, float(rng.uniform(1.1, 2.0)))

    for _ in range(n_prims):
        kind = kinds[rng.integers(0, len(kinds))]
        c = tuple(rng.uniform(-2.5, 2.5, 3))
        if kind == "sphere":
            r = float(rng.uniform(0.15, 0.9)) * (-1.0 if rng.random() < 0.15 else 1.0)
            s.add(Sphere(c, r, material()))
        elif kind == "plane":
            n = rng.normal(size=3)
            if rng.random() < 0.5:
                n = np.array([0.0, 1.0, 0.0])
            s.add(Plane(tuple(n), (0.0, float(rng.uniform(-3, -1)), 0.0), material()))
        elif kind == "cube":
            s.add(Cube(c, *[float(x) for x in rng.uniform(0.2, 1.5, 3)], material()))
        else:
            a = np.array(c)
            s.add(Triangle(tuple(a), tuple(a + rng.uniform(-1.5, 1.5, 3)), tuple(a + rng.uniform(-1.5, 1.5, 3)),
                           material()))
    return s.flatten()


def random_camera(rng):
    origin = tuple(rng.uniform(-6, 6, 3))
    look_at = tuple(rng.uniform(-1, 1, 3))
    focus = float(np.linalg.norm(np.array(origin) - np.array(look_at)))
    return Camera.new(float(rng.uniform(20, 80)), float(rng.uniform(1.0, 2.0)), float(rng.uniform(0, 0.3)),
                      focus, origin, look_at, (0.0, 1.0, 0.0))


@pytest.mark.parametrize("seed", range(12))
def test_random_mixed_scenes(renderer, oracle, tables, seed):
    """all primitive kinds and materials in random list order, incl. negative radii,
    glass cubes (front_face inherited from the previous accepted hit) and rays that
    start inside boxes"""
    rng = np.random.default_rng(1000 + seed)
    flat = random_scene(rng, int(rng.integers(1, 24)))
    cam = random_camera(rng)
    check(renderer, oracle, tables, flat, cam, [make_params(96, 64, spp=4, seed=seed)])


@pytest.mark.parametrize("kind", ["sphere", "plane", "cube", "tri"])
def test_single_kind_scenes(renderer, oracle, tables, kind):
    rng = np.random.default_rng({"sphere": 11, "plane": 22, "cube": 33, "tri": 44}[kind])
    flat = random_scene(rng, 9, kinds=(kind,))
    check(renderer, oracle, tables, flat, random_camera(rng), [make_params(80, 48, spp=3)])


def test_empty_scene_is_all_sky(renderer, oracle, tables):
    flat = HittableList().flatten()
    gpu = render_gpu(renderer, flat, Scene.get_camera(), [make_params(64, 40, spp=2)])
    assert (gpu[1] == -1).all() and gpu[3] == 64 * 40 * 2
    assert_same(gpu, render_oracle(oracle, tables, flat, Scene.get_camera(), [make_params(64, 40, spp=2)]), oracle)


def test_camera_inside_glass_cube(renderer, oracle, tables):
    """rays start inside a Dielectric box: Cube::hit reports t = tmin < 0 (src/geometry/mod.rs:205-209)"""
    s = HittableList()
    s.add(Sphere((0.0, 0.0, -3.0), 1.0, Lambertian((0.8, 0.2, 0.2))))
    s.add(Cube((0.0, 0.0, 0.0), 2.0, 2.0, 2.0, Dielectric((0.9, 0.9, 0.9), 1.5)))
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -2.0, 0.0), Metal((0.7, 0.7, 0.7), 0.3)))
    cam = Camera.new(60.0, 1.5, 0.0, 1.0, (0.1, 0.2, 0.3), (0.0, 0.0, -3.0), (0.0, 1.0, 0.0))
    check(renderer, oracle, tables, s.flatten(), cam, [make_params(96, 64, spp=4)])


def test_degenerate_rays_and_nans(renderer, oracle, tables):
    """camera in the ground plane looking along it: 0/0 plane hits (NaN t counts as a hit,
    src/geometry/mod.rs:114-118), infinite slab bounds, NaN colours quantise to 0"""
    s = HittableList()
    s.add(Plane((0.0, 1.0, 0.0), (0.0, 0.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
    s.add(Cube((0.0, 0.0, -3.0), 1.0, 1.0, 1.0, Lambertian((0.2, 0.9, 0.2))))
    s.add(Sphere((2.0, 0.0, -3.0), 0.0, Metal((0.9, 0.9, 0.9), 0.0)))          # zero radius: divides by 0
    cam = _lib.rt_camera()
    cam.origin[:] = (0.0, 0.0, 0.0)
    cam.lower_left_corner[:] = (-2.0, -1.0, -2.0)
    cam.horizontal[:] = (4.0, 0.0, 0.0)
    cam.vertical[:] = (0.0, 2.0, 0.0)
    cam.u[:] = (1.0, 0.0, 0.0)
    cam.v[:] = (0.0, 1.0, 0.0)
    cam.lens_radius = 0.0
    tb = dict(tables)
    tb["disc"] = np.zeros(128, dtype=f32)        # no jitter: the middle row looks exactly along the plane
    renderer.set_tables(tb)
    try:
        p = make_params(65, 33, spp=2)
        gpu = render_gpu(renderer, s.flatten(), cam, [p])
        orc = render_oracle(oracle, tb, s.flatten(), cam, [p])
        a, b = gpu[0], orc[0][..., :3]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        ok = ~np.isnan(a)
        assert np.array_equal(_bits(a)[ok], _bits(b)[ok])
        assert np.array_equal(gpu[1], orc[1]) and gpu[3] == orc[2]
        assert np.array_equal(gpu[2], oracle.pack(orc[0]))
    finally:
        renderer.set_tables(tables)


@pytest.mark.parametrize("thin", [0.0, 1e-30, 1e-20, 1e-9])
def test_plates_whose_normal_saturates_the_integer_conversion(renderer, oracle, tables, thin):
    """Cube::hit's normal is `(1.0001 * (p - c) / half_width) as i32` per axis (src/geometry/mod.rs:211-215).
    On a plate of (next to) no thickness the quotient is far beyond the i32 range, infinite, or 0/0:
    Rust's cast saturates and turns NaN into 0."""
    s = HittableList()
    s.add(Cube((0.0, 0.0, -3.0), thin, 2.0, 2.0, Lambertian((0.8, 0.3, 0.3))))
    s.add(Cube((-1.5, 0.5, -2.0), 1.5, thin, 1.5, Metal((0.8, 0.8, 0.8), 0.1)))
    s.add(Cube((1.5, 0.0, -2.5), 1.0, 1.0, thin, Dielectric((0.9, 0.9, 1.0), 1.5)))
    s.add(Cube((0.5, -0.5, -1.5), thin, thin, 1.0, Lambertian((0.2, 0.8, 0.2))))
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
    cam = Camera.new(60.0, 1.5, 0.0, 1.0, (0.7, 0.4, 1.0), (0.0, 0.0, -2.5), (0.0, 1.0, 0.0))
    p = make_params(96, 64, spp=4)
    gpu = render_gpu(renderer, s.flatten(), cam, [p])
    orc = render_oracle(oracle, tables, s.flatten(), cam, [p])
    a, b = gpu[0], orc[0][..., :3]
    assert np.array_equal(np.isnan(a), np.isnan(b))
    ok = ~np.isnan(a)
    assert np.array_equal(_bits(a)[ok], _bits(b)[ok])
    assert np.array_equal(gpu[1], orc[1]) and gpu[3] == orc[2]
    assert np.array_equal(gpu[2], oracle.pack(orc[0]))
    assert (gpu[1] >= 0).sum() > 500 and set(np.unique(gpu[1])) >= {0, 1, 2}     # the plates are in the picture


def test_many_spheres_small_frame(renderer, oracle, tables):
    """configs[3]'s scene (10,000 spheres + plane) on a frame the oracle can finish"""
    wl = workloads.workload("c4")
    check(renderer, oracle, tables, wl["flat"], wl["camera"], [make_params(96, 54, spp=2)])


@pytest.mark.parametrize("max_bounces", [10, 11, 16])
def test_tree_scenes_on_both_sides_of_the_shallow_depth_budget(renderer, oracle, tables, max_bounces):
    """scenes with a box tree have one kernel for up to 10 bounces (the reference's budget,
    src/raytracer.rs:132) and one for up to 16, with less of the tree in LDS"""
    wl = workloads.workload("c4")
    check(renderer, oracle, tables, wl["flat"], wl["camera"], [make_params(96, 54, spp=2, max_bounces=max_bounces)])
    wl = workloads.workload("c3")
    check(renderer, oracle, tables, wl["flat"], wl["camera"], [make_params(96, 54, spp=2, max_bounces=max_bounces)])


def test_mesh_scene_small_frame(renderer, oracle, tables):
    """configs[2] stand-in: default list + 960 triangles"""
    wl = workloads.workload("c3")
    check(renderer, oracle, tables, wl["flat"], wl["camera"], [make_params(160, 90, spp=2)])


# ------------------------------------------------------------------ error behaviour
def test_error_codes(renderer, tables):
    lib = _lib.load()
    ctx = renderer._ctx
    renderer.set_scene(Scene.get_scene().flatten())
    renderer.set_camera(Scene.get_camera())
    renderer.begin_frame(64, 36)
    bad = [make_params(64, 37), make_params(64, 36, band_begin=2, band_stride=2),
           make_params(64, 36, band_stride=0), make_params(64, 36, max_bounces=17),
           make_params(64, 36, spp=4, sample_begin=3, sample_count=2)]
    for p in bad:
        assert lib.rt_render(ctx, C.byref(p)) == _lib.RT_E_BADARG
        assert lib.rt_last_error(ctx)
    assert lib.rt_render(ctx, None) == _lib.RT_E_BADARG
    assert lib.rt_begin_frame(ctx, 1, 10) == _lib.RT_E_BADARG
    buf = np.zeros(10, dtype=np.uint32)
    assert lib.rt_read_packed(ctx, buf.ctypes.data_as(C.POINTER(C.c_uint32)), 10) == _lib.RT_E_BADARG
    # malformed scenes are refused and the previous scene stays usable
    from raytracing_amd import flat_scene_from_arrays
    dup = flat_scene_from_arrays(sph_cx=[0, 0], sph_cy=[0, 0], sph_cz=[0, 0], sph_r=[1, 1], sph_mat=[0, 0],
                                 sph_id=[1, 1], mat_kind=[1], mat_r=[1], mat_g=[1], mat_b=[1], mat_param=[0])
    assert lib.rt_upload_scene(ctx, C.byref(dup.struct)) == _lib.RT_E_BADARG
    badmat = flat_scene_from_arrays(sph_cx=[0], sph_cy=[0], sph_cz=[0], sph_r=[1], sph_mat=[1], sph_id=[0],
                                    mat_kind=[7], mat_r=[1], mat_g=[1], mat_b=[1], mat_param=[0])
    assert lib.rt_upload_scene(ctx, C.byref(badmat.struct)) == _lib.RT_E_BADARG
    renderer.render(make_params(64, 36))
    assert renderer.ray_count() >= 64 * 36
    # a fresh context refuses to render before its inputs are set
    ctx2 = C.c_void_p()
    assert lib.rt_create(0, C.byref(ctx2)) == 0
    try:
        p = make_params(64, 36)
        assert lib.rt_render(ctx2, C.byref(p)) == _lib.RT_E_STATE
        assert lib.rt_pack(ctx2) == _lib.RT_E_STATE
    finally:
        lib.rt_destroy(ctx2)
