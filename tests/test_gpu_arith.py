"""The short division and square-root sequences of the render kernel against `/` and sqrtf,
on the GPU, over random operands of the ranges the kernel uses them in (rt_check_arith)."""
import pytest

from raytracing_amd import Renderer, load_tables

pytestmark = pytest.mark.gpu


def test_short_sequences_give_the_bits_of_division_and_sqrt():
    with Renderer(0, load_tables()) as r:
        total = 0
        for seed in range(4):
            bad_div, bad_sqrt = r.check_arith(1 << 28, seed)     # 2.7e8 quotients and roots each
            assert (bad_div, bad_sqrt) == (0, 0), (seed, bad_div, bad_sqrt)
            total += 1 << 28
        assert total >= 10 ** 9


# ---------------------------------------------------------------- the ranges around the short sequences
def _camera(origin, scale):
    """a pinhole camera at `origin` looking down -z whose ray directions are `scale` long"""
    from raytracing_amd import _lib
    cam = _lib.rt_camera()
    cam.origin[:] = origin
    cam.lower_left_corner[:] = (origin[0] - scale, origin[1] - 0.6 * scale, origin[2] - scale)
    cam.horizontal[:] = (2.0 * scale, 0.0, 0.0)
    cam.vertical[:] = (0.0, 1.2 * scale, 0.0)
    cam.u[:] = (1.0, 0.0, 0.0)
    cam.v[:] = (0.0, 1.0, 0.0)
    cam.lens_radius = 0.0
    return cam


@pytest.mark.parametrize("scale", [2.0 ** -13, 2.0 ** -12, 2.0 ** -11, 1.0, 2.0 ** 11, 2.0 ** 12, 2.0 ** 13])
@pytest.mark.parametrize("t_min", [2.0 ** -21, 2.0 ** -20, 1e-3])
def test_direction_lengths_and_t_min_around_the_limits(renderer, oracle, tables, scale, t_min):
    """|d|^2 on both sides of 2^-24 and 2^24 and t_min on both sides of 2^-20: the wave takes the short
    sequences on one side and `/`, sqrtf on the other, and must agree with the oracle on both"""
    import numpy as np
    from raytracing_amd import Dielectric, HittableList, Lambertian, Metal, Plane, Sphere, make_params
    from tests.test_gpu_parity import check
    s = HittableList()
    s.add(Sphere((0.0, 0.0, -3.0), 1.0, Metal((0.8, 0.7, 0.6), 0.1)))
    s.add(Sphere((1.5, 0.2, -2.5), 0.5, Dielectric((1.0, 1.0, 1.0), 1.5)))
    s.add(Sphere((-1.5, 0.0, -2.0), -0.4, Lambertian((0.3, 0.6, 0.3))))
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
    check(renderer, oracle, tables, s.flatten(), _camera((0.0, 0.0, 1.0), scale),
          [make_params(48, 32, spp=4, t_min=np.float32(t_min))])


@pytest.mark.parametrize("far", [2.0 ** 29, 2.0 ** 30, 2.0 ** 30 * 1.001, 2.0 ** 31])
def test_origins_and_spheres_around_the_coordinate_limit(renderer, oracle, tables, far):
    """ray origins and sphere centres on both sides of 2^30, radii on both sides of 2^-30"""
    import numpy as np
    from raytracing_amd import HittableList, Lambertian, Metal, Sphere, make_params
    from tests.test_gpu_parity import check
    s = HittableList()
    s.add(Sphere((far, 0.0, -far / 4), far / 8, Metal((0.8, 0.7, 0.6), 0.0)))
    s.add(Sphere((far - far / 16, far / 32, far / 8), far / 64, Lambertian((0.3, 0.6, 0.3))))
    s.add(Sphere((far, 0.0, far / 2), float(np.float32(2.0 ** -30) * np.float32(far / 2.0 ** 30)), Metal((0.5, 0.5, 0.5), 0.0)))
    check(renderer, oracle, tables, s.flatten(), _camera((far, 0.0, far / 2), 1.0), [make_params(48, 32, spp=2)])


@pytest.mark.parametrize("flat_dir", [0.0, 2.0 ** -32, 2.0 ** -31, 2.0 ** -30, 2.0 ** -29, 1e-3])
@pytest.mark.parametrize("on_face", [False, True])
def test_cube_slabs_with_tiny_direction_components_and_origins_on_a_face(renderer, oracle, tables, flat_dir, on_face):
    """the six quotients of Cube::hit (src/geometry/mod.rs:158-206): directions whose y component is
    tiny or exactly zero (quotients of 2^30 and more, 0/0 on a face plane), and a camera standing in
    the plane of a cube's face"""
    from raytracing_amd import Cube, Dielectric, HittableList, Lambertian, Metal, Plane, Sphere, _lib, make_params
    from tests.test_gpu_tree import _same_with_nans
    from tests.test_gpu_parity import render_gpu, render_oracle
    s = HittableList()
    s.add(Cube((0.0, 0.0, -3.0), 1.0, 1.0, 1.0, Lambertian((0.7, 0.3, 0.7))))
    s.add(Cube((1.5, 0.25, -2.5), 0.5, 1.5, 0.5, Dielectric((1.0, 1.0, 1.0), 1.5)))
    s.add(Sphere((-1.5, 0.0, -2.0), 0.4, Metal((0.8, 0.8, 0.8), 0.0)))
    s.add(Plane((0.0, 1.0, 0.0), (0.0, -0.5, 0.0), Lambertian((0.5, 0.5, 0.5))))
    origin = (0.5 if on_face else 0.1, 0.5 if on_face else 0.05, 1.0)      # x = 0.5, y = 0.5: faces of the first cube
    cam = _lib.rt_camera()
    cam.origin[:] = origin
    cam.lower_left_corner[:] = (origin[0] - 1.0, origin[1] - flat_dir, origin[2] - 1.0)
    cam.horizontal[:] = (2.0, 0.0, 0.0)
    cam.vertical[:] = (0.0, 2.0 * flat_dir, 0.0)
    cam.u[:] = (1.0, 0.0, 0.0)
    cam.v[:] = (0.0, 1.0, 0.0)
    cam.lens_radius = 0.0
    p = [make_params(48, 32, spp=3)]
    flat = s.flatten()
    _same_with_nans(render_gpu(renderer, flat, cam, p), render_oracle(oracle, tables, flat, cam, p), oracle)


@pytest.mark.parametrize("far", [2.0 ** 29, 2.0 ** 30 * 0.99, 2.0 ** 30 * 1.01, 2.0 ** 32])
def test_cubes_around_the_coordinate_limit(renderer, oracle, tables, far):
    """cube corners on both sides of 2^30"""
    from raytracing_amd import Cube, HittableList, Lambertian, Metal, Sphere, make_params
    from tests.test_gpu_parity import check
    s = HittableList()
    s.add(Cube((far, 0.0, -far / 4), far / 8, far / 8, far / 8, Metal((0.8, 0.7, 0.6), 0.0)))
    s.add(Cube((far - far / 16, far / 32, far / 8), far / 64, far / 32, far / 64, Lambertian((0.3, 0.6, 0.3))))
    s.add(Sphere((far, 0.0, far / 2 - far / 3), far / 16, Lambertian((0.5, 0.5, 0.5))))
    check(renderer, oracle, tables, s.flatten(), _camera((far - far / 64, 0.0, far / 2), 1.0), [make_params(48, 32, spp=2)])
