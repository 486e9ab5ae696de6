#!/usr/bin/env python3
"""Development aid: kernel time of a scene with a middling sphere run (48) next to a long triangle
run (3000) at 1080p, 64 spp, and a CRC of the picture, for the library named by RT_HIP_LIBRARY.
With --spheres-only the triangles are left out: a scene whose only long run is 48 spheres, which the
library walks through a small box tree (render_tiles<2, 10>) and, with RT_HIP_FLAT_SCAN=1 in the
environment, scans in list order through the rejection test (render_tiles<1, 16>).
usage: [RT_HIP_LIBRARY=lib.so] [RT_HIP_FLAT_SCAN=1] mixed_scene_probe.py [--spheres-only]"""
import os
import sys
import zlib
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), '..'))
from raytracing_amd import Camera, Renderer, flat_scene_from_arrays, load_tables, make_params, HittableList, Sphere, Triangle, Plane, Lambertian, Metal
rng = np.random.default_rng(5)
s = HittableList()
s.add(Plane((0.0, 1.0, 0.0), (0.0, -1.0, 0.0), Lambertian((0.5, 0.5, 0.5))))
for _ in range(48):
    c = rng.uniform(-4, 4, 3); c[1] = abs(c[1]) * 0.3
    s.add(Sphere(tuple(c), float(rng.uniform(0.1, 0.5)), Metal(tuple(rng.uniform(0.3, 0.9, 3)), 0.1)))
for _ in range(0 if "--spheres-only" in sys.argv else 3000):
    a = rng.uniform(-5, 5, 3); a[1] = abs(a[1]) * 0.5
    s.add(Triangle(tuple(a), tuple(a + rng.uniform(-0.3, 0.3, 3)), tuple(a + rng.uniform(-0.3, 0.3, 3)), Lambertian(tuple(rng.uniform(0.2, 0.9, 3)))))
flat = s.flatten()
cam = Camera.new(40.0, 16 / 9, 0.0, 1.0, (0.0, 3.0, 10.0), (0.0, 0.5, 0.0), (0.0, 1.0, 0.0))
w, h = 1920, 1080
with Renderer(0, load_tables()) as r:
    r.set_scene(flat); r.set_camera(cam); r.begin_frame(w, h)
    r.render(make_params(w, h, spp=64, sample_begin=0, sample_count=64)); r.sync()
    r.begin_frame(w, h)
    ms = []
    for k in range(4):
        r.timer_start(); r.render(make_params(w, h, spp=64, sample_begin=0, sample_count=64)); ms.append(r.timer_stop())
    print(os.environ.get("RT_HIP_LIBRARY", "default"), r.kernel_variant(), "48 spheres + %d triangles 1080p 64 spp: best %.3f ms" % (flat.counts["n_tris"], min(ms)),
          "crc %08x" % zlib.crc32(r.read_packed().tobytes()))
