#!/usr/bin/env python3
"""Development aid: kernel time of the reference's ten 600 x 337 frames at one sample in one
rt_render_frames launch, and of one frame alone, for the library named by RT_HIP_LIBRARY
(RT_HIP_GROUPS=n in the environment fixes the number of workgroups of the launches)."""
import os
import sys
import zlib
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd import Renderer, load_tables, make_params, workloads

wl = workloads.workload("c1")
w, h, cams = wl["width"], wl["height"], wl["cameras"]
p = make_params(w, h, spp=1)
with Renderer(0, load_tables()) as r:
    r.set_scene(wl["flat"]); r.set_camera(cams[0]); r.begin_frames(w, h, len(cams))
    r.render_frames(p, cams); r.sync()
    many, one = [], []
    for k in range(8):
        r.begin_frames(w, h, len(cams))
        r.timer_start(); r.render_frames(p, cams); many.append(r.timer_stop())
    seg = r.ray_count()
    crc = 0
    for f in range(len(cams)):
        r.select_frame(f)
        crc = zlib.crc32(r.read_packed().tobytes(), crc)
    r.select_frame(0)
    for k in range(8):
        r.timer_start(); r.render(p); one.append(r.timer_stop())
print("%s groups %s: 10 frames best %.4f ms mean %.4f ms (%.1f Mrays/s), one frame best %.4f ms, crc %08x"
      % (os.path.basename(os.environ.get("RT_HIP_LIBRARY", "default")), os.environ.get("RT_HIP_GROUPS", "auto"),
         min(many), sum(many) / len(many), seg / min(many) / 1e3, min(one), crc))
