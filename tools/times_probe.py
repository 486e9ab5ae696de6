#!/usr/bin/env python3
"""Development aid: share of wave time spent inside the closest-hit scan, from a library
built with -DRTK_TIMES.  usage: times_probe.py <lib.so> <workload> <spp>"""
import ctypes as C, os, sys
sys.path.insert(0, '.')
os.environ["RT_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])
from raytracing_amd import Renderer, load_tables, make_params, workloads
wl = workloads.workload(sys.argv[2]); sps = int(sys.argv[3])
r = Renderer(0, load_tables()); r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(wl["width"], wl["height"])
r.render(make_params(wl["width"], wl["height"], spp=wl["spp"], sample_begin=0, sample_count=sps)); r.sync()
r.begin_frame(wl["width"], wl["height"])
r.timer_start(); r.render(make_params(wl["width"], wl["height"], spp=wl["spp"], sample_begin=0, sample_count=sps)); ms = r.timer_stop()
out = (C.c_uint64 * 8)(); r._lib.rtdbg_read_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]; r._lib.rtdbg_read_counters(r._ctx, out)
v = list(out); tot = v[4] + v[7]
print("ms %.2f segments %d: scan %.3f, shade %.3f, parking finished paths %.3f, path colours + unit accounting and sums %.3f, work-item fetch %.3f of the wave time"
      % (ms, v[0], v[4] / tot, (v[7] - v[5] - v[6] - v[3]) / tot, v[5] / tot, v[3] / tot, v[6] / tot))
