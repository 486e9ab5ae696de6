#!/usr/bin/env python3
"""Development aid: run one render step with a library built with -DRTK_STATS
and print the wave-level event counters.  usage: stats_probe.py <lib.so> [workload] [spp]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["RT_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])
from raytracing_amd import Renderer, load_tables, make_params, workloads  # noqa: E402

name = sys.argv[2] if len(sys.argv) > 2 else "c4"
sps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
wl = workloads.workload(name)
r = Renderer(0, load_tables())
r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(wl["width"], wl["height"])
r.timer_start()
r.render(make_params(wl["width"], wl["height"], spp=wl["spp"], sample_begin=0, sample_count=sps))
ms = r.timer_stop()
out = (C.c_uint64 * 8)()
r._lib.rtdbg_read_counters.argtypes = [C.c_void_p, C.POINTER(C.c_uint64)]
assert r._lib.rtdbg_read_counters(r._ctx, out) == 0
names = ["segments", "wave_scans", "box_trips", "leaf_trips", "exact_lane0", "box_lane_visits", "leaf_lane_visits", "exact_wave_iters"]
v = list(out)
print("ms", round(ms, 2), dict(zip(names, v)))
if v[1]:
    print("lane utilisation %.3f" % (v[0] / (64.0 * v[1])))
if v[1] and v[5]:
    print("per wave-scan: box trips %.1f (lanes busy %.1f), leaf trips %.1f (lanes busy %.1f), exact-loop iterations %.1f"
          % (v[2] / v[1], v[5] / max(1, v[2]), v[3] / v[1], v[6] / max(1, v[3]), v[7] / v[1]))
    print("per ray: box visits %.2f, leaf visits %.2f, exact tests %.2f" % (v[5] / v[0], v[6] / v[0], 64.0 * v[4] / v[0]))
