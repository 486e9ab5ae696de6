"""Time of rt_upload_scene (validation, run detection, tree build on the host, copies) for the
large workloads.  usage: upload_time.py"""
import sys, time
sys.path.insert(0, '.')
from raytracing_amd import Renderer, load_tables, workloads
r = Renderer(0, load_tables())
for name in ("c4", "c5"):
    wl = workloads.workload(name)
    for k in range(2):
        t0 = time.perf_counter(); r.set_scene(wl["flat"]); r.sync(); dt = time.perf_counter() - t0
        print(name, "rt_upload_scene %.1f ms" % (dt * 1e3))
