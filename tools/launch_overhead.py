"""Fixed cost of a render launch: one launch against several back to back on the stream, for a
1/8 share of the bands.  usage: launch_overhead.py [workload]"""
import sys
sys.path.insert(0, ".")
from raytracing_amd import Renderer, load_tables, make_params, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
wl = workloads.workload(name)
w, h, spp = wl["width"], wl["height"], wl["spp"]
r = Renderer(0, load_tables())
r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(w, h)
p = make_params(w, h, spp=spp, sample_begin=0, sample_count=64, band_begin=0, band_stride=8)
for n in (1, 1, 2, 4, 8):
    best = 1e9
    for _ in range(3):
        r.timer_start()
        for _ in range(n):
            r.render(p)
        best = min(best, r.timer_stop())
    print("%s: %d launches back to back %.2f ms, %.2f ms each" % (name, n, best, best / n))
