"""Development aid: when the waves of one render launch start and end (library built with
-DRTK_WAVE_ENDS writes the times into the staging ring).  usage: wave_ends.py <lib.so> [workload] [stride]"""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, ".")
os.environ["RT_HIP_LIBRARY"] = os.path.abspath(sys.argv[1])
from raytracing_amd import Renderer, load_tables, make_params, workloads
name = sys.argv[2] if len(sys.argv) > 2 else "c4"
stride = int(sys.argv[3]) if len(sys.argv) > 3 else 8
wl = workloads.workload(name); w, h, spp = wl["width"], wl["height"], wl["spp"]
r = Renderer(0, load_tables()); r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(w, h)
p = make_params(w, h, spp=spp, sample_begin=0, sample_count=64, band_begin=1 % stride, band_stride=stride)
r.render(p); r.sync()
r.timer_start(); r.render(p); ms = r.timer_stop()
# the staging buffer's address: read it back through the debug hook of the library
lib = r._lib
lib.rtdbg_stage_ptr.restype = C.c_void_p; lib.rtdbg_stage_ptr.argtypes = [C.c_void_p]
ptr = lib.rtdbg_stage_ptr(r._ctx)
n = int(os.environ.get("WAVES", "4096"))   # waves of the launch (8192 for a scene without a tree)
STRIDE = 24576 // 8          # every wave logs at the start of its 24 KiB of the staging ring
buf = (C.c_uint64 * (STRIDE * n))()
import ctypes
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpy(buf, C.c_void_p(ptr), 8 * STRIDE * n, 2)
raw = np.array(buf, dtype=np.uint64).reshape(n, STRIDE)[:, :2].copy()
seg = (raw[:, 0] >> np.uint64(40)).astype(np.float64)          # lane-segments the wave traced
raw[:, 0] &= np.uint64((1 << 40) - 1); raw[:, 1] &= np.uint64((1 << 40) - 1)
t = raw.astype(np.float64) * 1e-5   # ms
t0 = t[:, 0].min(); b = t[:, 0] - t0; e = t[:, 1] - t0
print("%s stride %d: kernel %.2f ms; waves start %.3f .. %.3f ms; end: first %.2f, median %.2f, 90%% %.2f, 99%% %.2f, last %.2f ms"
      % (name, stride, ms, b.min(), b.max(), e.min(), np.median(e), np.percentile(e, 90), np.percentile(e, 99), e.max()))
print("  waves still running at last-1.0/0.5/0.25/0.1 ms:", [(int((e > e.max() - x).sum())) for x in (1.0, 0.5, 0.25, 0.1)])
late = e > np.percentile(e, 99)
print("  lane-segments per wave: all %.0f +- %.0f; the 1 %% that end last %.0f +- %.0f" % (seg.mean(), seg.std(), seg[late].mean(), seg[late].std()))
print("  correlation of end time with lane-segments: %.2f" % np.corrcoef(e, seg)[0, 1])
