"""Predicts multi-GPU strong scaling on one GPU: renders the interleaved band share of every rank
of an N-rank job in turn and compares the slowest share with the whole frame.
A share is timed as one rt_render call over min(spp, 512) samples (one launch), like a step of bench.py.
usage: band_balance.py [workload] [ranks...]   (SPS=64 in the environment: 64-sample launches instead)"""
import os
import sys
import numpy as np
sys.path.insert(0, ".")
from raytracing_amd import Renderer, load_tables, make_params, workloads

name = sys.argv[1] if len(sys.argv) > 1 else "c4"
ranks = [int(x) for x in sys.argv[2:]] or [2, 4, 8]
wl = workloads.workload(name)
w, h, spp = wl["width"], wl["height"], wl["spp"]
sps = min(spp, int(os.environ.get("SPS", "512")))
r = Renderer(0, load_tables())
r.set_scene(wl["flat"]); r.set_camera(wl["camera"]); r.begin_frame(w, h)


def timed(begin, stride):
    p = make_params(w, h, spp=spp, sample_begin=0, sample_count=sps, band_begin=begin, band_stride=stride)
    best = 1e9
    for _ in range(3):
        r.timer_start(); r.render(p); best = min(best, r.timer_stop())
    return best


timed(0, 1)
full = timed(0, 1)
print("%s whole frame, %d spp per call: %.2f ms" % (name, sps, full))
for n in ranks:
    t = [timed(g, n) for g in range(n)]
    print("ranks %d: shares %.2f..%.2f ms (mean %.2f); predicted speed-up %.2f of %d (efficiency %.3f)"
          % (n, min(t), max(t), np.mean(t), full / max(t), n, full / max(t) / n))
