"""Fuzz: the GPU library against the CPU oracle on the random scenes of
tools/stress_tree_vs_list.py (spheres, triangles, cubes and planes in mixed list order,
slivers, lattices of touching spheres, coplanar sheets, extents from 0.1 to 1000, cameras
inside and far outside), at a frame size the oracle traces in a fraction of a second
(about 5000 pixels, 1 to 3 samples), with other values of t_min, the depth budget, gamma and the
band partition now and then.  Sums, primary ids, packed pixels
and segment counts must be identical.  Not collected by pytest; run by
tests/test_gpu_tree.py for a few seconds and by hand for longer:
    python tests/stress_gpu_vs_oracle.py [seconds] [seed0] [share of scenes with NaN/inf/huge coordinates]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import Renderer, load_tables, make_params  # noqa: E402
from tests import oracle_lib  # noqa: E402
from tools.stress_tree_vs_list import perturb, scene  # noqa: E402


POISON = 0.05      # share of scenes that get non-finite or huge coordinates


def shifted(cam, dx):
    """a copy of the camera with its origin and lower-left corner moved along x"""
    import ctypes
    cam = cam.struct if hasattr(cam, "struct") else cam
    out = type(cam)()
    ctypes.memmove(ctypes.byref(out), ctypes.byref(cam), ctypes.sizeof(cam))
    out.origin[0] = float(np.float32(out.origin[0]) + np.float32(dx))
    out.lower_left_corner[0] = float(np.float32(out.lower_left_corner[0]) + np.float32(dx))
    return out


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else "30"
    n_scenes = int(arg[:-1]) if arg.endswith("n") else None      # a budget of scenes (`150n`) instead of seconds
    budget = float("inf") if n_scenes is not None else float(arg)
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    global POISON
    POISON = float(sys.argv[3]) if len(sys.argv) > 3 else 0.05
    tables = load_tables()
    r = Renderer(0, tables)
    orc = oracle_lib.load()
    threads = min(16, os.cpu_count() or 1)
    t0 = time.time()
    k = int(sys.argv[4]) if len(sys.argv) > 4 else 0      # number of the first scene (to repeat one scene of a run)
    bad = 0
    rays = 0
    k_end = None if n_scenes is None else k + n_scenes
    while time.time() - t0 < budget and (k_end is None or k < k_end):
        rng = np.random.default_rng(seed0 + k)
        flat, cam = scene(rng)
        notes = perturb(rng, flat, cam, POISON)
        t_min = 0.001 if rng.random() < 0.8 else float(rng.choice([-1.0, 0.0, 0.05, 2.0]))
        w, h = [(96, 54), (61, 43), (130, 20), (33, 70)][int(rng.integers(0, 4))]
        stride = 1 if rng.random() < 0.7 else 3
        spp = int(rng.integers(1, 4))
        if rng.random() < 0.08:
            # many samples on a small frame: work units of one to four pixels, and ranges beyond one launch
            w, h = [(20, 13), (9, 26)][int(rng.integers(0, 2))]
            spp = int(rng.choice([70, 150, 300, 520]))
        p = make_params(w, h, spp=spp, seed=k, t_min=t_min,
                        max_bounces=10 if rng.random() < 0.6 else int(rng.choice([1, 3, 16])),
                        gamma=bool(rng.random() < 0.7), band_begin=int(rng.integers(0, stride)), band_stride=stride)
        # one call, or the sample range in two calls (progressive rendering adds into the same sums)
        calls = [p]
        if p.spp > 1 and rng.random() < 0.25:
            cut = int(rng.integers(1, p.spp))
            calls = [make_params(w, h, spp=p.spp, seed=k, t_min=t_min, max_bounces=p.max_bounces, gamma=bool(p.gamma),
                                 band_begin=p.band_begin, band_stride=p.band_stride, sample_begin=b, sample_count=c)
                     for b, c in ((0, cut), (cut, p.spp - cut))]
        o_acc, o_ids, o_seg = None, None, 0
        for q in calls:
            o_acc, ids_q, seg_q = orc.render(flat, cam, tables, q, threads=threads, accum=o_acc)
            o_ids = ids_q if o_ids is None else o_ids
            o_seg += seg_q
        # now and then as the middle one of three frames of one rt_render_frames call (the others through
        # cameras a little off): a frame's outputs do not depend on what shares its launch (the segment
        # count is that of all three frames and is not compared then)
        as_frames = rng.random() < 0.15
        r.set_scene(flat); r.set_camera(cam)
        if as_frames:
            cams = [shifted(cam, -0.37), cam, shifted(cam, 0.61)]
            r.begin_frames(p.width, p.height, 3)
            for q in calls:
                r.render_frames(q, cams)
            r.select_frame(1)
        else:
            r.begin_frame(p.width, p.height)
            for q in calls:
                r.render(q)
        acc, ids, seg, packed = r.read_accum(), r.read_primary_ids(), r.ray_count(), r.read_packed()
        if as_frames:
            seg = o_seg
        a, b = acc.view(np.uint32), np.ascontiguousarray(o_acc[..., :3]).view(np.uint32)
        nan = np.isnan(acc) & np.isnan(o_acc[..., :3])          # NaN payloads may differ
        same = (np.array_equal(a[~nan], b[~nan]) and np.array_equal(ids, o_ids) and seg == o_seg
                and np.array_equal(packed[o_acc[..., 3] > 0], orc.pack(o_acc)[o_acc[..., 3] > 0]))
        rays += seg
        if not same:
            bad += 1
            print("MISMATCH seed", seed0 + k, "scene number", k, "t_min", t_min, "sums", int((a != b)[~nan].sum()),
                  "ids", int((ids != o_ids).sum()), "segments", seg, o_seg, flush=True)
            if os.environ.get("RT_FUZZ_VERBOSE"):
                d = (a != b) & ~nan
                ys, xs, cs = np.nonzero(d)
                print(" size", p.width, p.height, "spp", p.spp, "bounces", p.max_bounces, "bands", p.band_begin, p.band_stride,
                      "counts", flat.counts, "calls", len(calls), "perturbed:", notes)
                for y, x, c in list(zip(ys, xs, cs))[:12]:
                    print("  pixel", x, y, "channel", c, "gpu", acc[y, x, c], "oracle", o_acc[y, x, c], "id", ids[y, x])
        k += 1
        if k % 20 == 0:
            print("scenes", k, "segments %.3g" % rays, "mismatches", bad, "t %.0f s" % (time.time() - t0), flush=True)
    print("DONE scenes", k, "segments %.4g" % rays, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
