"""Fuzz: the tree scan against the list-order scan of the same library on the GPU
(RT_HIP_FLAT_SCAN=1 disables the tree; triangles then run their exact test one by one
and spheres the rejection scan), over random scenes at frame sizes far beyond the CPU
oracle.  Every output must be bit-identical.
usage: stress_tree_vs_list.py [seconds | <count>n] [seed0] [share of scenes with NaN/inf/huge/tiny numbers, default 0]
(a budget written as `80n` is a number of scenes instead of a time)"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
from raytracing_amd import Camera, Renderer, flat_scene_from_arrays, load_tables, make_params  # noqa: E402


def scene(rng):
    style = rng.integers(0, 6)
    n_s = int(rng.choice([0, 40, 70, 300, 2000]))      # 40: the rejection scan of 32..63 spheres
    n_t = int(rng.choice([0, 40, 70, 500, 4000]))
    if n_s + n_t == 0:
        n_s = 500
    ext = float(10.0 ** rng.uniform(-1, 3))            # scene extent 0.1 .. 1000
    off = rng.uniform(-1, 1, 3) * ext * float(rng.choice([0, 0, 1, 100]))
    c = (off + rng.uniform(-ext, ext, (n_s, 3))).astype(np.float32)
    r = (ext * 10.0 ** rng.uniform(-3.5, -0.7, n_s)).astype(np.float32)
    if style == 1:
        r *= np.where(rng.random(n_s) < 0.2, -1, 1).astype(np.float32)
    if style == 2 and n_s:                              # lattice: many exact ties and tangencies
        g = np.stack(np.meshgrid(*[np.arange(13)] * 3), -1).reshape(-1, 3)[:n_s]
        c = (off + (g - 6) * ext / 6).astype(np.float32)
        r = np.full(n_s, ext / 12, dtype=np.float32)
    v0 = (off + rng.uniform(-ext, ext, (n_t, 3))).astype(np.float32)
    sz = (ext * 10.0 ** rng.uniform(-3, -0.3, (n_t, 1)))
    v1 = (v0 + rng.normal(size=(n_t, 3)) * sz).astype(np.float32)
    v2 = (v0 + rng.normal(size=(n_t, 3)) * sz).astype(np.float32)
    if style == 3 and n_t:                              # slivers
        v2 = (v1 + (v1 - v0) * 1e-4 + rng.normal(size=(n_t, 3)) * sz * 1e-5).astype(np.float32)
    if style == 4 and n_t:                              # coplanar sheet: rays graze many at once
        for v in (v0, v1, v2):
            v[:, 1] = off[1]
    # the list: the spheres and the triangles, each cut into up to three runs by cubes and planes
    # put at random positions (a cube may be glass: it inherits front_face from the hit before it)
    n_c, n_p = int(rng.integers(0, 4)), int(rng.integers(0, 3))
    n = n_s + n_t + n_c + n_p
    slots = rng.permutation(2) if n_s and n_t else [0, 1]
    body = []
    for kind in slots:
        body += [("s", k) for k in range(n_s)] if kind == 0 else [("t", k) for k in range(n_t)]
    for k in range(n_c):
        body.insert(int(rng.integers(0, len(body) + 1)), ("c", k))
    for k in range(n_p):
        body.insert(int(rng.integers(0, len(body) + 1)), ("p", k))
    ids = {"s": np.zeros(n_s, np.uint32), "t": np.zeros(n_t, np.uint32), "c": np.zeros(n_c, np.uint32),
           "p": np.zeros(n_p, np.uint32)}
    for pos, (kind, k) in enumerate(body):
        ids[kind][k] = pos
    kinds = rng.integers(0, 3, n).astype(np.int32)
    cc = (off + rng.uniform(-ext, ext, (n_c, 3))).astype(np.float32)
    cs = (ext * 10.0 ** rng.uniform(-1.5, 0.3, (n_c, 3))).astype(np.float32)
    pn = rng.normal(size=(n_p, 3)).astype(np.float32)
    pp = (off + rng.uniform(-ext, ext, (n_p, 3)) * 1.5).astype(np.float32)
    flat = flat_scene_from_arrays(
        sph_cx=c[:, 0], sph_cy=c[:, 1], sph_cz=c[:, 2], sph_r=r, sph_mat=ids["s"], sph_id=ids["s"],
        tri_v0x=v0[:, 0], tri_v0y=v0[:, 1], tri_v0z=v0[:, 2], tri_v1x=v1[:, 0], tri_v1y=v1[:, 1], tri_v1z=v1[:, 2],
        tri_v2x=v2[:, 0], tri_v2y=v2[:, 1], tri_v2z=v2[:, 2], tri_mat=ids["t"], tri_id=ids["t"],
        cb_cx=cc[:, 0], cb_cy=cc[:, 1], cb_cz=cc[:, 2], cb_w=cs[:, 0], cb_h=cs[:, 1], cb_d=cs[:, 2],
        cb_mat=ids["c"], cb_id=ids["c"],
        pl_nx=pn[:, 0], pl_ny=pn[:, 1], pl_nz=pn[:, 2], pl_px=pp[:, 0], pl_py=pp[:, 1], pl_pz=pp[:, 2],
        pl_mat=ids["p"], pl_id=ids["p"],
        mat_kind=kinds, mat_r=rng.uniform(0.2, 1, n), mat_g=rng.uniform(0.2, 1, n), mat_b=rng.uniform(0.2, 1, n),
        mat_param=np.where(kinds == 2, 1.5, rng.uniform(0, 0.4, n)))
    where = rng.integers(0, 3)
    origin = off + (rng.uniform(-1, 1, 3) * ext * (0.3 if where == 0 else 2.5 if where == 1 else 30.0))
    if style == 4:
        origin[1] = off[1] + ext * float(rng.choice([0.0, 1e-3, 0.2]))      # in or just above the sheet
    look = off + rng.uniform(-0.3, 0.3, 3) * ext
    cam = Camera.new(float(rng.uniform(5, 100)), 16 / 9, float(rng.choice([0, 0.05 * ext])),
                     float(np.linalg.norm(origin - look)) + 1e-3, tuple(origin), tuple(look), (0.0, 1.0, 0.0))
    return flat, cam


BAD_VALUES = np.array([np.nan, np.inf, -np.inf, 1e19, -3e25, 1e38, 1e14, -1e16, 8e17, 9e14, -1.1e15], dtype=np.float32)
COORDINATES = (["sph_cx", "sph_cy", "sph_cz", "sph_r", "cb_cx", "cb_cy", "cb_cz", "cb_w", "cb_h", "cb_d",
                "pl_px", "pl_py", "pl_pz"] + ["tri_" + v + c for v in ("v0", "v1", "v2") for c in "xyz"])


def perturb(rng, flat, cam, share):
    """With probability `share`: numbers at which the reference's f32 arithmetic leaves the ordinary
    range.  Single coordinates that are NaN, infinite or huge (such primitives hit every ray with a
    NaN distance, or none; a plane far away sends scattered rays back from there), plane normals
    far from unit length (Plane::hit does not normalise, so scattered directions get that length),
    the whole scene and camera scaled by a power of ten, and - on its own 6 % - a camera whose ray
    directions are extremely long or short or whose origin is so far away that |o|^2 overflows.
    Returns a list of what was done."""
    notes = []
    a = flat.arrays
    if rng.random() < share:
        for key in ("sph_cx", "sph_r", "tri_v0y", "tri_v1x", "tri_v2z", "cb_cx", "pl_py"):
            if len(a[key]) and rng.random() < 0.5:
                at = int(rng.integers(0, len(a[key])))
                a[key][at] = BAD_VALUES[int(rng.integers(0, len(BAD_VALUES)))]
                notes.append((key, at, float(a[key][at])))
        if len(a["pl_nx"]) and rng.random() < 0.4:
            f = np.float32(rng.choice([1e-6, 1e-3, 30.0, 3e3, 3e5]))
            for key in ("pl_nx", "pl_ny", "pl_nz"):
                a[key] *= f
            notes.append(("plane normals times", float(f)))
        if rng.random() < 0.2:
            f = np.float32(rng.choice([1e-22, 1e-17, 1e-12, 1e-6, 1e6, 1e10, 1e13]))
            for key in COORDINATES:
                a[key] *= f
            c = cam.struct
            for field in ("origin", "lower_left_corner", "horizontal", "vertical"):
                v = getattr(c, field)
                v[:] = [float(np.float32(x) * f) for x in v[:]]
            c.lens_radius = float(np.float32(c.lens_radius) * f)
            notes.append(("everything times", float(f)))
    if rng.random() < 0.06:
        c = cam.struct
        scale = float(rng.choice([1e-25, 1e-9, 1e8, 1e20, 1.0]))
        shift = float(rng.choice([0.0, 0.0, 1e17, 3e19, 1e30]))
        o = np.array(c.origin[:], dtype=np.float32)
        llc = np.array(c.lower_left_corner[:], dtype=np.float32)
        o2 = (o + np.float32(shift)).astype(np.float32)
        c.lower_left_corner[:] = [float(x) for x in (o2 + (llc - o) * np.float32(scale)).astype(np.float32)]
        c.origin[:] = [float(x) for x in o2]
        c.horizontal[:] = [float(np.float32(x) * np.float32(scale)) for x in c.horizontal[:]]
        c.vertical[:] = [float(np.float32(x) * np.float32(scale)) for x in c.vertical[:]]
        notes.append(("camera directions times", scale, "origin plus", shift))
    return notes


def main():
    arg = sys.argv[1] if len(sys.argv) > 1 else "60"
    n_scenes = int(arg[:-1]) if arg.endswith("n") else None      # a budget of scenes instead of seconds
    budget = float("inf") if n_scenes is not None else float(arg)
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    share = float(sys.argv[3]) if len(sys.argv) > 3 else 0.0
    r = Renderer(0, load_tables())
    t0 = time.time()
    k = 0
    rays = 0
    bad = 0
    while time.time() - t0 < budget and (n_scenes is None or k < n_scenes):
        rng = np.random.default_rng(seed0 + k)
        flat, cam = scene(rng)
        notes = perturb(rng, flat, cam, share) if share > 0.0 else []
        p = make_params(1280, 720, spp=2, seed=k)
        out = []
        for mode in ("tree", "list"):
            if mode == "list":
                os.environ["RT_HIP_FLAT_SCAN"] = "1"
            try:
                r.set_scene(flat)
            finally:
                os.environ.pop("RT_HIP_FLAT_SCAN", None)
            r.set_camera(cam)
            r.begin_frame(p.width, p.height)
            r.render(p)
            out.append((r.read_accum().view(np.uint32), r.read_primary_ids(), r.ray_count()))
        nan = np.isnan(out[0][0].view(np.float32)) & np.isnan(out[1][0].view(np.float32))     # NaN payloads may differ
        differ = (out[0][0] != out[1][0]) & ~nan
        same = not differ.any() and np.array_equal(out[0][1], out[1][1]) and out[0][2] == out[1][2]
        rays += out[0][2]
        if not same:
            bad += 1
            print("MISMATCH seed", seed0 + k, "pixels", int(differ.any(-1).sum()), "ids", int((out[0][1] != out[1][1]).sum()),
                  "segments", out[0][2], out[1][2], "perturbed:", notes, flush=True)
        k += 1
        if k % 10 == 0:
            print("scenes", k, "segments %.3g" % rays, "mismatches", bad, "t %.0f s" % (time.time() - t0), flush=True)
    print("DONE scenes", k, "segments %.4g" % rays, "mismatches", bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
