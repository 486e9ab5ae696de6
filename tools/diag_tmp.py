import os, sys
import numpy as np
sys.path.insert(0, '.')
from raytracing_amd import Renderer, load_tables, make_params
from tests import oracle_lib
from tools.stress_tree_vs_list import scene
tables = load_tables(); r = Renderer(0, tables); orc = oracle_lib.load()
seed0 = 4900000
for seed in [int(x) for x in sys.argv[1:]]:
    k = seed - seed0
    rng = np.random.default_rng(seed)
    flat, cam = scene(rng)
    a = flat.arrays
    rng.random()
    bad_values = np.array([np.nan, np.inf, -np.inf, 1e19, -3e25, 1e38], dtype=np.float32)
    poison = []
    for key in ("sph_cx", "sph_r", "tri_v0y", "tri_v1x", "tri_v2z", "cb_cx", "pl_py"):
        if len(a[key]) and rng.random() < 0.5:
            v = bad_values[int(rng.integers(0, len(bad_values)))]; i = int(rng.integers(0, len(a[key])))
            a[key][i] = v; poison.append((key, i, float(v)))
    t_min = 0.001 if rng.random() < 0.8 else float(rng.choice([-1.0, 0.0, 0.05, 2.0]))
    w, h = [(96, 54), (61, 43), (130, 20), (33, 70)][int(rng.integers(0, 4))]
    stride = 1 if rng.random() < 0.7 else 3
    p = make_params(w, h, spp=int(rng.integers(1, 4)), seed=k, t_min=t_min,
                    max_bounces=10 if rng.random() < 0.6 else int(rng.choice([1, 3, 16])),
                    gamma=bool(rng.random() < 0.7), band_begin=int(rng.integers(0, stride)), band_stride=stride)
    o_acc, o_ids, o_seg = orc.render(flat, cam, tables, p, threads=16)
    out = {}
    for mode in ("tree", "list"):
        if mode == "list": os.environ["RT_HIP_FLAT_SCAN"] = "1"
        r.set_scene(flat); os.environ.pop("RT_HIP_FLAT_SCAN", None)
        r.set_camera(cam); r.begin_frame(w, h); r.render(p)
        out[mode] = (r.read_primary_ids(), r.ray_count())
    kind = {}
    for key, nm in (("sph_id","S"),("tri_id","T"),("cb_id","C"),("pl_id","P")):
        for n, i in enumerate(a[key]): kind[int(i)] = nm + str(n)
    print("seed", seed, "poison", poison, "t_min", t_min, "counts", {k2: len(a[k2]) for k2 in ("sph_id","tri_id","cb_id","pl_id")}, "cam", [round(x, 3) for x in cam.struct.origin])
    for nm in ("tree", "list"):
        ids = out[nm][0]; d = ids != o_ids
        pairs = {}
        for g, o in zip(ids[d][:2000], o_ids[d][:2000]):
            pairs[(int(g), int(o))] = pairs.get((int(g), int(o)), 0) + 1
        print("  ", nm, "ids differ", int(d.sum()), "segments", out[nm][1], o_seg, "(gpu,oracle):", [((kind.get(g, g), kind.get(o, o)), c) for (g, o), c in sorted(pairs.items(), key=lambda x: -x[1])[:4]])
    ids = out["tree"][0]; d = np.argwhere(ids != o_ids)
    if len(d):
        y, x = d[0]
        segs, col = orc.path(flat, cam, tables, p, int(x), int(h - 1 - y), 0)
        print("   pixel", y, x, "gpu", kind.get(int(ids[y, x])), "oracle", kind.get(int(o_ids[y, x])))
        for e in segs[:2]: print("   o", e[0:3], "d", e[3:6], "hit", int(e[6]), int(e[7]), kind.get(int(e[7])), "t", e[8])
        for nm in ("C0", "C1", "S146", "S1827", "P0"):
            for i2, n2 in kind.items():
                if n2 == nm: print("   list position of", nm, i2)
        # which single primitives does the first ray hit, one at a time (oracle)
        o3, d3 = segs[0][0:3], segs[0][3:6]
