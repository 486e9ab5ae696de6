"""Wavefront .obj reader: `v x y z` and `f a b c ...` records only.

The reference's README advertises .obj meshes but its sources hold no loader,
so the format rules are stated here: indices are 1-based; a negative index
counts back from the last vertex read so far; `a/b/c` forms use the part before
the first slash; polygons with more than three vertices are split as a fan
around their first vertex; every other record is skipped.  The C++ host mirror
(csrc/host/scene.hpp, load_obj) follows the same rules.
"""
from typing import List, Tuple

from .scene import HittableList, Triangle


class ObjError(ValueError):
    pass


def parse_obj(text: str) -> Tuple[List[Tuple[float, float, float]], List[Tuple[int, int, int]]]:
    """-> (vertices, triangles as 0-based vertex index triples, file order)"""
    verts: List[Tuple[float, float, float]] = []
    tris: List[Tuple[int, int, int]] = []
    for lineno, line in enumerate(text.splitlines(), 1):
        parts = line.split()
        if not parts:
            continue
        if parts[0] == "v":
            try:
                verts.append((float(parts[1]), float(parts[2]), float(parts[3])))
            except (IndexError, ValueError):
                raise ObjError("line %d: bad vertex" % lineno)
        elif parts[0] == "f":
            idx = []
            for tok in parts[1:]:
                try:
                    k = int(tok.split("/")[0])
                except ValueError:
                    k = 0
                at = k - 1 if k > 0 else len(verts) + k
                if k == 0 or at < 0 or at >= len(verts):
                    raise ObjError("line %d: vertex index out of range" % lineno)
                idx.append(at)
            if len(idx) < 3:
                raise ObjError("line %d: face with < 3 vertices" % lineno)
            for k in range(1, len(idx) - 1):
                tris.append((idx[0], idx[k], idx[k + 1]))
    return verts, tris


def load_obj(path, scene: HittableList, material, scale=1.0, translate=(0.0, 0.0, 0.0)) -> int:
    """Append the triangles of an .obj file to `scene` in file order; returns how many."""
    with open(path, "r") as f:
        verts, tris = parse_obj(f.read())
    import numpy as np
    f32 = np.float32
    # x*scale + translate in binary32, like the C++ loader
    moved = [tuple(float(f32(f32(f32(c) * f32(scale)) + f32(t))) for c, t in zip(v, translate)) for v in verts]
    for a, b, c in tris:
        scene.add(Triangle(moved[a], moved[b], moved[c], material))
    return len(tris)


def write_obj(path, verts, tris):
    """Small writer used by tests and examples (1-based indices)."""
    with open(path, "w") as f:
        for v in verts:
            f.write("v %.9g %.9g %.9g\n" % tuple(v))
        for t in tris:
            f.write("f %d %d %d\n" % (t[0] + 1, t[1] + 1, t[2] + 1))
