#!/usr/bin/env python3
"""Writes tests/golden/renders.npz: outputs of the CPU oracle (oracle/rt_oracle.c, the C
restatement of the reference's render loop) for a few small frames.  They are not outputs
of the reference itself, which cannot be built here; they pin the oracle and the GPU
library to one committed result, so a change of behaviour on either side shows up.

    python tests/golden/make_renders.py          (from the repo root; needs the built oracle
                                                  and librt_hip.so for the host helpers)
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from raytracing_amd import Scene, load_tables, make_params, workloads  # noqa: E402
from tests import oracle_lib  # noqa: E402

CASES = {
    # name: (scene, camera, width, height, spp)
    "default": lambda: (Scene.get_scene().flatten(), Scene.get_camera(), 160, 90, 2),
    "default_dof": lambda: (Scene.get_scene().flatten(), Scene.get_camera(aperture=0.1), 96, 54, 4),
    "mesh": lambda: (workloads.mesh_scene(), Scene.get_camera(), 96, 54, 2),
    "spheres10k": lambda: (workloads.random_spheres(), workloads.random_spheres_camera(), 64, 36, 2),
}


def render_case(oracle, tables, name):
    flat, cam, w, h, spp = CASES[name]()
    acc, ids, seg = oracle.render(flat, cam, tables, make_params(w, h, spp=spp), threads=8)
    return {"accum": acc[..., :3].copy(), "ids": ids, "packed": oracle.pack(acc), "segments": np.uint64(seg)}


def main():
    tables = load_tables()
    oracle = oracle_lib.load()
    out = {}
    for name in CASES:
        for key, val in render_case(oracle, tables, name).items():
            out[name + "/" + key] = val
    path = os.path.join(ROOT, "tests", "golden", "renders.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
