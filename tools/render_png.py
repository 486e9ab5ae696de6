#!/usr/bin/env python3
"""Renders the reference's own scene (Scene::get_scene, src/scene/mod.rs:15-76) on the GPU at the size
of the pictures the reference publishes (1080x607, output_images/) and writes compressed PNGs:
    profiles/renders/c1_default_scene_1080x607_64spp.png   aperture 0 (the source's setting)
    profiles/renders/c2_dof_1080x607_64spp.png             aperture 0.1 (src/scene/mod.rs:70, commented out)
For looking at only: output_images/image.png of the reference shows another revision of the scene
(floating cube, dark blue ground), so no pixel comparison with it is meaningful.
usage: render_png.py [spp]"""
import os
import struct
import sys
import zlib

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from raytracing_amd import Renderer, Scene, image_height, load_tables, make_params  # noqa: E402


def write_png(path, packed):
    h, w = packed.shape
    rgb = np.stack([(packed >> 16) & 255, (packed >> 8) & 255, packed & 255], axis=-1).astype(np.uint8)
    raw = b"".join(b"\x00" + rgb[j].tobytes() for j in range(h))

    def chunk(tag, data):
        return struct.pack(">I", len(data)) + tag + data + struct.pack(">I", zlib.crc32(tag + data) & 0xffffffff)
    with open(path, "wb") as f:
        f.write(b"\x89PNG\r\n\x1a\n" + chunk(b"IHDR", struct.pack(">IIBBBBB", w, h, 8, 2, 0, 0, 0))
                + chunk(b"IDAT", zlib.compress(raw, 9)) + chunk(b"IEND", b""))


def main():
    spp = int(sys.argv[1]) if len(sys.argv) > 1 else 64
    w = 1080
    h = image_height(w)
    out = os.path.join(ROOT, "gpurun_out", "renders")
    os.makedirs(out, exist_ok=True)
    tables = load_tables()
    flat = Scene.get_scene().flatten()
    for name, aperture in (("c1_default_scene", 0.0), ("c2_dof", 0.1)):
        with Renderer(0, tables) as r:
            r.set_scene(flat)
            r.set_camera(Scene.get_camera(aperture=aperture))
            r.begin_frame(w, h)
            r.render(make_params(w, h, spp=spp, sample_begin=0, sample_count=spp))
            write_png(os.path.join(out, "%s_%dx%d_%dspp.png" % (name, w, h, spp)), r.read_packed().reshape(h, w))
    print("wrote", out)


if __name__ == "__main__":
    main()
