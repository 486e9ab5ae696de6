"""Synthetic inputs of the benchmark configurations.  All randomness comes from
a PCG32 stream with a fixed seed so every machine builds the same scene.

c1  default scene of the reference, 600x337, 1 spp, aperture 0
c2  the same scene with aperture 0.1, 1920x1080, 64 spp
c3  default scene + a triangle mesh read from data/uv_sphere.obj (a UV sphere), 1920x1080, 256 spp
c4  10,000 random spheres on a ground plane, 3840x2160, 256 spp
c5  1024x512 displaced triangle grid (~1.05 M triangles) + default scene, 3840x2160, 1024 spp
"""
import math
import os

import numpy as np

from .scene import (Camera, Scene, Sphere, Triangle, Lambertian, flat_scene_from_arrays, image_height)
from . import _lib
from .obj import load_obj, write_obj


class Pcg32:
    """PCG-XSH-RR 64/32 (O'Neill), the generator the scene builders draw from"""
    MULT = 6364136223846793005
    MASK = (1 << 64) - 1

    def __init__(self, seed, seq=54):
        self.inc = ((seq << 1) | 1) & self.MASK
        self.state = 0
        self.next_u32()
        self.state = (self.state + seed) & self.MASK
        self.next_u32()

    def next_u32(self):
        old = self.state
        self.state = (old * self.MULT + self.inc) & self.MASK
        xs = (((old >> 18) ^ old) >> 27) & 0xFFFFFFFF
        rot = old >> 59
        return ((xs >> rot) | (xs << ((-rot) & 31))) & 0xFFFFFFFF

    def uniform(self, lo=0.0, hi=1.0):
        return lo + (hi - lo) * (self.next_u32() >> 8) * (1.0 / 16777216.0)


def default_camera(aperture=0.0, spp=1):
    return Scene.get_camera(aperture=aperture, samples_per_px=spp)


def random_spheres(n=10000, seed=1234, extent=30.0):
    """n non-overlapping spheres resting on the plane y=0: centres uniform in
    x,z in [-extent, extent], radius uniform [0.1, 0.4]; 60 % Lambertian, 30 %
    Metal (fuzz U[0,0.5]), 10 % Dielectric 1.5; albedo U[0.1,0.9]^3.  List order:
    the spheres, then the ground plane (Lambertian 0.5 grey)."""
    rng = Pcg32(seed)
    cx = np.zeros(n); cz = np.zeros(n); rr = np.zeros(n)
    kind = np.zeros(n, dtype=np.int32); alb = np.zeros((n, 3)); par = np.zeros(n)
    k = 0
    while k < n:
        x = rng.uniform(-extent, extent); z = rng.uniform(-extent, extent); r = rng.uniform(0.1, 0.4)
        if k:
            d2 = (cx[:k] - x) ** 2 + (cz[:k] - z) ** 2 + (rr[:k] - r) ** 2
            if np.any(d2 < (rr[:k] + r) ** 2):
                continue
        cx[k], cz[k], rr[k] = x, z, r
        m = rng.uniform()
        alb[k] = (rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9), rng.uniform(0.1, 0.9))
        if m < 0.6:
            kind[k] = _lib.RT_LAMBERTIAN
        elif m < 0.9:
            kind[k] = _lib.RT_METAL; par[k] = rng.uniform(0.0, 0.5)
        else:
            kind[k] = _lib.RT_DIELECTRIC; par[k] = 1.5
        k += 1
    ids = np.arange(n, dtype=np.uint32)
    return flat_scene_from_arrays(
        sph_cx=cx, sph_cy=rr, sph_cz=cz, sph_r=rr, sph_mat=ids, sph_id=ids,
        pl_nx=[0.0], pl_ny=[1.0], pl_nz=[0.0], pl_px=[0.0], pl_py=[0.0], pl_pz=[0.0],
        pl_mat=[n], pl_id=[n],
        mat_kind=np.concatenate([kind, [_lib.RT_LAMBERTIAN]]),
        mat_r=np.concatenate([alb[:, 0], [0.5]]), mat_g=np.concatenate([alb[:, 1], [0.5]]),
        mat_b=np.concatenate([alb[:, 2], [0.5]]), mat_param=np.concatenate([par, [0.0]]))


def random_spheres_camera(spp=256):
    origin, look_at = (0.0, 6.0, 25.0), (0.0, 0.0, 0.0)
    focus = math.sqrt(sum((a - b) ** 2 for a, b in zip(origin, look_at)))
    aspect = float(np.float32(16.0) / np.float32(9.0))
    return Camera.new(27.0, aspect, 0.0, focus, origin, look_at, (0.0, 1.0, 0.0), spp)


def uv_sphere_mesh(center=(1.3, 0.3, 0.6), radius=0.3, n_lat=16, n_lon=32):
    """closed triangle mesh (2*n_lon*(n_lat-1) triangles), outward winding"""
    tris = []
    def pt(i, j):
        th = math.pi * i / n_lat
        ph = 2.0 * math.pi * j / n_lon
        return (center[0] + radius * math.sin(th) * math.cos(ph),
                center[1] + radius * math.cos(th),
                center[2] + radius * math.sin(th) * math.sin(ph))
    for i in range(n_lat):
        for j in range(n_lon):
            a, b, c, d = pt(i, j), pt(i + 1, j), pt(i + 1, j + 1), pt(i, j + 1)
            if i != 0:
                tris.append((a, d, b))
            if i != n_lat - 1:
                tris.append((b, d, c))
    return tris


MESH_OBJ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data", "uv_sphere.obj")


def write_mesh_obj(path=MESH_OBJ):
    """uv_sphere_mesh() as a Wavefront .obj: vertices as binary32 values (nine significant digits,
    which read back exactly), shared between the triangles that meet in them; faces in list order"""
    index, verts, faces = {}, [], []
    for tri in uv_sphere_mesh():
        face = []
        for v in tri:
            key = tuple(float(np.float32(c)) for c in v)
            if key not in index:
                index[key] = len(verts)
                verts.append(key)
            face.append(index[key])
        faces.append(tuple(face))
    write_obj(path, verts, faces)
    return len(verts), len(faces)


def mesh_scene():
    """c3: the default list followed by the triangles of data/uv_sphere.obj (the 960-triangle UV sphere
    of uv_sphere_mesh(), written by write_mesh_obj), Lambertian (0.2, 0.25, 0.6): the mesh comes in
    through the .obj reader like any other, in file order"""
    scene = Scene.get_scene()
    load_obj(MESH_OBJ, scene, Lambertian((0.2, 0.25, 0.6)))
    return scene.flatten()


def displaced_grid(nx=1024, nz=512, seed=1234):
    """c5: (nx-1)*(nz-1)*2 triangles over x in [-8,8], z in [-9,-1], height = four
    sine octaves around y=-0.05 .. placed behind and below the default list's
    objects so they stay visible; list order: the 8 default primitives, then the
    grid triangles row by row."""
    flat0 = Scene.get_scene().flatten()
    xs = np.linspace(-8.0, 8.0, nx)
    zs = np.linspace(-9.0, -1.6, nz)
    X, Z = np.meshgrid(xs, zs, indexing="xy")            # (nz, nx)
    rng = Pcg32(seed)
    Y = np.zeros_like(X)
    for o in range(4):
        fx = (1.5 + rng.uniform()) * (2 ** o); fz = (1.5 + rng.uniform()) * (2 ** o)
        px_ = rng.uniform(0, 6.28); pz_ = rng.uniform(0, 6.28)
        Y += (0.12 / (2 ** o)) * np.sin(fx * X * 0.5 + px_) * np.sin(fz * Z * 0.5 + pz_)
    Y += 0.02 + 0.05 * np.abs(Z + 1.6)                   # a slope rising away from the camera
    P = np.stack([X, Y, Z], -1).astype(np.float32)       # (nz, nx, 3)
    a = P[:-1, :-1].reshape(-1, 3); b = P[:-1, 1:].reshape(-1, 3)
    c = P[1:, :-1].reshape(-1, 3); d = P[1:, 1:].reshape(-1, 3)
    v0 = np.concatenate([a, b], 0); v1 = np.concatenate([c, c], 0); v2 = np.concatenate([b, d], 0)
    # interleave so both triangles of a cell are neighbours in the list
    n_cell = a.shape[0]
    order = np.arange(2 * n_cell).reshape(2, n_cell).T.reshape(-1)
    v0, v1, v2 = v0[order], v1[order], v2[order]
    n_tri = v0.shape[0]
    arr = {k: v.copy() for k, v in flat0.arrays.items()}
    n0 = flat0.n_prims
    n_mat0 = flat0.counts["n_materials"]
    arr.update(
        tri_v0x=v0[:, 0], tri_v0y=v0[:, 1], tri_v0z=v0[:, 2],
        tri_v1x=v1[:, 0], tri_v1y=v1[:, 1], tri_v1z=v1[:, 2],
        tri_v2x=v2[:, 0], tri_v2y=v2[:, 1], tri_v2z=v2[:, 2],
        tri_mat=np.full(n_tri, n_mat0, dtype=np.uint32),
        tri_id=np.arange(n0, n0 + n_tri, dtype=np.uint32),
        mat_kind=np.concatenate([arr["mat_kind"], [_lib.RT_LAMBERTIAN]]),
        mat_r=np.concatenate([arr["mat_r"], [0.35]]), mat_g=np.concatenate([arr["mat_g"], [0.45]]),
        mat_b=np.concatenate([arr["mat_b"], [0.3]]), mat_param=np.concatenate([arr["mat_param"], [0.0]]))
    return flat_scene_from_arrays(**arr)


def workload(name):
    """-> dict(flat, camera, width, height, spp, description)"""
    name = name.lower()
    if name == "c1":
        # the reference's own run: NUM_FRAMES_TO_RENDER = 10 frames, the camera one dolly step further
        # after each (src/raytracer.rs:32,165-168,190; Scene::move_cam, src/scene/mod.rs:78-83)
        w = 600
        keep = Scene.COUNTER
        cams = []
        for f in range(10):
            Scene.COUNTER = f
            cams.append(default_camera(0.0, 1))
        Scene.COUNTER = keep
        return dict(flat=Scene.get_scene().flatten(), camera=cams[0], cameras=cams, width=w,
                    height=image_height(w), spp=1,
                    description="reference default scene (8 primitives), 600x337, 1 spp, its 10 frames")
    if name == "c2":
        return dict(flat=Scene.get_scene().flatten(), camera=default_camera(0.1, 64), width=1920,
                    height=image_height(1920), spp=64,
                    description="default scene, aperture 0.1 (thin lens), 1920x1080, 64 spp")
    if name == "c3":
        return dict(flat=mesh_scene(), camera=default_camera(0.0, 256), width=1920,
                    height=image_height(1920), spp=256,
                    description="default scene + 960-triangle UV-sphere mesh, 1920x1080, 256 spp")
    if name == "c4":
        return dict(flat=random_spheres(), camera=random_spheres_camera(256), width=3840,
                    height=image_height(3840), spp=256,
                    description="10,000 random spheres + ground plane, 3840x2160, 256 spp")
    if name == "c5":
        return dict(flat=displaced_grid(), camera=default_camera(0.0, 1024), width=3840,
                    height=image_height(3840), spp=1024,
                    description="default scene + 1,045,506-triangle displaced grid, 3840x2160, 1024 spp")
    raise ValueError("unknown workload " + name)


def flops_per_segment(flat):
    """algorithmic work of one closest-hit query over the list, mul/add counted
    singly (no fused multiply-add in the reference): 18 per sphere, 14 per
    plane, 18 per cube (12 + 6 divisions), 45 per triangle, plus 80 for
    finalise + scatter + sky"""
    c = flat.counts
    return 18 * c["n_spheres"] + 14 * c["n_planes"] + 18 * c["n_cubes"] + 45 * c["n_tris"] + 80
