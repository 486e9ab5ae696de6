"""Host-side mirror of the reference's scene API (names and argument meaning
follow src/geometry/mod.rs, src/scene/mod.rs and src/camera/mod.rs of
ojferro/raytracing), flattened into the structure-of-arrays rt_scene that
librt_hip.so takes.  No intersection or shading code lives here: building the
list is all the host does.
"""
import ctypes as C
from dataclasses import dataclass
from typing import List, Sequence, Tuple

import numpy as np

from . import _lib

Vec = Tuple[float, float, float]


# ---- materials: src/geometry/mod.rs:268-304 --------------------------------
@dataclass
class Metal:
    albedo: Vec
    fuzz: float
    kind = _lib.RT_METAL

    @property
    def param(self):
        return self.fuzz


@dataclass
class Lambertian:
    albedo: Vec
    kind = _lib.RT_LAMBERTIAN
    param = 0.0


@dataclass
class Dielectric:
    albedo: Vec
    index_of_refraction: float
    kind = _lib.RT_DIELECTRIC

    @property
    def param(self):
        return self.index_of_refraction


# ---- primitives: src/geometry/mod.rs:47-57,97-109,134-155 ------------------
@dataclass
class Sphere:
    center: Vec
    radius: float
    material: object


@dataclass
class Plane:
    normal: Vec
    point: Vec
    material: object
    single_sided: bool = True   # stored and never read, as in the reference (:102)


@dataclass
class Cube:
    center: Vec
    w: float
    h: float
    d: float
    material: object


@dataclass
class Triangle:
    """Not in the reference sources (its README mentions .obj meshes only)."""
    v0: Vec
    v1: Vec
    v2: Vec
    material: object


class FlatScene:
    """rt_scene plus the numpy arrays that back its pointers."""

    def __init__(self, arrays, counts):
        self.arrays = arrays
        self.counts = counts
        s = _lib.rt_scene()
        for name, arr in arrays.items():
            ctype = {np.dtype(np.float32): C.c_float, np.dtype(np.uint32): C.c_uint32,
                     np.dtype(np.int32): C.c_int32}[arr.dtype]
            setattr(s, name, arr.ctypes.data_as(C.POINTER(ctype)))
        for name, n in counts.items():
            setattr(s, name, n)
        self.struct = s

    @property
    def n_prims(self):
        c = self.counts
        return c["n_spheres"] + c["n_planes"] + c["n_cubes"] + c["n_tris"]

    def kinds_in_list_order(self):
        """kind name per list position, for reports"""
        out = [None] * self.n_prims
        for key, name in (("sph_id", "sphere"), ("pl_id", "plane"), ("cb_id", "cube"), ("tri_id", "triangle")):
            for i in self.arrays[key]:
                out[int(i)] = name
        return out


class HittableList:
    """Insertion-ordered list, src/geometry/mod.rs:228-238.  Materials are
    stored once per primitive, like the reference's Box<dyn Material> fields."""

    def __init__(self):
        self.list: List[object] = []

    def add(self, hittable):
        self.list.append(hittable)
        return self

    def __len__(self):
        return len(self.list)

    def flatten(self) -> FlatScene:
        f32, u32 = np.float32, np.uint32
        groups = {Sphere: [], Plane: [], Cube: [], Triangle: []}
        mats = []
        for idx, h in enumerate(self.list):
            if type(h) not in groups:
                raise TypeError("not a hittable: %r" % (h,))
            mats.append(h.material)
            groups[type(h)].append((idx, len(mats) - 1, h))

        def col(items, fn, dtype=f32):
            return np.ascontiguousarray(np.array([fn(x) for x in items], dtype=dtype))

        a = {}
        sp = groups[Sphere]
        a["sph_cx"] = col(sp, lambda x: x[2].center[0]); a["sph_cy"] = col(sp, lambda x: x[2].center[1])
        a["sph_cz"] = col(sp, lambda x: x[2].center[2]); a["sph_r"] = col(sp, lambda x: x[2].radius)
        a["sph_mat"] = col(sp, lambda x: x[1], u32); a["sph_id"] = col(sp, lambda x: x[0], u32)
        pl = groups[Plane]
        for k, ax in enumerate("xyz"):
            a["pl_n" + ax] = col(pl, lambda x, k=k: x[2].normal[k])
            a["pl_p" + ax] = col(pl, lambda x, k=k: x[2].point[k])
        a["pl_mat"] = col(pl, lambda x: x[1], u32); a["pl_id"] = col(pl, lambda x: x[0], u32)
        cb = groups[Cube]
        for k, ax in enumerate("xyz"):
            a["cb_c" + ax] = col(cb, lambda x, k=k: x[2].center[k])
        a["cb_w"] = col(cb, lambda x: x[2].w); a["cb_h"] = col(cb, lambda x: x[2].h)
        a["cb_d"] = col(cb, lambda x: x[2].d)
        a["cb_mat"] = col(cb, lambda x: x[1], u32); a["cb_id"] = col(cb, lambda x: x[0], u32)
        tr = groups[Triangle]
        for vname in ("v0", "v1", "v2"):
            for k, ax in enumerate("xyz"):
                a["tri_" + vname + ax] = col(tr, lambda x, k=k, vname=vname: getattr(x[2], vname)[k])
        a["tri_mat"] = col(tr, lambda x: x[1], u32); a["tri_id"] = col(tr, lambda x: x[0], u32)
        a["mat_kind"] = col(mats, lambda m: m.kind, np.int32)
        a["mat_r"] = col(mats, lambda m: m.albedo[0]); a["mat_g"] = col(mats, lambda m: m.albedo[1])
        a["mat_b"] = col(mats, lambda m: m.albedo[2]); a["mat_param"] = col(mats, lambda m: m.param)
        counts = {"n_spheres": len(sp), "n_planes": len(pl), "n_cubes": len(cb), "n_tris": len(tr),
                  "n_materials": len(mats)}
        return FlatScene(a, counts)


def flat_scene_from_arrays(**arrays) -> FlatScene:
    """Build a FlatScene straight from arrays (large synthetic scenes).  Missing
    groups are empty; `*_id` must together be a permutation of the list."""
    f32, u32 = np.float32, np.uint32
    names_f = (["sph_cx", "sph_cy", "sph_cz", "sph_r"]
               + ["pl_n" + c for c in "xyz"] + ["pl_p" + c for c in "xyz"]
               + ["cb_c" + c for c in "xyz"] + ["cb_w", "cb_h", "cb_d"]
               + ["tri_" + v + c for v in ("v0", "v1", "v2") for c in "xyz"]
               + ["mat_r", "mat_g", "mat_b", "mat_param"])
    names_u = ["sph_mat", "sph_id", "pl_mat", "pl_id", "cb_mat", "cb_id", "tri_mat", "tri_id"]
    a = {}
    for n in names_f:
        a[n] = np.ascontiguousarray(arrays.get(n, np.zeros(0)), dtype=f32)
    for n in names_u:
        a[n] = np.ascontiguousarray(arrays.get(n, np.zeros(0)), dtype=u32)
    a["mat_kind"] = np.ascontiguousarray(arrays.get("mat_kind", np.zeros(0)), dtype=np.int32)
    counts = {"n_spheres": len(a["sph_id"]), "n_planes": len(a["pl_id"]), "n_cubes": len(a["cb_id"]),
              "n_tris": len(a["tri_id"]), "n_materials": len(a["mat_kind"])}
    return FlatScene(a, counts)


class Camera:
    """Camera::new, src/camera/mod.rs:35-73.  The arithmetic runs in the
    library (rt_camera_new) so host languages cannot disagree on tanf."""

    def __init__(self, struct, samples_per_px=1):
        self.struct = struct
        self.samples_per_px = samples_per_px

    @staticmethod
    def new(v_fov, aspect_ratio, aperture, focus_dist, origin: Sequence[float],
            look_at: Sequence[float], v_up: Sequence[float], samples_per_px=1):
        lib = _lib.load()
        out = _lib.rt_camera()
        f3 = C.c_float * 3
        rc = lib.rt_camera_new(v_fov, aspect_ratio, aperture, focus_dist, f3(*origin), f3(*look_at),
                               f3(*v_up), C.byref(out))
        if rc != 0:
            raise _lib.RtError(rc, "rt_camera_new")
        return Camera(out, samples_per_px)


class Scene:
    """Scene::get_scene / get_camera / move_cam, src/scene/mod.rs:15-88.  The
    dolly counter is an explicit class attribute instead of a racy global."""

    COUNTER = 0

    @staticmethod
    def get_scene() -> HittableList:
        scene = HittableList()
        # yellow fuzzy metal sphere (:20-22)
        scene.add(Sphere((0.80, 0.5, -1.0), 0.5, Metal((0.8, 0.6, 0.2), 0.25)))
        # red diffuse sphere (:25-27)
        scene.add(Sphere((-0.10, 0.25, -0.10), 0.25, Lambertian((0.7, 0.3, 0.3))))
        # shiny metal sphere (:30-32)
        scene.add(Sphere((-0.80, 0.5, -1.0), 0.5, Metal((0.8, 0.8, 0.8), 0.0)))
        # solid glass sphere (:35-37)
        scene.add(Sphere((0.25, 0.75, -0.5), 0.1, Dielectric((1.0, 1.0, 1.0), 1.5)))
        # hollow glass sphere: outer shell and negative-radius inner shell (:40-43)
        scene.add(Sphere((-0.25, 0.75, -0.42), 0.14, Dielectric((0.95, 0.95, 1.0), 1.5)))
        scene.add(Sphere((-0.25, 0.75, -0.42), -0.13, Dielectric((0.95, 0.95, 1.0), 1.5)))
        # cube (:46-48)
        scene.add(Cube((0.0, 0.5, -1.0), 0.50, 0.50, 0.50, Lambertian((0.7, 0.3, 0.7))))
        # ground plane (:51-53)
        scene.add(Plane((0.0, 1.0, 0.0), (0.0, 0.0, 0.0), Lambertian((0.3, 0.3, 0.3)), True))
        return scene

    @staticmethod
    def get_camera(aperture=0.0, samples_per_px=1) -> Camera:
        lib = _lib.load()
        out = _lib.rt_camera()
        rc = lib.rt_scene_camera(Scene.COUNTER, aperture, C.byref(out))
        if rc != 0:
            raise _lib.RtError(rc, "rt_scene_camera")
        return Camera(out, samples_per_px)

    @staticmethod
    def move_cam():
        Scene.COUNTER = (Scene.COUNTER + 1) % 100


def image_height(width, aspect_ratio=None):
    """`(image_width as f32 / aspect_ratio) as usize`, src/raytracer.rs:118-120"""
    lib = _lib.load()
    if aspect_ratio is None:
        aspect_ratio = float(np.float32(16.0) / np.float32(9.0))
    return int(lib.rt_image_height(width, aspect_ratio))
