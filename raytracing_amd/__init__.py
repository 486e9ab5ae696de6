"""raytracing_amd: the per-pixel render loop of ojferro/raytracing as hand-written
HIP kernels for MI355X (gfx950) behind a C ABI (include/rt_hip.h), with a host-side
mirror of the reference's scene API.  There is no CPU rendering path in this package."""
from . import _lib
from .obj import load_obj, parse_obj, write_obj
from .renderer import Renderer, gather_packed, load_tables, make_params, write_png, write_ppm
from .scene import (Camera, Cube, Dielectric, FlatScene, HittableList, Lambertian, Metal, Plane, Scene,
                    Sphere, Triangle, flat_scene_from_arrays, image_height)

__all__ = ["Renderer", "load_tables", "make_params", "write_ppm", "write_png", "Camera", "Cube", "Dielectric",
           "FlatScene", "HittableList", "Lambertian", "Metal", "Plane", "Scene", "Sphere", "Triangle",
           "flat_scene_from_arrays", "image_height", "load_obj", "parse_obj", "write_obj", "_lib"]
