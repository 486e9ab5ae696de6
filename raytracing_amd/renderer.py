"""Renderer: a thin object around one rt_ctx of librt_hip.so (one per GPU).

It takes the place of the reference's worker threads plus fan-in loop
(src/raytracer.rs:146-199): set tables/scene/camera, render sample ranges into
the per-pixel sums, read back sums, 0x00RRGGBB pixels and primary-hit ids.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .scene import Camera, FlatScene, HittableList

_DATA = os.path.join(os.path.dirname(os.path.abspath(__file__)), "data")
TABLE_FILES = {
    "disc": ("disc_blue_noise_64.txt", 128),
    "sphere": ("spherical_blue_noise_256.txt", 768),
    "screen": ("screenspace_blue_noise_128x128.txt", 16384),
}


def load_tables():
    """The three live tables of src/denoising/mod.rs:55,59,61 as float32 arrays,
    rounded from their decimal text by the library (strtof)."""
    lib = _lib.load()
    out = {}
    for key, (fname, count) in TABLE_FILES.items():
        with open(os.path.join(_DATA, fname), "rb") as f:
            text = f.read()
        arr = np.zeros(count, dtype=np.float32)
        n = lib.rt_parse_floats(text, len(text), arr.ctypes.data_as(C.POINTER(C.c_float)), count)
        if n != count:
            raise ValueError("%s: expected %d values, found %d" % (fname, count, n))
        out[key] = arr
    return out


def make_params(width, height, spp=1, sample_begin=0, sample_count=None, max_bounces=10, gamma=True,
                t_min=0.001, band_begin=0, band_stride=1, seed=0):
    p = _lib.rt_params()
    p.width, p.height = width, height
    p.spp = spp
    p.sample_begin = sample_begin
    p.sample_count = spp - sample_begin if sample_count is None else sample_count
    p.max_bounces = max_bounces
    p.gamma = 1 if gamma else 0
    p.t_min = t_min
    p.band_begin, p.band_stride = band_begin, band_stride
    p.sampling_mode = _lib.RT_SAMPLING_DET
    p.seed = seed
    return p


class Renderer:
    def __init__(self, device=0, tables=None):
        self._lib = _lib.load()
        self._ctx = C.c_void_p()
        rc = self._lib.rt_create(device, C.byref(self._ctx))
        if rc != 0:
            msg = self._lib.rt_last_error(None)
            raise _lib.RtError(rc, (msg or b"").decode())
        self.width = self.height = 0
        self._scene_keepalive = None
        self.set_tables(tables if tables is not None else load_tables())

    # -- plumbing
    def _check(self, rc):
        if rc != 0:
            msg = self._lib.rt_last_error(self._ctx)
            raise _lib.RtError(rc, (msg or b"").decode())

    def close(self):
        if self._ctx:
            self._lib.rt_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- inputs
    def set_tables(self, tables):
        fp = C.POINTER(C.c_float)
        d = np.ascontiguousarray(tables["disc"], dtype=np.float32)
        s = np.ascontiguousarray(tables["sphere"], dtype=np.float32)
        n = np.ascontiguousarray(tables["screen"], dtype=np.float32)
        if d.size != 128 or s.size != 768 or n.size != 16384:
            raise ValueError("table sizes must be 128 / 768 / 16384 floats")
        self._check(self._lib.rt_upload_tables(self._ctx, d.ctypes.data_as(fp), s.ctypes.data_as(fp),
                                               n.ctypes.data_as(fp)))

    def set_scene(self, scene):
        flat = scene.flatten() if isinstance(scene, HittableList) else scene
        if not isinstance(flat, FlatScene):
            raise TypeError("scene must be a HittableList or FlatScene")
        self._check(self._lib.rt_upload_scene(self._ctx, C.byref(flat.struct)))
        return flat

    def set_camera(self, cam):
        struct = cam.struct if isinstance(cam, Camera) else cam
        self._check(self._lib.rt_set_camera(self._ctx, C.byref(struct)))

    # -- frame
    def begin_frame(self, width, height):
        self._check(self._lib.rt_begin_frame(self._ctx, width, height))
        self.width, self.height = width, height

    def begin_frames(self, width, height, n_frames):
        """rt_begin_frames: per-pixel buffers for n_frames frames of one size, all zeroed"""
        self._check(self._lib.rt_begin_frames(self._ctx, width, height, n_frames))
        self.width, self.height = width, height

    def select_frame(self, frame):
        """rt_select_frame: the frame render() and the read_* calls work on"""
        self._check(self._lib.rt_select_frame(self._ctx, frame))

    def render(self, params):
        self._check(self._lib.rt_render(self._ctx, C.byref(params)))

    def render_frames(self, params, cameras):
        """rt_render_frames: frames 0..len(cameras)-1 at once, frame f through cameras[f]"""
        structs = [c.struct if isinstance(c, Camera) else c for c in cameras]
        arr = (_lib.rt_camera * len(structs))(*structs)
        self._check(self._lib.rt_render_frames(self._ctx, C.byref(params), arr, len(structs)))

    def render_static(self, params, accum_ptr, ids_ptr, segments_ptr=None):
        """rt_render_static: render into caller-owned device buffers (raw device addresses),
        working memory from the library's fixed per-GPU scratch; allocates nothing"""
        self._check(self._lib.rt_render_static(self._ctx, C.byref(params), C.c_void_p(accum_ptr),
                                               C.c_void_p(ids_ptr), C.c_void_p(segments_ptr or 0)))

    def pack(self):
        self._check(self._lib.rt_pack(self._ctx))

    def sync(self):
        self._check(self._lib.rt_sync(self._ctx))

    def check_arith(self, operands, seed=0):
        """rt_check_arith: (quotients, roots) of the short sequences that differ from `/` and sqrt."""
        bad = (C.c_uint64 * 2)()
        self._check(self._lib.rt_check_arith(self._ctx, operands, seed, bad))
        return int(bad[0]), int(bad[1])

    def kernel_variant(self, max_bounces=10):
        """rt_kernel_info: the render kernel a launch with this depth budget runs for the uploaded scene"""
        modes, depth = C.c_uint32(), C.c_uint32()
        self._check(self._lib.rt_kernel_info(self._ctx, max_bounces, C.byref(modes), C.byref(depth)))
        return "render_tiles<%d, %d>" % (modes.value, depth.value)

    def timer_start(self):
        self._check(self._lib.rt_timer_start(self._ctx))

    def timer_stop(self):
        ms = C.c_float()
        self._check(self._lib.rt_timer_stop(self._ctx, C.byref(ms)))
        return ms.value

    # -- outputs (display order, shape (H, W[, 3]))
    def read_accum(self):
        n = self.width * self.height
        out = np.empty(n * 3, dtype=np.float32)
        self._check(self._lib.rt_read_accum(self._ctx, out.ctypes.data_as(C.POINTER(C.c_float)), n))
        return out.reshape(self.height, self.width, 3)

    def read_sample_counts(self):
        n = self.width * self.height
        out = np.empty(n, dtype=np.uint32)
        self._check(self._lib.rt_read_sample_counts(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out.reshape(self.height, self.width)

    def read_packed(self):
        n = self.width * self.height
        out = np.empty(n, dtype=np.uint32)
        self._check(self._lib.rt_read_packed(self._ctx, out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
        return out.reshape(self.height, self.width)

    def read_primary_ids(self):
        n = self.width * self.height
        out = np.empty(n, dtype=np.int32)
        self._check(self._lib.rt_read_primary_ids(self._ctx, out.ctypes.data_as(C.POINTER(C.c_int32)), n))
        return out.reshape(self.height, self.width)

    def ray_count(self):
        v = C.c_uint64()
        self._check(self._lib.rt_ray_count(self._ctx, C.byref(v)))
        return v.value

    def export_packed_bands(self, dst_device_ptr, capacity_px, band_begin, band_stride):
        """quantise this GPU's bands into caller-owned device memory (compact block)"""
        self._check(self._lib.rt_export_packed_bands(self._ctx, C.c_void_p(dst_device_ptr), capacity_px,
                                                     band_begin, band_stride))

    def device_buffers(self):
        """raw device pointers (accum float4, packed u32, ids i32) and the hipStream_t"""
        ptrs = [C.c_void_p() for _ in range(4)]
        self._check(self._lib.rt_device_buffers(self._ctx, *[C.byref(p) for p in ptrs]))
        return tuple(p.value for p in ptrs)


def gather_packed(renderers):
    """rt_gather_packed: the frame whose bands b % len(renderers) == g renderer g has rendered,
    put together on the first renderer's GPU; (H, W) uint32 in display order"""
    first = renderers[0]
    n = first.width * first.height
    ctxs = (C.c_void_p * len(renderers))(*[r._ctx for r in renderers])
    out = np.empty(n, dtype=np.uint32)
    first._check(first._lib.rt_gather_packed(ctxs, len(renderers), out.ctypes.data_as(C.POINTER(C.c_uint32)), n))
    return out.reshape(first.height, first.width)


def write_ppm(path, packed):
    """PPM::write_file, src/image/mod.rs:61-67"""
    lib = _lib.load()
    arr = np.ascontiguousarray(packed, dtype=np.uint32)
    h, w = arr.shape
    rc = lib.rt_write_ppm(os.fsencode(path), arr.ctypes.data_as(C.POINTER(C.c_uint32)), w, h)
    if rc != 0:
        raise _lib.RtError(rc, "rt_write_ppm(%s)" % path)


def write_png(path, packed):
    """8-bit RGB PNG of the same pixels (the format of the reference's output_images/)"""
    lib = _lib.load()
    arr = np.ascontiguousarray(packed, dtype=np.uint32)
    h, w = arr.shape
    rc = lib.rt_write_png(os.fsencode(path), arr.ctypes.data_as(C.POINTER(C.c_uint32)), w, h)
    if rc != 0:
        raise _lib.RtError(rc, "rt_write_png(%s)" % path)
