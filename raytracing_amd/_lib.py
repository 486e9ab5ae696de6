"""ctypes binding of librt_hip.so, the C ABI declared in include/rt_hip.h.

The library is the only compute path.  If it has not been built, importing
this module raises; nothing falls back to the CPU.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# RT_HIP_LIBRARY overrides the path (e.g. to A/B two builds of the kernels)
LIB_PATH = os.environ.get("RT_HIP_LIBRARY") or os.path.join(_HERE, "librt_hip.so")

RT_OK, RT_E_BADARG, RT_E_HIP, RT_E_OOM, RT_E_STATE, RT_E_IO = 0, -1, -2, -3, -4, -5
RT_METAL, RT_LAMBERTIAN, RT_DIELECTRIC = 0, 1, 2
RT_SAMPLING_DET = 0
RT_BAND_ROWS = 8
RT_MAX_BOUNCES = 16

_fp = C.POINTER(C.c_float)
_up = C.POINTER(C.c_uint32)
_ip = C.POINTER(C.c_int32)


class rt_scene(C.Structure):
    _fields_ = (
        [("n_spheres", C.c_uint32)]
        + [(n, _fp) for n in ("sph_cx", "sph_cy", "sph_cz", "sph_r")]
        + [("sph_mat", _up), ("sph_id", _up)]
        + [("n_planes", C.c_uint32)]
        + [(n, _fp) for n in ("pl_nx", "pl_ny", "pl_nz", "pl_px", "pl_py", "pl_pz")]
        + [("pl_mat", _up), ("pl_id", _up)]
        + [("n_cubes", C.c_uint32)]
        + [(n, _fp) for n in ("cb_cx", "cb_cy", "cb_cz", "cb_w", "cb_h", "cb_d")]
        + [("cb_mat", _up), ("cb_id", _up)]
        + [("n_tris", C.c_uint32)]
        + [(n, _fp) for n in ("tri_v0x", "tri_v0y", "tri_v0z", "tri_v1x", "tri_v1y", "tri_v1z",
                              "tri_v2x", "tri_v2y", "tri_v2z")]
        + [("tri_mat", _up), ("tri_id", _up)]
        + [("n_materials", C.c_uint32), ("mat_kind", _ip)]
        + [(n, _fp) for n in ("mat_r", "mat_g", "mat_b", "mat_param")]
    )


class rt_camera(C.Structure):
    _fields_ = [
        ("origin", C.c_float * 3),
        ("lower_left_corner", C.c_float * 3),
        ("horizontal", C.c_float * 3),
        ("vertical", C.c_float * 3),
        ("u", C.c_float * 3),
        ("v", C.c_float * 3),
        ("lens_radius", C.c_float),
    ]

    def as_tuple(self):
        return (tuple(self.origin), tuple(self.lower_left_corner), tuple(self.horizontal),
                tuple(self.vertical), tuple(self.u), tuple(self.v), self.lens_radius)


class rt_params(C.Structure):
    _fields_ = [
        ("width", C.c_uint32), ("height", C.c_uint32),
        ("spp", C.c_uint32), ("sample_begin", C.c_uint32), ("sample_count", C.c_uint32),
        ("max_bounces", C.c_uint32), ("gamma", C.c_uint32), ("t_min", C.c_float),
        ("band_begin", C.c_uint32), ("band_stride", C.c_uint32),
        ("sampling_mode", C.c_uint32), ("seed", C.c_uint32),
    ]


# every symbol include/rt_hip.h declares: name -> (restype, argtypes)
_vp = C.c_void_p
_vpp = C.POINTER(C.c_void_p)
_f3 = C.c_float * 3
SIGNATURES = {
    "rt_create": (C.c_int, [C.c_int, _vpp]),
    "rt_destroy": (None, [_vp]),
    "rt_last_error": (C.c_char_p, [_vp]),
    "rt_upload_tables": (C.c_int, [_vp, _fp, _fp, _fp]),
    "rt_upload_scene": (C.c_int, [_vp, C.POINTER(rt_scene)]),
    "rt_set_camera": (C.c_int, [_vp, C.POINTER(rt_camera)]),
    "rt_begin_frame": (C.c_int, [_vp, C.c_uint32, C.c_uint32]),
    "rt_begin_frames": (C.c_int, [_vp, C.c_uint32, C.c_uint32, C.c_uint32]),
    "rt_select_frame": (C.c_int, [_vp, C.c_uint32]),
    "rt_render_frames": (C.c_int, [_vp, C.POINTER(rt_params), C.POINTER(rt_camera), C.c_uint32]),
    "rt_render": (C.c_int, [_vp, C.POINTER(rt_params)]),
    "rt_render_static": (C.c_int, [_vp, C.POINTER(rt_params), _vp, _vp, _vp]),
    "rt_pack": (C.c_int, [_vp]),
    "rt_sync": (C.c_int, [_vp]),
    "rt_read_accum": (C.c_int, [_vp, _fp, C.c_size_t]),
    "rt_read_sample_counts": (C.c_int, [_vp, _up, C.c_size_t]),
    "rt_read_packed": (C.c_int, [_vp, _up, C.c_size_t]),
    "rt_read_primary_ids": (C.c_int, [_vp, _ip, C.c_size_t]),
    "rt_ray_count": (C.c_int, [_vp, C.POINTER(C.c_uint64)]),
    "rt_device_buffers": (C.c_int, [_vp, _vpp, _vpp, _vpp, _vpp]),
    "rt_export_packed_bands": (C.c_int, [_vp, _vp, C.c_size_t, C.c_uint32, C.c_uint32]),
    "rt_gather_packed": (C.c_int, [_vpp, C.c_uint32, _up, C.c_size_t]),
    "rt_check_arith": (C.c_int, [_vp, C.c_uint64, C.c_uint32, C.POINTER(C.c_uint64)]),
    "rt_kernel_info": (C.c_int, [_vp, C.c_uint32, _up, _up]),
    "rt_build_id": (C.c_char_p, []),
    "rt_timer_start": (C.c_int, [_vp]),
    "rt_timer_stop": (C.c_int, [_vp, _fp]),
    "rt_camera_new": (C.c_int, [C.c_float, C.c_float, C.c_float, C.c_float, _f3, _f3, _f3,
                                C.POINTER(rt_camera)]),
    "rt_scene_camera": (C.c_int, [C.c_uint32, C.c_float, C.POINTER(rt_camera)]),
    "rt_image_height": (C.c_uint32, [C.c_uint32, C.c_float]),
    "rt_parse_floats": (C.c_size_t, [C.c_char_p, C.c_size_t, _fp, C.c_size_t]),
    "rt_write_ppm": (C.c_int, [C.c_char_p, _up, C.c_uint32, C.c_uint32]),
    "rt_write_png": (C.c_int, [C.c_char_p, _up, C.c_uint32, C.c_uint32]),
}

_lib = None


def load():
    """Load librt_hip.so once.  Raises OSError with a build hint if it is absent."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError(
            LIB_PATH + " is missing: build it with `python -c 'import __graft_entry__ as g; "
            "g.build()'` or `make -C raytracing_amd/csrc`. There is no CPU fallback.")
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)   # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


class RtError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("librt_hip error %d: %s" % (code, message))
        self.code = code
