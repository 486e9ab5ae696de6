"""ctypes wrapper of oracle/librt_oracle.so, the CPU restatement used as the
checker.  Test infrastructure only."""
import ctypes as C
import os
import subprocess

import numpy as np

from raytracing_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "oracle", "librt_oracle.so")
MODE_DET, MODE_REF_SEQ = 0, 1

_fp = C.POINTER(C.c_float)


class SeqState(C.Structure):
    _fields_ = [("counter_disc", C.c_uint32), ("counter_sphere", C.c_uint32), ("pcg_state", C.c_uint64)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        f3 = C.c_float * 3
        lib.orc_render.restype = C.c_int
        lib.orc_render.argtypes = [C.POINTER(_lib.rt_scene), C.POINTER(_lib.rt_camera), _fp, _fp, _fp,
                                   C.POINTER(_lib.rt_params), C.c_int, C.c_uint32, _fp,
                                   C.POINTER(C.c_int32), C.POINTER(C.c_uint64), C.POINTER(SeqState)]
        lib.orc_pack.restype = None
        lib.orc_pack.argtypes = [_fp, C.c_size_t, C.POINTER(C.c_uint32)]
        lib.orc_camera_new.restype = C.c_int
        lib.orc_camera_new.argtypes = [C.c_float] * 4 + [f3, f3, f3, C.POINTER(_lib.rt_camera)]
        lib.orc_scene_camera.restype = C.c_int
        lib.orc_scene_camera.argtypes = [C.c_uint32, C.c_float, C.POINTER(_lib.rt_camera)]
        lib.orc_image_height.restype = C.c_uint32
        lib.orc_image_height.argtypes = [C.c_uint32, C.c_float]
        lib.orc_trace_one.restype = C.c_int
        lib.orc_trace_one.argtypes = [C.POINTER(_lib.rt_scene), _fp, _fp, f3, f3, C.c_float, C.c_float,
                                      C.c_uint32, C.c_uint32, C.c_uint32, _fp]
        lib.orc_path.restype = C.c_int
        lib.orc_path.argtypes = [C.POINTER(_lib.rt_scene), C.POINTER(_lib.rt_camera), _fp, _fp, _fp,
                                 C.POINTER(_lib.rt_params), C.c_uint32, C.c_uint32, C.c_uint32, _fp,
                                 C.c_uint32, _fp]
        lib.orc_get_ray.restype = None
        lib.orc_get_ray.argtypes = [C.POINTER(_lib.rt_camera)] + [C.c_float] * 4 + [_fp]
        lib.orc_sky.restype = None
        lib.orc_sky.argtypes = [f3, C.c_int, _fp]

    def render(self, flat, cam, tables, params, mode=MODE_DET, threads=8, accum=None, want_ids=True,
               seq=None):
        """returns (accum[H,W,4], ids[H,W] or None, segments)"""
        h, w = params.height, params.width
        if accum is None:
            accum = np.zeros((h, w, 4), dtype=np.float32)
        ids = np.full((h, w), -1, dtype=np.int32) if want_ids else None
        seg = C.c_uint64(0)
        cam_struct = cam.struct if hasattr(cam, "struct") else cam
        rc = self.lib.orc_render(
            C.byref(flat.struct), C.byref(cam_struct),
            tables["disc"].ctypes.data_as(_fp), tables["sphere"].ctypes.data_as(_fp),
            tables["screen"].ctypes.data_as(_fp), C.byref(params), mode, threads,
            accum.ctypes.data_as(_fp), ids.ctypes.data_as(C.POINTER(C.c_int32)) if want_ids else None,
            C.byref(seg), C.byref(seq) if seq is not None else None)
        if rc != 0:
            raise RuntimeError("orc_render failed: %d" % rc)
        return accum, ids, seg.value

    def pack(self, accum):
        a = np.ascontiguousarray(accum, dtype=np.float32)
        h, w = a.shape[:2]
        out = np.zeros((h, w), dtype=np.uint32)
        self.lib.orc_pack(a.ctypes.data_as(_fp), h * w, out.ctypes.data_as(C.POINTER(C.c_uint32)))
        return out

    def camera_new(self, v_fov, aspect, aperture, focus, origin, look_at, up):
        out = _lib.rt_camera()
        f3 = C.c_float * 3
        self.lib.orc_camera_new(v_fov, aspect, aperture, focus, f3(*origin), f3(*look_at), f3(*up), C.byref(out))
        return out

    def scene_camera(self, counter=0, aperture=0.0):
        out = _lib.rt_camera()
        self.lib.orc_scene_camera(counter, aperture, C.byref(out))
        return out

    def image_height(self, w, aspect):
        return int(self.lib.orc_image_height(w, aspect))

    def trace_one(self, flat, tables, origin, direction, t_min=0.001, t_max=float("inf"), pix=(0, 0),
                  sphere_tick=0):
        f3 = C.c_float * 3
        out = np.zeros(16, dtype=np.float32)
        rc = self.lib.orc_trace_one(C.byref(flat.struct), tables["sphere"].ctypes.data_as(_fp),
                                    tables["screen"].ctypes.data_as(_fp), f3(*origin), f3(*direction),
                                    t_min, t_max, pix[0], pix[1], sphere_tick, out.ctypes.data_as(_fp))
        if rc != 0:
            raise RuntimeError("orc_trace_one failed: %d" % rc)
        return {"hit": bool(out[0]), "index": int(out[1]), "t": out[2], "front_face": bool(out[3]),
                "p": out[4:7].copy(), "normal": out[7:10].copy(), "dir": out[10:13].copy(),
                "attenuation": out[13:16].copy()}

    def path(self, flat, cam, tables, params, i, j, s):
        """segments of the path of pixel (i, j from the bottom), sample s:
        (array[n, 9] of {origin, direction, hit, list index, t}, colour[3])"""
        out = np.zeros((32, 9), dtype=np.float32)
        colour = np.zeros(3, dtype=np.float32)
        cam_struct = cam.struct if hasattr(cam, "struct") else cam
        n = self.lib.orc_path(C.byref(flat.struct), C.byref(cam_struct),
                              tables["disc"].ctypes.data_as(_fp), tables["sphere"].ctypes.data_as(_fp),
                              tables["screen"].ctypes.data_as(_fp), C.byref(params), i, j, s,
                              out.ctypes.data_as(_fp), 32, colour.ctypes.data_as(_fp))
        if n < 0:
            raise RuntimeError("orc_path failed: %d" % n)
        return out[:n].copy(), colour

    def get_ray(self, cam, s, t, a, b):
        out = np.zeros(6, dtype=np.float32)
        self.lib.orc_get_ray(C.byref(cam), s, t, a, b, out.ctypes.data_as(_fp))
        return out[:3].copy(), out[3:].copy()

    def sky(self, d, gamma=True):
        out = np.zeros(3, dtype=np.float32)
        self.lib.orc_sky((C.c_float * 3)(*d), 1 if gamma else 0, out.ctypes.data_as(_fp))
        return out


_oracle = None


def load():
    global _oracle
    if _oracle is None:
        if not os.path.exists(SO):
            subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
        _oracle = Oracle(C.CDLL(SO))
    return _oracle
