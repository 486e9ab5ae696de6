"""rt_render_static, the entry point that allocates nothing: its working memory is a fixed
array in the library's code object, one per GPU, shared by every call on that GPU.  Called
through the C ABI on caller-owned device buffers and compared bit for bit with
rt_begin_frame + rt_render on the same inputs; two calls in a row, the second on a smaller
frame and scene, so that anything the first call left in the scratch (staged colours, queue
counters) would show; then the first input again, and a multi-launch sample range."""
import ctypes as C

import numpy as np
import pytest

from raytracing_amd import Scene, make_params, workloads

pytestmark = pytest.mark.gpu


def _reference(renderer, flat, cam, params_list):
    renderer.set_scene(flat)
    renderer.set_camera(cam)
    renderer.begin_frame(params_list[0].width, params_list[0].height)
    for p in params_list:
        renderer.render(p)
    return renderer.read_accum(), renderer.read_sample_counts(), renderer.read_primary_ids(), renderer.ray_count()


class DeviceBuffer:
    """caller-owned device memory, from the HIP runtime the library itself is linked against"""
    hip = None

    def __init__(self, nbytes, fill_byte):
        if DeviceBuffer.hip is None:
            DeviceBuffer.hip = C.CDLL("libamdhip64.so")
            for name in ("hipMalloc", "hipMemset", "hipMemcpy", "hipFree", "hipDeviceSynchronize"):
                getattr(DeviceBuffer.hip, name).restype = C.c_int
        self.nbytes = nbytes
        self.ptr = C.c_void_p()
        assert self.hip.hipMalloc(C.byref(self.ptr), C.c_size_t(nbytes)) == 0
        assert self.hip.hipMemset(self.ptr, C.c_int(fill_byte), C.c_size_t(nbytes)) == 0
        assert self.hip.hipDeviceSynchronize() == 0

    def to_host(self, dtype):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        assert self.hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), self.ptr, C.c_size_t(self.nbytes), C.c_int(2)) == 0
        return out

    def free(self):
        self.hip.hipFree(self.ptr)


def _static(renderer, flat, cam, params_list):
    w, h = params_list[0].width, params_list[0].height
    renderer.set_scene(flat)
    renderer.set_camera(cam)
    accum = DeviceBuffer(w * h * 16, 0)
    ids = DeviceBuffer(w * h * 4, 0xff)           # -1 = nothing recorded
    seg = DeviceBuffer(8, 0)
    for p in params_list:
        renderer.render_static(p, accum.ptr.value, ids.ptr.value, seg.ptr.value)
    renderer.sync()
    a = accum.to_host(np.float32).reshape(h, w, 4)
    out = (a[..., :3].copy(), a[..., 3].astype(np.uint32), ids.to_host(np.int32).reshape(h, w),
           int(seg.to_host(np.uint64)[0]))
    for b in (accum, ids, seg):
        b.free()
    return out


def _same(got, want):
    assert np.array_equal(got[0].view(np.uint32), want[0].view(np.uint32)), "sums differ"
    assert np.array_equal(got[1], want[1]), "sample counts differ"
    assert np.array_equal(got[2], want[2]), "primary ids differ"
    assert got[3] == want[3], "segment counts differ"


def test_static_scratch_entry_point_equals_rt_render(renderer):
    big = workloads.workload("c4")
    flat1, cam1 = big["flat"], big["camera"]
    p1 = [make_params(640, 360, spp=12)]
    flat2, cam2 = Scene.get_scene().flatten(), Scene.get_camera(aperture=0.1)
    p2 = [make_params(61, 43, spp=3)]
    want1 = _reference(renderer, flat1, cam1, p1)
    want2 = _reference(renderer, flat2, cam2, p2)
    _same(_static(renderer, flat1, cam1, p1), want1)
    _same(_static(renderer, flat2, cam2, p2), want2)      # smaller input right after a larger one
    _same(_static(renderer, flat1, cam1, p1), want1)
    # 150 samples = three launches (64 + 64 + 22) sharing the ring, in two calls; one band share
    p3 = [make_params(40, 24, spp=150, sample_begin=0, sample_count=70),
          make_params(40, 24, spp=150, sample_begin=70, sample_count=80)]
    _same(_static(renderer, flat2, cam2, p3), _reference(renderer, flat2, cam2, p3))
    p4 = [make_params(120, 67, spp=2, band_begin=1, band_stride=3)]
    _same(_static(renderer, flat2, cam2, p4), _reference(renderer, flat2, cam2, p4))


def test_static_entry_point_rejects_bad_arguments(renderer):
    from raytracing_amd import _lib
    renderer.set_scene(Scene.get_scene().flatten())
    renderer.set_camera(Scene.get_camera())
    buf = DeviceBuffer(1024, 0)
    with pytest.raises(_lib.RtError) as e:
        renderer.render_static(make_params(65537, 4, spp=1), buf.ptr.value, buf.ptr.value)
    assert e.value.code == _lib.RT_E_BADARG and "RT_STATIC_MAX_PIXELS" in str(e.value)
    with pytest.raises(_lib.RtError) as e:
        renderer.render_static(make_params(8, 8, spp=1), 0, buf.ptr.value)
    assert e.value.code == _lib.RT_E_BADARG
    buf.free()
