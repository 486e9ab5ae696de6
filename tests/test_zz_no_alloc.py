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
    # 150 samples in two calls (units of a few pixels), 600 samples = two launches (512 + 88) sharing
    # the ring; one band share
    p3 = [make_params(40, 24, spp=150, sample_begin=0, sample_count=70),
          make_params(40, 24, spp=150, sample_begin=70, sample_count=80)]
    _same(_static(renderer, flat2, cam2, p3), _reference(renderer, flat2, cam2, p3))
    p5 = [make_params(21, 13, spp=600)]
    _same(_static(renderer, flat2, cam2, p5), _reference(renderer, flat2, cam2, p5))
    p4 = [make_params(120, 67, spp=2, band_begin=1, band_stride=3)]
    _same(_static(renderer, flat2, cam2, p4), _reference(renderer, flat2, cam2, p4))


def _free_device_memory():
    hip = DeviceBuffer.hip
    free, total = C.c_size_t(), C.c_size_t()
    assert hip.hipDeviceSynchronize() == 0
    assert hip.hipMemGetInfo(C.byref(free), C.byref(total)) == 0
    return free.value


def test_static_entry_point_allocates_nothing(tables):
    """what the entry point is for: on a context that never had a frame (no staging ring, no per-pixel
    buffers of the library's own) calls of different frame sizes, sample ranges, scenes and kernels leave
    the device's free memory where it was after a warm-up call of each kernel.  The context's stream is
    handed out without a frame, with null buffers."""
    from raytracing_amd import Renderer
    big = workloads.workload("c4")
    small, cam_small = Scene.get_scene().flatten(), Scene.get_camera(aperture=0.1)
    outputs = [DeviceBuffer(640 * 360 * 16, 0), DeviceBuffer(640 * 360 * 4, 0xff), DeviceBuffer(8, 0)]
    DeviceBuffer.hip.hipMemGetInfo.restype = C.c_int
    with Renderer(0, tables) as r:
        accum, packed, ids, stream = r.device_buffers()
        assert accum is None and packed is None and ids is None and stream
        calls = [(big["flat"], big["camera"], make_params(640, 360, spp=2)),
                 (small, cam_small, make_params(61, 43, spp=3)),
                 (small, cam_small, make_params(21, 13, spp=600)),                     # two launches
                 (big["flat"], big["camera"], make_params(320, 180, spp=4, max_bounces=16)),   # the deep kernel
                 (small, cam_small, make_params(120, 67, spp=2, band_begin=1, band_stride=3))]

        def run(flat, cam, p):
            r.set_camera(cam)
            r.render_static(p, outputs[0].ptr.value, outputs[1].ptr.value, outputs[2].ptr.value)
            r.sync()

        # scenes are uploaded (that allocates) and every kernel has run once before the measurement
        r.set_scene(big["flat"])
        run(*calls[0]); run(*calls[3])
        before_big = _free_device_memory()
        for flat, cam, p in (calls[0], calls[3], calls[0]):
            run(flat, cam, p)
        assert _free_device_memory() == before_big
        r.set_scene(small)
        run(*calls[1])
        before_small = _free_device_memory()
        for flat, cam, p in (calls[1], calls[2], calls[4], calls[1]):
            run(flat, cam, p)
        assert _free_device_memory() == before_small
    for b in outputs:
        b.free()


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
