// rt_api.cpp - the C ABI of librt_hip.so (include/rt_hip.h): context, uploads,
// launches and read-backs.  No rendering happens on the host: if HIP or the
// GPU is missing every entry point that needs them fails with RT_E_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <algorithm>
#include <string>
#include <vector>

#include "rt_device.h"
#include "rt_scene_build.h"

namespace {

thread_local std::string g_create_error;

struct Ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    std::string err;

    float *d_disc = nullptr, *d_sphere = nullptr, *d_screen = nullptr;
    bool have_tables = false;

    DevScene dsc{};
    std::vector<void *> scene_allocs;
    bool have_scene = false;

    rt_camera cam{};
    bool have_cam = false;

    uint32_t W = 0, H = 0;
    uint32_t n_frames = 0, sel = 0;   // frames the per-pixel buffers hold (rt_begin_frames), and the one the
                                      // single-frame calls work on (rt_select_frame)
    float4 *d_accum = nullptr;        // n_frames x W x H each
    uint32_t *d_packed = nullptr;
    int32_t *d_ids = nullptr;
    rt_camera *d_cams = nullptr;      // the cameras of the last rt_render_frames, n_frames of them
    std::vector<rt_camera> cams_host; // what d_cams holds
    hipEvent_t ev_cams = nullptr;     // the upload of cams_host has read it
    bool cams_in_flight = false;
    // the selected frame's part of the three buffers
    size_t frame_px() const { return (size_t)W * H; }
    float4 *accum() const { return d_accum + frame_px() * sel; }
    uint32_t *packed() const { return d_packed + frame_px() * sel; }
    int32_t *ids() const { return d_ids + frame_px() * sel; }
    StagedRgb *d_stage = nullptr;  // staging ring of the render launches, see DevFrame::stage
    uint32_t stage_waves = 0;      // wavefronts it has room for
    RtkStaticScratch fixed{};      // the device's static scratch (rt_render_static)
    uint32_t *d_block = nullptr;   // this context's bands as a compact block (rt_gather_packed)
    size_t block_px = 0;
    uint32_t *d_gather = nullptr;  // the blocks of all contexts, on the gathering context's device
    size_t gather_px = 0;
    hipEvent_t ev_block = nullptr; // this context's block has arrived in the gather buffer
    unsigned long long *d_segments = nullptr;
    uint32_t *d_queue = nullptr;       // two counters used by alternate launches, see DevFrame::queue
    uint32_t launches = 0;             // render launches so far (picks the counter)
    uint32_t n_cus = 0;                // compute units of the device
    uint32_t disc_plain = 0;           // see DevFrame::disc_plain
    bool have_frame = false;
};

inline Ctx *C(rt_ctx *c) { return reinterpret_cast<Ctx *>(c); }
inline const Ctx *C(const rt_ctx *c) { return reinterpret_cast<const Ctx *>(c); }

int fail(Ctx *c, int code, const std::string &msg) {
    if (c) c->err = msg; else g_create_error = msg;
    return code;
}
int hip_fail(Ctx *c, hipError_t e, const char *what) {
    std::string m = std::string(what) + ": " + hipGetErrorString(e);
    return fail(c, e == hipErrorOutOfMemory ? RT_E_OOM : RT_E_HIP, m);
}
#define HIP_TRY(ctx, call)                                         \
    do {                                                           \
        hipError_t e_ = (call);                                    \
        if (e_ != hipSuccess) return hip_fail((ctx), e_, #call);   \
    } while (0)

void free_scene(Ctx *c) {
    for (void *p : c->scene_allocs) (void)hipFree(p);
    c->scene_allocs.clear();
    c->dsc = DevScene{};
    c->have_scene = false;
}
void free_frame(Ctx *c) {
    if (c->d_accum) (void)hipFree(c->d_accum);
    if (c->d_packed) (void)hipFree(c->d_packed);
    if (c->d_ids) (void)hipFree(c->d_ids);
    if (c->d_cams) (void)hipFree(c->d_cams);
    c->d_cams = nullptr; c->n_frames = 0; c->sel = 0;
    // (callers have synchronised the stream: no upload still reads the copy)
    c->cams_host.clear(); c->cams_in_flight = false;
    if (c->d_block) (void)hipFree(c->d_block);
    if (c->d_gather) (void)hipFree(c->d_gather);
    c->d_block = nullptr; c->d_gather = nullptr; c->block_px = c->gather_px = 0;
    c->d_accum = nullptr; c->d_packed = nullptr; c->d_ids = nullptr;
    c->have_frame = false;
    c->W = c->H = 0;
}

template <typename T>
int upload(Ctx *c, const std::vector<T> &host, const T **dev_out) {
    // never hand the kernel a null array: an empty group still gets one element
    size_t n = host.empty() ? 1 : host.size();
    void *d = nullptr;
    HIP_TRY(c, hipMalloc(&d, n * sizeof(T)));
    c->scene_allocs.push_back(d);
    HIP_TRY(c, hipMemsetAsync(d, 0, n * sizeof(T), c->stream));
    if (!host.empty())
        HIP_TRY(c, hipMemcpyAsync(d, host.data(), host.size() * sizeof(T), hipMemcpyHostToDevice, c->stream));
    *dev_out = static_cast<const T *>(d);
    return RT_OK;
}

// number of 8-row bands b < ceil(H/8) with b % stride == begin
uint32_t bands_of(uint32_t height, uint32_t begin, uint32_t stride) {
    const uint32_t total = (height + RT_BAND_ROWS - 1u) / RT_BAND_ROWS;
    return total > begin ? (total - begin + stride - 1u) / stride : 0u;
}

}  // namespace

extern "C" {

int rt_create(int device, rt_ctx **out) {
    if (!out) return fail(nullptr, RT_E_BADARG, "rt_create: out is null");
    *out = nullptr;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return hip_fail(nullptr, e, "hipGetDeviceCount");
    if (device < 0 || device >= n) return fail(nullptr, RT_E_BADARG, "rt_create: no such device");
    Ctx *c = new (std::nothrow) Ctx();
    if (!c) return fail(nullptr, RT_E_OOM, "rt_create: out of host memory");
    c->device = device;
    auto bail = [&](hipError_t err, const char *what) {
        int rc = hip_fail(nullptr, err, what);
        rt_destroy(reinterpret_cast<rt_ctx *>(c));
        return rc;
    };
    if ((e = hipSetDevice(device)) != hipSuccess) return bail(e, "hipSetDevice");
    if ((e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking)) != hipSuccess) return bail(e, "hipStreamCreate");
    if ((e = hipEventCreate(&c->ev0)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreate(&c->ev1)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_block, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipEventCreateWithFlags(&c->ev_cams, hipEventDisableTiming)) != hipSuccess) return bail(e, "hipEventCreate");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_disc), 128 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_sphere), 768 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_screen), 16384 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_segments), 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c->d_segments, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_queue), 64)) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c->d_queue, 0, 64)) != hipSuccess) return bail(e, "hipMemset");
    int cus = 0;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device)) != hipSuccess) return bail(e, "hipDeviceGetAttribute");
    c->n_cus = cus > 0 ? (uint32_t)cus : 1u;
    // (the staging ring is allocated by the first rt_render, for the kernel its scene selects)
    // address of this device's static scratch, resolved once per device
    if ((e = rtk_static_scratch(device, &c->fixed)) != hipSuccess) return bail(e, "hipGetSymbolAddress(scratch)");
    *out = reinterpret_cast<rt_ctx *>(c);
    return RT_OK;
}

void rt_destroy(rt_ctx *ctx) {
    Ctx *c = C(ctx);
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->stream) (void)hipStreamSynchronize(c->stream);
    free_scene(c);
    free_frame(c);
    if (c->d_disc) (void)hipFree(c->d_disc);
    if (c->d_sphere) (void)hipFree(c->d_sphere);
    if (c->d_screen) (void)hipFree(c->d_screen);
    if (c->d_segments) (void)hipFree(c->d_segments);
    if (c->d_queue) (void)hipFree(c->d_queue);
    if (c->d_stage) (void)hipFree(c->d_stage);
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
    if (c->ev_block) (void)hipEventDestroy(c->ev_block);
    if (c->ev_cams) (void)hipEventDestroy(c->ev_cams);
    if (c->stream) (void)hipStreamDestroy(c->stream);
    delete c;
}

const char *rt_last_error(const rt_ctx *ctx) {
    return ctx ? C(ctx)->err.c_str() : g_create_error.c_str();
}

int rt_upload_tables(rt_ctx *ctx, const float *disc, const float *sphere, const float *screen) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!disc || !sphere || !screen) return fail(c, RT_E_BADARG, "rt_upload_tables: null table");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(c->d_disc, disc, 128 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_sphere, sphere, 768 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipMemcpyAsync(c->d_screen, screen, 16384 * sizeof(float), hipMemcpyHostToDevice, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // caller's memory may go away after return
    c->disc_plain = 1;
    for (int k = 0; k < 128; ++k)
        if (!(std::fabs(disc[k]) >= 0x1p-90f && std::fabs(disc[k]) < 1.0f)) c->disc_plain = 0;
    c->have_tables = true;
    return RT_OK;
}

int rt_upload_scene(rt_ctx *ctx, const rt_scene *s) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!s) return fail(c, RT_E_BADARG, "rt_upload_scene: scene is null");
    // RT_HIP_FLAT_SCAN=1 keeps long runs in list order (no tree):
    // the brute-force scan whose work matches the reference's linear list.
    const char *flat = std::getenv("RT_HIP_FLAT_SCAN");
    const bool use_tree = !(flat && flat[0] && flat[0] != '0');
    HostScene hs;
    std::string err;
    int rc;
    try {
        rc = build_host_scene(s, use_tree, hs, err);
    } catch (const std::bad_alloc &) {
        return fail(c, RT_E_OOM, "rt_upload_scene: out of host memory");
    }
    if (rc != RT_OK) return fail(c, rc, "rt_upload_scene: " + err);

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // no render may still read the old scene
    free_scene(c);
    DevScene d{};
    d.n_runs = (uint32_t)hs.runs.size();
    d.need_ff_shared = hs.need_ff_shared;
    d.lds_run = hs.lds_run; d.lds_level = hs.lds_level;
    d.lds_first = hs.lds_first; d.lds_count = hs.lds_count;
    d.lds_deep_level = hs.lds_deep_level; d.lds_deep_first = hs.lds_deep_first; d.lds_deep_count = hs.lds_deep_count;
    d.lds_kind = hs.lds_kind;
    d.modes = hs.modes;
    d.cheap_ok = hs.cheap_ok;
    if ((rc = upload(c, hs.runs, &d.runs)) != RT_OK || (rc = upload(c, hs.sph_geom, &d.sph_geom)) != RT_OK ||
        (rc = upload(c, hs.sph_mat, &d.sph_mat)) != RT_OK || (rc = upload(c, hs.sph_id, &d.sph_id)) != RT_OK ||
        (rc = upload(c, hs.planes, &d.planes)) != RT_OK || (rc = upload(c, hs.cubes, &d.cubes)) != RT_OK ||
        (rc = upload(c, hs.tris, &d.tris)) != RT_OK || (rc = upload(c, hs.tri_id, &d.tri_id)) != RT_OK ||
        (rc = upload(c, hs.cull_rec, &d.cull_rec)) != RT_OK ||
        (rc = upload(c, hs.tree_rec, &d.tree_rec)) != RT_OK || (rc = upload(c, hs.order, &d.order)) != RT_OK ||
        (rc = upload(c, hs.mat_rec, &d.mat_rec)) != RT_OK) {
        std::string keep = c->err;
        free_scene(c);
        c->err = keep;
        return rc;
    }
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // host vectors die at return
    c->dsc = d;
    c->have_scene = true;
    return RT_OK;
}

int rt_set_camera(rt_ctx *ctx, const rt_camera *cam) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!cam) return fail(c, RT_E_BADARG, "rt_set_camera: camera is null");
    c->cam = *cam;
    c->have_cam = true;
    return RT_OK;
}

int rt_begin_frames(rt_ctx *ctx, uint32_t width, uint32_t height, uint32_t n_frames) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (width < 2 || height < 2 || width > 65536u || height > 65536u || (uint64_t)width * height > 0xffffffffull)
        return fail(c, RT_E_BADARG, "rt_begin_frame: width and height must be in 2..65536 and width*height below 2^32");
    if (n_frames == 0 || n_frames > RT_MAX_FRAMES || (uint64_t)width * height * n_frames > 0xffffffffull)
        return fail(c, RT_E_BADARG, "rt_begin_frames: n_frames must be in 1..RT_MAX_FRAMES and all frames together below 2^32 pixels");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_px = (size_t)width * height * n_frames;
    if (!c->have_frame || c->W != width || c->H != height || c->n_frames != n_frames) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_frame(c);
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_accum), n_px * sizeof(float4)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_packed), n_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_ids), n_px * sizeof(int32_t)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_cams), n_frames * sizeof(rt_camera)));
        c->W = width; c->H = height; c->n_frames = n_frames;
        c->have_frame = true;
    }
    c->sel = 0;
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, n_px * sizeof(float4), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_packed, 0, n_px * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_ids, 0xff, n_px * sizeof(int32_t), c->stream));   // -1 = nothing recorded
    HIP_TRY(c, hipMemsetAsync(c->d_segments, 0, 8 * sizeof(unsigned long long), c->stream));
    return RT_OK;
}

int rt_begin_frame(rt_ctx *ctx, uint32_t width, uint32_t height) { return rt_begin_frames(ctx, width, height, 1u); }

int rt_select_frame(rt_ctx *ctx, uint32_t frame) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!c->have_frame) return fail(c, RT_E_STATE, "rt_select_frame: no frame");
    if (frame >= c->n_frames) return fail(c, RT_E_BADARG, "rt_select_frame: frame beyond those of rt_begin_frames");
    c->sel = frame;
    return RT_OK;
}

// argument checks shared by rt_render and rt_render_static
static int check_render(Ctx *c, const rt_params *p, const char *who) {
    const std::string w(who);
    if (!p) return fail(c, RT_E_BADARG, w + ": params is null");
    if (p->band_stride == 0 || p->band_begin >= p->band_stride)
        return fail(c, RT_E_BADARG, w + ": need band_begin < band_stride");
    if (p->max_bounces > RT_MAX_BOUNCES) return fail(c, RT_E_BADARG, w + ": max_bounces above RT_MAX_BOUNCES");
    if ((uint64_t)p->sample_begin + p->sample_count > p->spp)
        return fail(c, RT_E_BADARG, w + ": sample range exceeds spp");
    if (p->sampling_mode != RT_SAMPLING_DET) return fail(c, RT_E_BADARG, w + ": unknown sampling_mode");
    if (!(p->t_min == p->t_min)) return fail(c, RT_E_BADARG, w + ": t_min is NaN");
    return RT_OK;
}

// One launch per DEV_SAMPLE_CHUNK samples: the colours of a launch's samples are staged in the
// ring behind fr.stage until their pixel's work unit is complete.  `clear` (may be null): bytes
// zeroed on the stream before every launch (the queue counters of a scratch that other calls
// share); otherwise the two counters behind `queue` alternate and the running launch zeroes
// the other one.
static int enqueue_render(Ctx *c, const rt_params *p, DevFrame fr, uint32_t stage_waves, uint32_t *queue,
                          void *clear, size_t clear_bytes, const rt_camera *cams = nullptr, uint32_t n_frames = 1u) {
    const uint32_t n_bands = bands_of(p->height, p->band_begin, p->band_stride);
    if (n_bands == 0 || p->sample_count == 0) return RT_OK;
    rt_params part = *p;
    fr.disc_plain = c->disc_plain;
    // development aid: RT_HIP_UNIT_PX=1|2|4|8|16|32|64 fixes the pixels of a tile per work unit
    // (default: by launch size, see rtk_launch_render); a unit's samples must fit a ring slot
    const char *up = std::getenv("RT_HIP_UNIT_PX");
    int unit_px_log2 = -1;
    for (int k = 0; up && k <= 6; ++k) if (std::atoi(up) == (1 << k)) unit_px_log2 = k;
    uint32_t chunk = DEV_SAMPLE_CHUNK;
    if (unit_px_log2 >= 0) chunk = std::min<uint32_t>(chunk, (1u << DEV_STAGE_ITEM_BITS) >> unit_px_log2);
    for (uint32_t done = 0; done < p->sample_count; done += chunk) {
        part.sample_begin = p->sample_begin + done;
        part.sample_count = std::min(chunk, p->sample_count - done);
        if (clear) {
            HIP_TRY(c, hipMemsetAsync(clear, 0, clear_bytes, c->stream));
            fr.queue = queue; fr.queue_next = queue + 1;
        } else {
            fr.queue = queue + (c->launches & 1u);
            fr.queue_next = queue + ((c->launches + 1u) & 1u);
        }
        HIP_TRY(c, rtk_launch_render(c->dsc, c->cam, part, fr, n_bands, c->n_cus, stage_waves, unit_px_log2, c->stream,
                                     cams, n_frames));
        if (!clear) c->launches += 1u;     // only a launch that ran has zeroed the other counter
    }
    return RT_OK;
}

// The staging ring of rt_render: one part (24 KiB) per wavefront a launch of the scene's kernel can
// have resident on this device - on 256 CUs 8,192 wavefronts = 192 MiB for a scene without a box
// tree, half of that for one with a tree.  Allocated at the first render and kept; grows when a later
// scene needs more.
static int ensure_stage(Ctx *c) {
    const uint32_t need = rtk_resident_waves(c->n_cus, c->dsc.modes);
    if (c->d_stage && c->stage_waves >= need) return RT_OK;
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    if (c->d_stage) (void)hipFree(c->d_stage);
    c->d_stage = nullptr; c->stage_waves = 0;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_stage), (size_t)need * DEV_STAGE_WAVE_BYTES));
    c->stage_waves = need;
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_params *p) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    int rc = check_render(c, p, "rt_render");
    if (rc != RT_OK) return rc;
    if (!c->have_tables || !c->have_scene || !c->have_cam || !c->have_frame)
        return fail(c, RT_E_STATE, "rt_render: tables, scene, camera and frame must be set first");
    if (p->width != c->W || p->height != c->H)
        return fail(c, RT_E_BADARG, "rt_render: width/height differ from rt_begin_frame");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_stage(c)) != RT_OK) return rc;
    DevFrame fr{c->d_disc, c->d_sphere, c->d_screen, c->accum(), c->ids(), c->d_stage, c->d_segments,
                nullptr, nullptr, 0u};
    return enqueue_render(c, p, fr, c->stage_waves, c->d_queue, nullptr, 0);
}

int rt_render_frames(rt_ctx *ctx, const rt_params *p, const rt_camera *cameras, uint32_t n_frames) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    int rc = check_render(c, p, "rt_render_frames");
    if (rc != RT_OK) return rc;
    if (!c->have_tables || !c->have_scene || !c->have_frame)
        return fail(c, RT_E_STATE, "rt_render_frames: tables, scene and frames must be set first");
    if (!cameras || n_frames == 0 || n_frames > c->n_frames)
        return fail(c, RT_E_BADARG, "rt_render_frames: no cameras, or more of them than rt_begin_frames made frames");
    if (p->width != c->W || p->height != c->H)
        return fail(c, RT_E_BADARG, "rt_render_frames: width/height differ from rt_begin_frames");
    HIP_TRY(c, hipSetDevice(c->device));
    if ((rc = ensure_stage(c)) != RT_OK) return rc;
    if (rtk_has_frames_kernel(c->dsc.modes) && n_frames > 1u) {
        // one launch (per 512 samples) over the tiles of all frames; the cameras go to device memory first,
        // through a copy of the context's own (the caller's may go away after the call), on the stream, behind
        // the launches that still read the last ones.  Not when they are the cameras that are there already:
        // progressive calls on the same frames.
        if (c->cams_host.size() != n_frames || std::memcmp(c->cams_host.data(), cameras, n_frames * sizeof(rt_camera)) != 0) {
            // (the last upload must have read the context's copy before it is overwritten)
            if (c->cams_in_flight) HIP_TRY(c, hipEventSynchronize(c->ev_cams));
            c->cams_in_flight = false;
            c->cams_host.assign(cameras, cameras + n_frames);
            HIP_TRY(c, hipMemcpyAsync(c->d_cams, c->cams_host.data(), n_frames * sizeof(rt_camera), hipMemcpyHostToDevice, c->stream));
            HIP_TRY(c, hipEventRecord(c->ev_cams, c->stream));
            c->cams_in_flight = true;
        }
        DevFrame fr{c->d_disc, c->d_sphere, c->d_screen, c->d_accum, c->d_ids, c->d_stage, c->d_segments,
                    nullptr, nullptr, 0u};
        return enqueue_render(c, p, fr, c->stage_waves, c->d_queue, nullptr, 0, c->d_cams, n_frames);
    }
    // a single frame, or a scene with long runs: the kernels for those take one frame per launch, and a
    // frame of such a scene keeps a launch busy by itself
    const rt_camera keep = c->cam;
    const uint32_t keep_sel = c->sel;
    for (uint32_t f = 0; f < n_frames && rc == RT_OK; ++f) {
        c->cam = cameras[f]; c->sel = f;
        DevFrame fr{c->d_disc, c->d_sphere, c->d_screen, c->accum(), c->ids(), c->d_stage, c->d_segments,
                    nullptr, nullptr, 0u};
        rc = enqueue_render(c, p, fr, c->stage_waves, c->d_queue, nullptr, 0);
    }
    c->cam = keep; c->sel = keep_sel;
    return rc;
}

int rt_render_static(rt_ctx *ctx, const rt_params *p, void *accum_device, int32_t *ids_device,
                     uint64_t *segments_device) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    int rc = check_render(c, p, "rt_render_static");
    if (rc != RT_OK) return rc;
    if (!c->have_tables || !c->have_scene || !c->have_cam)
        return fail(c, RT_E_STATE, "rt_render_static: tables, scene and camera must be set first");
    if (!accum_device || !ids_device) return fail(c, RT_E_BADARG, "rt_render_static: accum_device or ids_device is null");
    if (p->width < 2 || p->height < 2 || p->width > 65536u || p->height > 65536u ||
        (uint64_t)p->width * p->height > RT_STATIC_MAX_PIXELS)
        return fail(c, RT_E_BADARG, "rt_render_static: frame outside 2..65536 per side or above RT_STATIC_MAX_PIXELS pixels");
    HIP_TRY(c, hipSetDevice(c->device));
    DevFrame fr{c->d_disc, c->d_sphere, c->d_screen, static_cast<float4 *>(accum_device), ids_device,
                c->fixed.stage, segments_device ? reinterpret_cast<unsigned long long *>(segments_device) : c->fixed.segments,
                nullptr, nullptr, 0u};
    return enqueue_render(c, p, fr, c->fixed.stage_waves, c->fixed.queue, c->fixed.queue, c->fixed.head_bytes);
}

int rt_pack(rt_ctx *ctx) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!c->have_frame) return fail(c, RT_E_STATE, "rt_pack: no frame");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, rtk_launch_pack(c->accum(), c->packed(), c->W * c->H, c->stream));
    return RT_OK;
}

int rt_sync(rt_ctx *ctx) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

static int read_back(Ctx *c, void *dst, const void *src, size_t bytes) {
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, c->stream));
    HIP_TRY(c, hipStreamSynchronize(c->stream));
    return RT_OK;
}

static int check_read(Ctx *c, const void *out, size_t n_px, const char *who) {
    if (!c) return RT_E_BADARG;
    if (!c->have_frame) return fail(c, RT_E_STATE, std::string(who) + ": no frame");
    if (!out || n_px != (size_t)c->W * c->H) return fail(c, RT_E_BADARG, std::string(who) + ": n_px must be width*height");
    return RT_OK;
}

int rt_read_accum(rt_ctx *ctx, float *rgb, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, rgb, n_px, "rt_read_accum");
    if (rc != RT_OK) return rc;
    std::vector<float4> tmp;
    try { tmp.resize(n_px); } catch (...) { return fail(c, RT_E_OOM, "rt_read_accum: out of host memory"); }
    if ((rc = read_back(c, tmp.data(), c->accum(), n_px * sizeof(float4))) != RT_OK) return rc;
    for (size_t k = 0; k < n_px; ++k) { rgb[3 * k] = tmp[k].x; rgb[3 * k + 1] = tmp[k].y; rgb[3 * k + 2] = tmp[k].z; }
    return RT_OK;
}

int rt_read_sample_counts(rt_ctx *ctx, uint32_t *counts, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, counts, n_px, "rt_read_sample_counts");
    if (rc != RT_OK) return rc;
    std::vector<float4> tmp;
    try { tmp.resize(n_px); } catch (...) { return fail(c, RT_E_OOM, "rt_read_sample_counts: out of host memory"); }
    if ((rc = read_back(c, tmp.data(), c->accum(), n_px * sizeof(float4))) != RT_OK) return rc;
    for (size_t k = 0; k < n_px; ++k) counts[k] = (uint32_t)tmp[k].w;
    return RT_OK;
}

int rt_read_packed(rt_ctx *ctx, uint32_t *out, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, out, n_px, "rt_read_packed");
    if (rc != RT_OK) return rc;
    if ((rc = rt_pack(ctx)) != RT_OK) return rc;
    return read_back(c, out, c->packed(), n_px * sizeof(uint32_t));
}

int rt_read_primary_ids(rt_ctx *ctx, int32_t *ids, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, ids, n_px, "rt_read_primary_ids");
    if (rc != RT_OK) return rc;
    return read_back(c, ids, c->ids(), n_px * sizeof(int32_t));
}

int rt_ray_count(rt_ctx *ctx, uint64_t *segments) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!segments) return fail(c, RT_E_BADARG, "rt_ray_count: out is null");
    unsigned long long v = 0;
    int rc = read_back(c, &v, c->d_segments, sizeof v);
    if (rc != RT_OK) return rc;
    *segments = v;
    return RT_OK;
}

// Instrumentation counters of kernels built with -DRTK_STATS (slot 0 is the
// segment count; the others stay zero in the shipped build).  Development aid,
// deliberately not part of rt_hip.h.
// development aid: the staging buffer (instrumented builds log into it)
void *rtdbg_stage_ptr(rt_ctx *ctx) { return C(ctx) ? C(ctx)->d_stage : nullptr; }
int rtdbg_read_counters(rt_ctx *ctx, uint64_t out[8]) {
    Ctx *c = C(ctx);
    if (!c || !out) return RT_E_BADARG;
    unsigned long long v[8] = {0};
    int rc = read_back(c, v, c->d_segments, sizeof v);
    if (rc != RT_OK) return rc;
    for (int k = 0; k < 8; ++k) out[k] = v[k];
    return RT_OK;
}

int rt_device_buffers(rt_ctx *ctx, void **accum, void **packed, void **ids, void **stream) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    // (without a frame the buffers are null and only the stream is handed out: callers of
    // rt_render_static need no frame but must be able to order their own work against the stream)
    if (accum) *accum = c->have_frame ? c->accum() : nullptr;
    if (packed) *packed = c->have_frame ? c->packed() : nullptr;
    if (ids) *ids = c->have_frame ? c->ids() : nullptr;
    if (stream) *stream = c->stream;
    return RT_OK;
}

int rt_export_packed_bands(rt_ctx *ctx, uint32_t *dst_device, size_t capacity_px, uint32_t band_begin,
                           uint32_t band_stride) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!c->have_frame) return fail(c, RT_E_STATE, "rt_export_packed_bands: no frame");
    if (!dst_device || band_stride == 0 || band_begin >= band_stride)
        return fail(c, RT_E_BADARG, "rt_export_packed_bands: null destination or band_begin >= band_stride");
    const uint32_t n_bands = bands_of(c->H, band_begin, band_stride);
    if ((size_t)n_bands * RT_BAND_ROWS * c->W > capacity_px)
        return fail(c, RT_E_BADARG, "rt_export_packed_bands: destination too small");
    if (n_bands == 0) return RT_OK;
    if (n_bands * RT_BAND_ROWS > 65535u) return fail(c, RT_E_BADARG, "rt_export_packed_bands: too many rows");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, rtk_launch_pack_bands(c->accum(), dst_device, c->W, c->H, band_begin, band_stride, n_bands,
                                     c->stream));
    return RT_OK;
}

int rt_gather_packed(rt_ctx *const *ctxs, uint32_t n_ctx, uint32_t *out, size_t n_px) {
    if (!ctxs || n_ctx == 0 || !ctxs[0]) return RT_E_BADARG;
    Ctx *root = C(ctxs[0]);
    if (!out) return fail(root, RT_E_BADARG, "rt_gather_packed: out is null");
    if (!root->have_frame) return fail(root, RT_E_STATE, "rt_gather_packed: no frame");
    const uint32_t W = root->W, H = root->H;
    if (n_px != (size_t)W * H) return fail(root, RT_E_BADARG, "rt_gather_packed: n_px must be width*height");
    for (uint32_t g = 0; g < n_ctx; ++g) {
        Ctx *c = C(ctxs[g]);
        if (!c || !c->have_frame || c->W != W || c->H != H)
            return fail(root, RT_E_BADARG, "rt_gather_packed: every context needs a frame of the same size");
        for (uint32_t k = 0; k < g; ++k)
            if (ctxs[k] == ctxs[g]) return fail(root, RT_E_BADARG, "rt_gather_packed: a context is listed twice");
    }
    // every block is padded to the band count of context 0, the largest
    const size_t block_px = (size_t)bands_of(H, 0, n_ctx) * RT_BAND_ROWS * W;
    // development aid: RT_HIP_GATHER_STAGED=1 sends the blocks of contexts on the root's own GPU through
    // the staging block and a peer copy as well (a peer copy within one device is valid), so that the
    // path several GPUs take can be exercised on one
    const char *staged_env = std::getenv("RT_HIP_GATHER_STAGED");
    const bool force_staged = staged_env && staged_env[0] && staged_env[0] != '0';
    // a failure on another context is reported where the caller looks for it
    auto on_root = [&](Ctx *c, int rc) { if (c != root) root->err = c->err; return rc; };
#define HIP_TRY_ON(ctx, call)                                                       \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) return on_root((ctx), hip_fail((ctx), e_, #call));    \
    } while (0)
    HIP_TRY(root, hipSetDevice(root->device));
    if (root->gather_px < block_px * n_ctx) {
        HIP_TRY(root, hipStreamSynchronize(root->stream));
        if (root->d_gather) (void)hipFree(root->d_gather);
        root->d_gather = nullptr; root->gather_px = 0;
        HIP_TRY(root, hipMalloc(reinterpret_cast<void **>(&root->d_gather), block_px * n_ctx * sizeof(uint32_t)));
        root->gather_px = block_px * n_ctx;
    }
    for (uint32_t g = 0; g < n_ctx; ++g) {
        Ctx *c = C(ctxs[g]);
        HIP_TRY_ON(c, hipSetDevice(c->device));
        uint32_t *dst = root->d_gather + block_px * g;
        // the previous gather's reader must be done with the gather buffer before anything writes it
        if (c != root) HIP_TRY_ON(c, hipStreamWaitEvent(c->stream, root->ev_block, 0));
        if (c->device == root->device && !(force_staged && c != root)) {
            // same GPU: quantise straight into the gather buffer
            int rc = rt_export_packed_bands(ctxs[g], dst, block_px, g, n_ctx);
            if (rc != RT_OK) return on_root(c, rc);
        } else {
            if (c->block_px < block_px) {
                HIP_TRY_ON(c, hipStreamSynchronize(c->stream));
                if (c->d_block) (void)hipFree(c->d_block);
                c->d_block = nullptr; c->block_px = 0;
                HIP_TRY_ON(c, hipMalloc(reinterpret_cast<void **>(&c->d_block), block_px * sizeof(uint32_t)));
                c->block_px = block_px;
            }
            int rc = rt_export_packed_bands(ctxs[g], c->d_block, block_px, g, n_ctx);
            if (rc != RT_OK) return on_root(c, rc);
            // one copy per peer over its own link to the gathering GPU
            HIP_TRY_ON(c, hipMemcpyPeerAsync(dst, root->device, c->d_block, c->device, block_px * sizeof(uint32_t), c->stream));
        }
        if (c != root) HIP_TRY_ON(c, hipEventRecord(c->ev_block, c->stream));
    }
#undef HIP_TRY_ON
    HIP_TRY(root, hipSetDevice(root->device));
    for (uint32_t g = 1; g < n_ctx; ++g) HIP_TRY(root, hipStreamWaitEvent(root->stream, C(ctxs[g])->ev_block, 0));
    HIP_TRY(root, rtk_launch_compose_bands(root->d_gather, root->packed(), W, H, n_ctx, block_px, root->stream));
    // the next gather's writers wait for this one's reader (contexts that share the root's GPU write
    // into the gather buffer directly)
    HIP_TRY(root, hipEventRecord(root->ev_block, root->stream));
    return read_back(root, out, root->packed(), n_px * sizeof(uint32_t));
}

int rt_check_arith(rt_ctx *ctx, uint64_t operands, uint32_t seed, uint64_t mismatches[2]) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!mismatches || operands == 0 || operands > (1ull << 40)) return fail(c, RT_E_BADARG, "rt_check_arith: null result or count out of range");
    HIP_TRY(c, hipSetDevice(c->device));
    unsigned long long *bad = nullptr;
    HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&bad), 2 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(bad, 0, 2 * sizeof(unsigned long long), c->stream);
    const uint32_t rounds = 256u;
    const uint64_t per_block = 256ull * rounds;
    uint64_t blocks = (operands + per_block - 1) / per_block;
    unsigned long long host[2] = {0, 0};
    for (uint32_t part = 0; e == hipSuccess && blocks != 0; ++part) {
        const uint32_t now = (uint32_t)std::min<uint64_t>(blocks, 1u << 20);
        e = rtk_launch_check_arith(now, rounds, seed + 0x9E3779B9u * part, bad, c->stream);
        blocks -= now;
    }
    if (e == hipSuccess) e = hipMemcpyAsync(host, bad, sizeof(host), hipMemcpyDeviceToHost, c->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(c->stream);
    (void)hipFree(bad);
    if (e != hipSuccess) return hip_fail(c, e, "rt_check_arith");
    mismatches[0] = host[0]; mismatches[1] = host[1];
    return RT_OK;
}

int rt_kernel_info(rt_ctx *ctx, uint32_t max_bounces, uint32_t *scan_modes, uint32_t *depth_budget) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!c->have_scene) return fail(c, RT_E_STATE, "rt_kernel_info: no scene");
    if (max_bounces > RT_MAX_BOUNCES) return fail(c, RT_E_BADARG, "rt_kernel_info: max_bounces above RT_MAX_BOUNCES");
    uint32_t modes = 0, depth = 0;
    rtk_kernel_variant(c->dsc.modes, max_bounces, &modes, &depth);
    if (scan_modes) *scan_modes = modes;
    if (depth_budget) *depth_budget = depth;
    return RT_OK;
}

#ifndef RT_BUILD_ID
#define RT_BUILD_ID "unknown"
#endif
const char *rt_build_id(void) { return RT_BUILD_ID; }

int rt_timer_start(rt_ctx *ctx) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ev0, c->stream));
    return RT_OK;
}

int rt_timer_stop(rt_ctx *ctx, float *milliseconds) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!milliseconds) return fail(c, RT_E_BADARG, "rt_timer_stop: out is null");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipEventRecord(c->ev1, c->stream));
    HIP_TRY(c, hipEventSynchronize(c->ev1));
    HIP_TRY(c, hipEventElapsedTime(milliseconds, c->ev0, c->ev1));
    return RT_OK;
}

}  // extern "C"
