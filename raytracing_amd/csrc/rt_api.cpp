// rt_api.cpp - the C ABI of librt_hip.so (include/rt_hip.h): context, uploads,
// launches and read-backs.  No rendering happens on the host: if HIP or the
// GPU is missing every entry point that needs them fails with RT_E_HIP.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

#include "rt_device.h"

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
    float4 *d_accum = nullptr;
    uint32_t *d_packed = nullptr;
    int32_t *d_ids = nullptr;
    unsigned long long *d_segments = nullptr;
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

inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }

struct ListEntry { int kind = -1; uint32_t k = 0; };

// Threshold of the kernel's conservative sphere rejection test (see cull_pass):
//   thr = r^2 (1+eps) - |c|^2 (1-eps),  eps = 2^-16,
// evaluated in double and rounded UP to float so rounding can only make the
// test reject less.  Spheres whose magnitudes would leave the range where the
// kernel's error bound holds (squares near overflow or underflow) get +inf:
// never rejected, always handed to the exact test.
float cull_threshold(float cx, float cy, float cz, float r) {
    const double eps = 1.0 / 65536.0;
    const double c2 = (double)cx * cx + (double)cy * cy + (double)cz * cz;
    const double r2 = (double)r * r;
    if (!(c2 < 1e30) || !(r2 < 1e30) || !(c2 + r2 > 1e-24)) return INFINITY;
    const double thr = r2 * (1.0 + eps) - c2 * (1.0 - eps);
    float f = (float)thr;
    if ((double)f < thr) f = std::nextafterf(f, INFINITY);
    return f;
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
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_disc), 128 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_sphere), 768 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_screen), 16384 * sizeof(float))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMalloc(reinterpret_cast<void **>(&c->d_segments), 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMalloc");
    if ((e = hipMemset(c->d_segments, 0, 8 * sizeof(unsigned long long))) != hipSuccess) return bail(e, "hipMemset");
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
    if (c->ev0) (void)hipEventDestroy(c->ev0);
    if (c->ev1) (void)hipEventDestroy(c->ev1);
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
    c->have_tables = true;
    return RT_OK;
}

int rt_upload_scene(rt_ctx *ctx, const rt_scene *s) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!s) return fail(c, RT_E_BADARG, "rt_upload_scene: scene is null");
    const uint64_t n64 = (uint64_t)s->n_spheres + s->n_planes + s->n_cubes + s->n_tris;
    if (n64 > 0x7fffffffu) return fail(c, RT_E_BADARG, "rt_upload_scene: too many primitives");
    const uint32_t n = (uint32_t)n64;
    if ((s->n_spheres && !(s->sph_cx && s->sph_cy && s->sph_cz && s->sph_r && s->sph_mat && s->sph_id)) ||
        (s->n_planes && !(s->pl_nx && s->pl_ny && s->pl_nz && s->pl_px && s->pl_py && s->pl_pz && s->pl_mat && s->pl_id)) ||
        (s->n_cubes && !(s->cb_cx && s->cb_cy && s->cb_cz && s->cb_w && s->cb_h && s->cb_d && s->cb_mat && s->cb_id)) ||
        (s->n_tris && !(s->tri_v0x && s->tri_v0y && s->tri_v0z && s->tri_v1x && s->tri_v1y && s->tri_v1z &&
                        s->tri_v2x && s->tri_v2y && s->tri_v2z && s->tri_mat && s->tri_id)) ||
        (s->n_materials && !(s->mat_kind && s->mat_r && s->mat_g && s->mat_b && s->mat_param)))
        return fail(c, RT_E_BADARG, "rt_upload_scene: null array in a non-empty group");
    if (n > 0 && s->n_materials == 0) return fail(c, RT_E_BADARG, "rt_upload_scene: primitives but no materials");
    for (uint32_t m = 0; m < s->n_materials; ++m)
        if (s->mat_kind[m] < RT_METAL || s->mat_kind[m] > RT_DIELECTRIC)
            return fail(c, RT_E_BADARG, "rt_upload_scene: unknown material kind");

    // list order: ids of all groups must be a permutation of 0..n-1
    std::vector<ListEntry> list;
    try { list.resize(n); } catch (...) { return fail(c, RT_E_OOM, "rt_upload_scene: out of host memory"); }
    auto place = [&](uint32_t id, uint32_t mat, int kind, uint32_t k) -> bool {
        if (id >= n || list[id].kind != -1 || mat >= s->n_materials) return false;
        list[id].kind = kind; list[id].k = k;
        return true;
    };
    bool ok = true;
    for (uint32_t k = 0; ok && k < s->n_spheres; ++k) ok = place(s->sph_id[k], s->sph_mat[k], DEV_SPHERE, k);
    for (uint32_t k = 0; ok && k < s->n_planes; ++k) ok = place(s->pl_id[k], s->pl_mat[k], DEV_PLANE, k);
    for (uint32_t k = 0; ok && k < s->n_cubes; ++k) ok = place(s->cb_id[k], s->cb_mat[k], DEV_CUBE, k);
    for (uint32_t k = 0; ok && k < s->n_tris; ++k) ok = place(s->tri_id[k], s->tri_mat[k], DEV_TRI, k);
    if (!ok) return fail(c, RT_E_BADARG, "rt_upload_scene: ids are not a permutation of the list, or a material index is out of range");

    std::vector<uint4> runs;
    std::vector<float4> sph_geom, sph_cull, planes, cubes, tris, mat_albedo;
    std::vector<uint32_t> sph_mat;
    std::vector<int32_t> mat_kind;
    uint32_t need_ff_shared = 0;
    try {
        for (uint32_t m = 0; m < s->n_materials; ++m) {
            mat_albedo.push_back(make_float4(s->mat_r[m], s->mat_g[m], s->mat_b[m], s->mat_param[m]));
            mat_kind.push_back(s->mat_kind[m]);
        }
        auto pad_spheres = [&]() {
            while (sph_geom.size() % DEV_SPHERE_BATCH) {
                // a sphere no ray can reach: the conservative test always rejects it
                sph_geom.push_back(make_float4(0.f, 0.f, 0.f, 0.f));
                sph_cull.push_back(make_float4(0.f, 0.f, 0.f, -INFINITY));   // g > -inf: always rejected
                sph_mat.push_back(0);
            }
        };
        for (uint32_t id = 0; id < n; ++id) {
            const int kind = list[id].kind;
            const uint32_t k = list[id].k;
            uint32_t slot = 0;
            if (kind == DEV_SPHERE) slot = (uint32_t)sph_geom.size();
            else if (kind == DEV_PLANE) slot = (uint32_t)(planes.size() / 2);
            else if (kind == DEV_CUBE) slot = (uint32_t)(cubes.size() / 4);
            else slot = (uint32_t)(tris.size() / 3);
            if (runs.empty() || (int)runs.back().x != kind) {
                if (kind == DEV_SPHERE) { pad_spheres(); slot = (uint32_t)sph_geom.size(); }
                runs.push_back(make_uint4((uint32_t)kind, slot, 0u, id));
            }
            runs.back().z += 1;
            if (kind == DEV_SPHERE) {
                const float cx = s->sph_cx[k], cy = s->sph_cy[k], cz = s->sph_cz[k], r = s->sph_r[k];
                sph_geom.push_back(make_float4(cx, cy, cz, r));
                sph_cull.push_back(make_float4(cx, cy, cz, cull_threshold(cx, cy, cz, r)));
                sph_mat.push_back(s->sph_mat[k]);
            } else if (kind == DEV_PLANE) {
                planes.push_back(make_float4(s->pl_nx[k], s->pl_ny[k], s->pl_nz[k], bits_to_float(s->pl_mat[k])));
                planes.push_back(make_float4(s->pl_px[k], s->pl_py[k], s->pl_pz[k], 0.f));
            } else if (kind == DEV_CUBE) {
                // Cube::new, src/geometry/mod.rs:146-154: corners from center and extents;
                // w/2 is also what Cube::hit divides by (:213-215)
                const float cx = s->cb_cx[k], cy = s->cb_cy[k], cz = s->cb_cz[k];
                const float w = s->cb_w[k], h = s->cb_h[k], d = s->cb_d[k];
                const float hw = w / 2.0f, hh = h / 2.0f, hd = d / 2.0f;
                const float c0x = cx + (-w / 2.0f), c0y = cy + (-h / 2.0f), c0z = cz + (-d / 2.0f);
                const float c1x = cx + hw, c1y = cy + hh, c1z = cz + hd;
                cubes.push_back(make_float4(cx, cy, cz, bits_to_float(s->cb_mat[k])));
                cubes.push_back(make_float4(hw, hh, hd, 0.f));
                cubes.push_back(make_float4(c0x, c0y, c0z, 0.f));
                cubes.push_back(make_float4(c1x, c1y, c1z, 0.f));
                if (s->mat_kind[s->cb_mat[k]] == RT_DIELECTRIC) need_ff_shared = 1;
            } else {
                const float ax = s->tri_v0x[k], ay = s->tri_v0y[k], az = s->tri_v0z[k];
                const float e1x = s->tri_v1x[k] - ax, e1y = s->tri_v1y[k] - ay, e1z = s->tri_v1z[k] - az;
                const float e2x = s->tri_v2x[k] - ax, e2y = s->tri_v2y[k] - ay, e2z = s->tri_v2z[k] - az;
                tris.push_back(make_float4(ax, ay, az, bits_to_float(s->tri_mat[k])));
                tris.push_back(make_float4(e1x, e1y, e1z, 0.f));
                tris.push_back(make_float4(e2x, e2y, e2z, 0.f));
            }
        }
        pad_spheres();
    } catch (const std::bad_alloc &) {
        return fail(c, RT_E_OOM, "rt_upload_scene: out of host memory");
    }

    // The kernel reads the rejection records two spheres at a time:
    // {cx0,cx1, cy0,cy1, cz0,cz1, thr0,thr1}.  Runs start at even slots
    // (padded to multiples of DEV_SPHERE_BATCH), so pairs never straddle runs.
    for (size_t k = 0; k + 1 < sph_cull.size(); k += 2) {
        const float4 a = sph_cull[k], b = sph_cull[k + 1];
        sph_cull[k] = make_float4(a.x, b.x, a.y, b.y);
        sph_cull[k + 1] = make_float4(a.z, b.z, a.w, b.w);
    }

    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, hipStreamSynchronize(c->stream));   // no render may still read the old scene
    free_scene(c);
    DevScene d{};
    d.n_runs = (uint32_t)runs.size();
    d.need_ff_shared = need_ff_shared;
    int rc;
    if ((rc = upload(c, runs, &d.runs)) != RT_OK || (rc = upload(c, sph_geom, &d.sph_geom)) != RT_OK ||
        (rc = upload(c, sph_cull, &d.sph_cull)) != RT_OK || (rc = upload(c, sph_mat, &d.sph_mat)) != RT_OK ||
        (rc = upload(c, planes, &d.planes)) != RT_OK || (rc = upload(c, cubes, &d.cubes)) != RT_OK ||
        (rc = upload(c, tris, &d.tris)) != RT_OK || (rc = upload(c, mat_albedo, &d.mat_albedo)) != RT_OK ||
        (rc = upload(c, mat_kind, &d.mat_kind)) != RT_OK) {
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

int rt_begin_frame(rt_ctx *ctx, uint32_t width, uint32_t height) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (width < 2 || height < 2 || width > 65536u || height > 65536u)
        return fail(c, RT_E_BADARG, "rt_begin_frame: width and height must be in 2..65536");
    HIP_TRY(c, hipSetDevice(c->device));
    const size_t n_px = (size_t)width * height;
    if (!c->have_frame || c->W != width || c->H != height) {
        HIP_TRY(c, hipStreamSynchronize(c->stream));
        free_frame(c);
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_accum), n_px * sizeof(float4)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_packed), n_px * sizeof(uint32_t)));
        HIP_TRY(c, hipMalloc(reinterpret_cast<void **>(&c->d_ids), n_px * sizeof(int32_t)));
        c->W = width; c->H = height;
        c->have_frame = true;
    }
    HIP_TRY(c, hipMemsetAsync(c->d_accum, 0, n_px * sizeof(float4), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_packed, 0, n_px * sizeof(uint32_t), c->stream));
    HIP_TRY(c, hipMemsetAsync(c->d_ids, 0xff, n_px * sizeof(int32_t), c->stream));   // -1 = nothing recorded
    HIP_TRY(c, hipMemsetAsync(c->d_segments, 0, 8 * sizeof(unsigned long long), c->stream));
    return RT_OK;
}

int rt_render(rt_ctx *ctx, const rt_params *p) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!p) return fail(c, RT_E_BADARG, "rt_render: params is null");
    if (!c->have_tables || !c->have_scene || !c->have_cam || !c->have_frame)
        return fail(c, RT_E_STATE, "rt_render: tables, scene, camera and frame must be set first");
    if (p->width != c->W || p->height != c->H)
        return fail(c, RT_E_BADARG, "rt_render: width/height differ from rt_begin_frame");
    if (p->band_stride == 0 || p->band_begin >= p->band_stride)
        return fail(c, RT_E_BADARG, "rt_render: need band_begin < band_stride");
    if (p->max_bounces > RT_MAX_BOUNCES) return fail(c, RT_E_BADARG, "rt_render: max_bounces above RT_MAX_BOUNCES");
    if ((uint64_t)p->sample_begin + p->sample_count > p->spp)
        return fail(c, RT_E_BADARG, "rt_render: sample range exceeds spp");
    if (p->sampling_mode != RT_SAMPLING_DET) return fail(c, RT_E_BADARG, "rt_render: unknown sampling_mode");
    if (!(p->t_min == p->t_min)) return fail(c, RT_E_BADARG, "rt_render: t_min is NaN");
    const uint32_t n_bands = bands_of(p->height, p->band_begin, p->band_stride);
    if (n_bands == 0 || p->sample_count == 0) return RT_OK;
    if (n_bands > 65535u) return fail(c, RT_E_BADARG, "rt_render: too many bands for one launch");
    HIP_TRY(c, hipSetDevice(c->device));
    DevFrame fr{c->d_disc, c->d_sphere, c->d_screen, c->d_accum, c->d_ids, c->d_segments};
    HIP_TRY(c, rtk_launch_render(c->dsc, c->cam, *p, fr, n_bands, c->stream));
    return RT_OK;
}

int rt_pack(rt_ctx *ctx) {
    Ctx *c = C(ctx);
    if (!c) return RT_E_BADARG;
    if (!c->have_frame) return fail(c, RT_E_STATE, "rt_pack: no frame");
    HIP_TRY(c, hipSetDevice(c->device));
    HIP_TRY(c, rtk_launch_pack(c->d_accum, c->d_packed, c->W * c->H, c->stream));
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
    if ((rc = read_back(c, tmp.data(), c->d_accum, n_px * sizeof(float4))) != RT_OK) return rc;
    for (size_t k = 0; k < n_px; ++k) { rgb[3 * k] = tmp[k].x; rgb[3 * k + 1] = tmp[k].y; rgb[3 * k + 2] = tmp[k].z; }
    return RT_OK;
}

int rt_read_sample_counts(rt_ctx *ctx, uint32_t *counts, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, counts, n_px, "rt_read_sample_counts");
    if (rc != RT_OK) return rc;
    std::vector<float4> tmp;
    try { tmp.resize(n_px); } catch (...) { return fail(c, RT_E_OOM, "rt_read_sample_counts: out of host memory"); }
    if ((rc = read_back(c, tmp.data(), c->d_accum, n_px * sizeof(float4))) != RT_OK) return rc;
    for (size_t k = 0; k < n_px; ++k) counts[k] = (uint32_t)tmp[k].w;
    return RT_OK;
}

int rt_read_packed(rt_ctx *ctx, uint32_t *out, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, out, n_px, "rt_read_packed");
    if (rc != RT_OK) return rc;
    if ((rc = rt_pack(ctx)) != RT_OK) return rc;
    return read_back(c, out, c->d_packed, n_px * sizeof(uint32_t));
}

int rt_read_primary_ids(rt_ctx *ctx, int32_t *ids, size_t n_px) {
    Ctx *c = C(ctx);
    int rc = check_read(c, ids, n_px, "rt_read_primary_ids");
    if (rc != RT_OK) return rc;
    return read_back(c, ids, c->d_ids, n_px * sizeof(int32_t));
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
    if (!c->have_frame) return fail(c, RT_E_STATE, "rt_device_buffers: no frame");
    if (accum) *accum = c->d_accum;
    if (packed) *packed = c->d_packed;
    if (ids) *ids = c->d_ids;
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
    HIP_TRY(c, rtk_launch_pack_bands(c->d_accum, dst_device, c->W, c->H, band_begin, band_stride, n_bands,
                                     c->stream));
    return RT_OK;
}

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
