/*
 * rt_hip.h - C ABI of the MI355X (gfx950) render-loop library, librt_hip.so.
 *
 * This is the boundary a host program (the reference's Rust crate through an
 * `extern "C"` block, the bundled C++ headless driver, or Python through
 * ctypes) binds to.  Plain pointers and sizes only; no C++ or torch types.
 *
 * The reference (ojferro/raytracing) has no FFI of its own.  The seam this
 * header replaces is the worker call
 *     calculate_some_pxls(...)            src/raytracer.rs:164-168, 232-271
 * plus the fan-in that quantises each pixel
 *     write_colour(...)                   src/raytracer.rs:46-55, 176-185
 * Every entry point cites the reference code whose role it takes.
 *
 * Conventions
 *   - every function returns RT_OK (0) or a negative RT_E_* code; no panic or
 *     exception crosses the boundary (the reference `.unwrap()`s instead:
 *     src/raytracer.rs:126,204,268).  rt_last_error() gives the text.
 *   - a context belongs to one GPU and is not thread-safe; different contexts
 *     may be driven from different host threads.
 *   - all arithmetic is IEEE binary32 without fused contraction, as in the
 *     reference (src/vector/mod.rs:141-189).
 *   - image rows: `j` counts from the BOTTOM of the picture (src/raytracer.rs:245,
 *     256); buffers handed back are in display order, row = height-1-j
 *     (src/raytracer.rs:265).
 */
#ifndef RT_HIP_H
#define RT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rt_ctx rt_ctx;

enum {
    RT_OK = 0,
    RT_E_BADARG = -1, /* null pointer, size out of range, malformed scene */
    RT_E_HIP = -2,    /* a HIP runtime call failed (text in rt_last_error) */
    RT_E_OOM = -3,    /* host or device allocation failed */
    RT_E_STATE = -4,  /* call order: tables/scene/camera/frame not set */
    RT_E_IO = -5      /* file could not be written */
};

/* Material kinds: Metal / Lambertian / Dielectric, src/geometry/mod.rs:268-346 */
enum { RT_METAL = 0, RT_LAMBERTIAN = 1, RT_DIELECTRIC = 2 };

/* Sampling schedules.
 * RT_SAMPLING_DET: every table index is a pure function of (pixel, sample,
 *   bounce, seed).  The two disc draws of sample n = (j*W+i)*spp+s use table
 *   entries (2n+1)%64 and (2n+2)%64, which is what the reference's global
 *   counter (src/denoising/mod.rs:12-18) yields for one worker thread on its
 *   first frame; primary rays are therefore those of the reference.  The
 *   unit-sphere counter (src/denoising/mod.rs:37-41) starts at a per-path hash
 *   and advances once per closest-hit Metal/Lambertian scatter; the Dielectric
 *   coin (src/geometry/mod.rs:319) is a hash of (path, bounce).
 * It is the only schedule the library renders: the reference's own order of
 * the secondary draws depends on how its worker threads interleave and on an
 * OS-seeded generator, so no schedule can reproduce it; any other value of
 * rt_params.sampling_mode is RT_E_BADARG. */
enum { RT_SAMPLING_DET = 0 };

#define RT_BAND_ROWS 8u      /* rows per band = tile height of one wavefront */
#define RT_MAX_FRAMES 4096u  /* frames one rt_begin_frames may hold */
#define RT_MAX_BOUNCES 16u   /* the reference uses 10 (src/raytracer.rs:132) */

/*
 * Scene as structure-of-arrays, one group per primitive kind
 * (HittableList, src/geometry/mod.rs:228-238; Scene::get_scene,
 * src/scene/mod.rs:15-57).  `*_id[k]` is the position of that primitive in
 * the reference's insertion-ordered list; the ids of all groups together must
 * be a permutation of 0..n-1.  List order decides ties and is the id that
 * rt_read_primary_ids() reports.  All arrays are copied by rt_upload_scene().
 */
typedef struct rt_scene {
    /* Sphere{center, radius}: src/geometry/mod.rs:47-57 (radius may be < 0) */
    uint32_t n_spheres;
    const float *sph_cx, *sph_cy, *sph_cz, *sph_r;
    const uint32_t *sph_mat, *sph_id;
    /* Plane{normal, point}: src/geometry/mod.rs:97-109 */
    uint32_t n_planes;
    const float *pl_nx, *pl_ny, *pl_nz, *pl_px, *pl_py, *pl_pz;
    const uint32_t *pl_mat, *pl_id;
    /* Cube{center, w, h, d}: src/geometry/mod.rs:134-155 (corners derived) */
    uint32_t n_cubes;
    const float *cb_cx, *cb_cy, *cb_cz, *cb_w, *cb_h, *cb_d;
    const uint32_t *cb_mat, *cb_id;
    /* Triangle{v0, v1, v2}: no counterpart in the reference sources (its README
     * advertises .obj meshes, the code has none).  Semantics defined here:
     * two-sided Moller-Trumbore with e1 = v1-v0, e2 = v2-v0, plain f32
     * mul/add, miss when det == 0, u < 0, u > 1, v < 0, u+v > 1,
     * t < t_min or t_max < t.  A hit then counts only if the hit point
     * p = o + t*d lies in the bounding box of v0, v0+e1, v0+e2 widened on
     * every side by 2^-7 of its largest extent plus 2^-20 of the largest
     * |coordinate| among the vertices and o: evaluated in f32, the
     * barycentric test alone accepts points far outside a sliver or a
     * triangle seen edge-on, and may yield NaN (NaN fails the box test).
     * Outward normal unit(cross(e1, e2)), then set_face_normal and the material
     * scatter like a Sphere. */
    uint32_t n_tris;
    const float *tri_v0x, *tri_v0y, *tri_v0z;
    const float *tri_v1x, *tri_v1y, *tri_v1z;
    const float *tri_v2x, *tri_v2y, *tri_v2z;
    const uint32_t *tri_mat, *tri_id;
    /* materials: kind, albedo, param (Metal: fuzz; Dielectric: index of
     * refraction; Lambertian: unused) - src/geometry/mod.rs:268-304 */
    uint32_t n_materials;
    const int32_t *mat_kind;
    const float *mat_r, *mat_g, *mat_b, *mat_param;
} rt_scene;

/* The fields of Camera that get_ray reads, src/camera/mod.rs:9-28,75-79. */
typedef struct rt_camera {
    float origin[3];
    float lower_left_corner[3];
    float horizontal[3];
    float vertical[3];
    float u[3];
    float v[3];
    float lens_radius;
} rt_camera;

/* Arguments of calculate_some_pxls, src/raytracer.rs:232-242, plus the
 * partitioning the reference does with thread_id/num_threads (:245). */
typedef struct rt_params {
    uint32_t width, height;
    uint32_t spp;           /* samples per pixel of the whole frame */
    uint32_t sample_begin;  /* this call renders samples [begin, begin+count) */
    uint32_t sample_count;
    uint32_t max_bounces;   /* ray_colour depth budget, 10 in the reference */
    uint32_t gamma;         /* non-zero: sqrt on the sky radiance (:86-90) */
    float t_min;            /* 0.001 (:75) */
    uint32_t band_begin;    /* render 8-row bands b with b % band_stride == band_begin */
    uint32_t band_stride;   /* 1 = whole frame; G = one of G GPUs */
    uint32_t sampling_mode; /* RT_SAMPLING_DET */
    uint32_t seed;
} rt_params;

/* ---- context: takes the place of thread spawn/join, src/raytracer.rs:146-171,203-205 */
int rt_create(int device, rt_ctx **out);
void rt_destroy(rt_ctx *ctx);
const char *rt_last_error(const rt_ctx *ctx); /* ctx may be NULL: last create error */

/* ---- inputs */
/* tables of src/denoising/mod.rs:55 (64x2), :59 (256x3), :61 (128*128, idx = 128*i+j) */
int rt_upload_tables(rt_ctx *ctx, const float *disc, const float *sphere, const float *screen);
int rt_upload_scene(rt_ctx *ctx, const rt_scene *scene);
int rt_set_camera(rt_ctx *ctx, const rt_camera *cam);

/* ---- frame */
/* (re)allocate the per-pixel buffers for width x height and zero the sums:
 * `let mut px_colour = colour::new(0,0,0)`, src/raytracer.rs:249 */
int rt_begin_frame(rt_ctx *ctx, uint32_t width, uint32_t height);
/* The reference renders NUM_FRAMES_TO_RENDER = 10 frames in a row and moves the camera between
 * them (src/raytracer.rs:32,165-168,190; Scene::move_cam, src/scene/mod.rs:78-83).
 * rt_begin_frames is rt_begin_frame for n_frames frames of one size at once: the per-pixel buffers
 * hold them one after the other and all are zeroed; rt_begin_frame(w, h) = rt_begin_frames(w, h, 1).
 * rt_select_frame names the frame that rt_render, rt_pack, the rt_read_* calls,
 * rt_device_buffers, rt_export_packed_bands and rt_gather_packed work on (0 after a begin). */
int rt_begin_frames(rt_ctx *ctx, uint32_t width, uint32_t height, uint32_t n_frames);
int rt_select_frame(rt_ctx *ctx, uint32_t frame);
/* rt_render for frames 0..n_frames-1 at once, frame f seen through cameras[f] (host memory, copied
 * before the call returns; the camera of rt_set_camera is not used): the same sums, ids and
 * segment count as n_frames calls of rt_set_camera + rt_select_frame + rt_render.  For a scene
 * without long runs (fewer than 32 spheres and 64 triangles in a row, the reference's own scene)
 * the frames share ONE launch: a frame as small as the reference's 600 x 337 at one sample per pixel
 * does not fill the GPU, and ten of them in a launch take a fifth of the time of ten launches.
 * Other scenes are rendered frame after frame.  Asynchronous on the context's stream. */
int rt_render_frames(rt_ctx *ctx, const rt_params *params, const rt_camera *cameras, uint32_t n_frames);
/* enqueue the render loop for the bands and samples named in params; adds into
 * the per-pixel sums (`px_colour += ray_colour(...)`, src/raytracer.rs:262).
 * Asynchronous on the context's stream. */
int rt_render(rt_ctx *ctx, const rt_params *params);
/* rt_render for callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in
 * the reference.  It needs no rt_begin_frame: the caller owns the outputs, in device
 * memory, and the library's own working memory (the staging ring of unfinished work
 * units and the queue counters) is a fixed array inside the library's code object,
 * RT_STATIC_SCRATCH_BYTES per GPU, allocated when the code object is loaded on a GPU (the
 * first HIP call of a process that has loaded the library, whether or not this entry point
 * is ever used) and held until the process ends.  The call makes no hipMalloc/hipFree; it
 * is enqueued on the context's stream (rt_device_buffers hands it out).
 *   accum_device    float4 {r,g,b,n} per pixel, display order, width*height of them;
 *                   the call ADDS the samples it renders (zero it before sample 0)
 *   ids_device      int32 per pixel, display order; written for sample 0 only
 *   segments_device uint64 the segment count is added to, or NULL
 * Limits: 2..65536 pixels per side and at most RT_STATIC_MAX_PIXELS pixels, anything
 * larger is RT_E_BADARG; the ring holds RT_STATIC_MAX_WAVES wavefronts (a 256-CU GPU
 * filled to 16 per CU, which is all a scene with a box tree uses; rt_render gives a
 * scene without one twice as many), a launch is cut to that many.  Every call
 * on one GPU uses that GPU's one array: calls on the same GPU that overlap in time
 * (two contexts, two host threads) race; calls on one context are ordered by its
 * stream.  Asynchronous like rt_render; sums are bit-identical to rt_render's. */
#define RT_STATIC_MAX_WAVES 4096u
#define RT_STATIC_SCRATCH_BYTES (256u + RT_STATIC_MAX_WAVES * 24576u) /* 96 MiB + 256 B */
#define RT_STATIC_MAX_PIXELS 0xffffffffu
int rt_render_static(rt_ctx *ctx, const rt_params *params, void *accum_device,
                     int32_t *ids_device, uint64_t *segments_device);
/* enqueue the quantiser write_colour (src/raytracer.rs:46-55) over the whole frame */
int rt_pack(rt_ctx *ctx);
int rt_sync(rt_ctx *ctx);

/* ---- outputs (each synchronises the stream; rows in display order) */
/* PxData.c, src/raytracer.rs:225-230: un-normalised RGB sums, 3 floats per pixel */
int rt_read_accum(rt_ctx *ctx, float *rgb, size_t n_px);
/* PxData.num_samples: samples added so far, per pixel */
int rt_read_sample_counts(rt_ctx *ctx, uint32_t *counts, size_t n_px);
/* 0x00RRGGBB as written by write_colour; runs rt_pack first */
int rt_read_packed(rt_ctx *ctx, uint32_t *out, size_t n_px);
/* list index of the primitive the first segment of sample 0 hit, -1 = sky
 * (first HittableList::hit call, src/raytracer.rs:75).  No reference output. */
int rt_read_primary_ids(rt_ctx *ctx, int32_t *ids, size_t n_px);
/* path segments (= HittableList::hit calls) traced since rt_begin_frame */
int rt_ray_count(rt_ctx *ctx, uint64_t *segments);

/* ---- device-side access for callers that keep the image on the GPU
 * (multi-GPU gather).  Pointers stay valid until the next rt_begin_frame or
 * rt_destroy.  accum: float4 {r,g,b,n} per pixel; packed: uint32 per pixel;
 * ids: int32 per pixel; all in display order.  stream: the hipStream_t every
 * launch of this context is enqueued on.  Before the first rt_begin_frame the
 * three buffers come back NULL and only the stream is valid: a caller of
 * rt_render_static, which needs no frame, orders its own work (clearing its
 * accum buffer, a graph capture) against that stream. */
int rt_device_buffers(rt_ctx *ctx, void **accum, void **packed, void **ids, void **stream);
/* quantise (write_colour, src/raytracer.rs:46-55) the bands b with
 * b % band_stride == band_begin straight into dst_device (device memory of the
 * caller, capacity_px pixels) as one compact block: the k-th such band occupies
 * rows 8k..8k+7 of the block, the row with the smallest j (lowest on screen)
 * first, width pixels per row; rows above the top of the image are zero.  This
 * is what one GPU contributes to a multi-GPU gather, like one worker's share of
 * the channel fan-in, src/raytracer.rs:176-185.  Asynchronous on the context's
 * stream. */
int rt_export_packed_bands(rt_ctx *ctx, uint32_t *dst_device, size_t capacity_px,
                           uint32_t band_begin, uint32_t band_stride);
/* The fan-in of a multi-GPU frame below any host framework, the receive loop of
 * src/raytracer.rs:176-185: contexts ctxs[0..n_ctx) (one per GPU, or several on one
 * GPU) have rendered the same frame size with band_begin = g, band_stride = n_ctx.
 * Each quantises its own bands into a compact block on its own GPU, the blocks travel
 * to the GPU of ctxs[0] (one peer-to-peer copy per context, each over its own xGMI
 * link), are put together there in display order and copied to `out` (host memory,
 * width*height pixels, 0x00RRGGBB).  Synchronises every context's stream work issued
 * before the call.  Errors are reported on ctxs[0] (the text of a failure on another
 * context is copied there). */
int rt_gather_packed(rt_ctx *const *ctxs, uint32_t n_ctx, uint32_t *out, size_t n_px);
/* Self-check of the library's arithmetic, no counterpart in the reference: the render
 * kernel replaces some `/` and sqrt (Rust f32 semantics, src/vector/mod.rs:141-189) by
 * shorter instruction sequences that must give the same bits for operands of moderate
 * size.  Runs both on about `operands` random operand sets of that range on the GPU and
 * returns the number that differ in any bit: mismatches[0] quotients, [1] roots. */
int rt_check_arith(rt_ctx *ctx, uint64_t operands, uint32_t seed, uint64_t mismatches[2]);
/* Which instantiation of the render kernel a launch with this depth budget runs for the
 * uploaded scene, for profiles and bench records (no counterpart in the reference):
 * scan_modes bit 0 = some sphere run goes through the list-order rejection scan, bit 1 = some
 * run is walked through a box tree; depth_budget = the largest max_bounces that kernel takes. */
int rt_kernel_info(rt_ctx *ctx, uint32_t max_bounces, uint32_t *scan_modes, uint32_t *depth_budget);
/* First 16 hex digits of the SHA-256 of the sources the library was built from that decide what a
 * launch executes (kernels, scene layout and tree builder, launch code). */
const char *rt_build_id(void);
/* HIP-event stopwatch on the context's stream; stop synchronises */
int rt_timer_start(rt_ctx *ctx);
int rt_timer_stop(rt_ctx *ctx, float *milliseconds);

/* ---- host-side helpers (no GPU needed) */
/* Camera::new, src/camera/mod.rs:35-73 */
int rt_camera_new(float v_fov_deg, float aspect_ratio, float aperture, float focus_dist,
                  const float origin[3], const float look_at[3], const float v_up[3],
                  rt_camera *out);
/* Scene::get_camera with the dolly counter made an argument, src/scene/mod.rs:59-76,
 * and the aperture the reference keeps commented out (:70) */
int rt_scene_camera(uint32_t counter, float aperture, rt_camera *out);
/* `(image_width as f32 / aspect_ratio) as usize`, src/raytracer.rs:118-120 */
uint32_t rt_image_height(uint32_t width, float aspect_ratio);
/* decimal text -> binary32, round-to-nearest per token like a Rust literal;
 * returns the number of values found (may exceed cap; only cap are stored) */
size_t rt_parse_floats(const char *text, size_t len, float *out, size_t cap);
/* PPM::write_file, src/image/mod.rs:61-67: "P6 {w} {h} 255\n" + RGB bytes */
int rt_write_ppm(const char *path, const uint32_t *packed, uint32_t width, uint32_t height);
/* the same pixels as an 8-bit RGB PNG (uncompressed), the format of the reference's
 * output_images/ - no counterpart in its code, which only blits to a window */
int rt_write_png(const char *path, const uint32_t *packed, uint32_t width, uint32_t height);

#ifdef __cplusplus
}
#endif
#endif /* RT_HIP_H */
