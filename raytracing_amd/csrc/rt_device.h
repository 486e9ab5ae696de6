// rt_device.h - structures shared by the kernels and the C-ABI layer.
// Internal to librt_hip.so; the public interface is include/rt_hip.h.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_hip.h"

// primitive kinds of a run; values match the order the C ABI lists the groups
enum { DEV_SPHERE = 0, DEV_PLANE = 1, DEV_CUBE = 2, DEV_TRI = 3 };

// spheres of one run are padded to a multiple of this many entries with
// spheres that can never be hit, so the scan can take them in fixed batches
#define DEV_SPHERE_BATCH 16u
// sphere runs at least this long go through the conservative rejection test
#define DEV_CULL_MIN_RUN 32u

// The primitive list in device memory.  The reference's insertion-ordered
// Vec<Box<dyn Hittable>> (src/geometry/mod.rs:228-238) is cut into maximal
// runs of one kind; a run's primitives lie contiguously in that kind's array,
// in list order.  All arrays are read with wave-uniform indices.
struct DevScene {
    const uint4 *runs;        // {kind, first slot, count, list id of the first}
    uint32_t n_runs;
    uint32_t need_ff_shared;  // some Cube has a Dielectric material: its front_face
                              // is inherited from the previously accepted hit
    const float4 *sph_geom;   // cx, cy, cz, radius
    const float4 *sph_cull;   // conservative-test records, two spheres per 8 floats:
                              // {cx0,cx1, cy0,cy1, cz0,cz1, thr0,thr1}
    const uint32_t *sph_mat;
    const float4 *planes;     // 2 per plane: {n, material bits}, {point, 0}
    const float4 *cubes;      // 4 per cube: {center, material bits}, {w/2,h/2,d/2,0}, {corner0,0}, {corner1,0}
    const float4 *tris;       // 3 per triangle: {v0, material bits}, {e1,0}, {e2,0}
    const float4 *mat_albedo; // r, g, b, param
    const int32_t *mat_kind;
};

// Per-frame buffers and the lookup tables
struct DevFrame {
    const float *tbl_disc;    // 128
    const float *tbl_sphere;  // 768
    const float *tbl_screen;  // 16384
    float4 *accum;            // {r,g,b,n} per pixel, display order
    int32_t *primary_ids;     // per pixel, display order
    unsigned long long *segments;
};

hipError_t rtk_launch_render(const DevScene &sc, const rt_camera &cam, const rt_params &p,
                             const DevFrame &fr, uint32_t n_bands, hipStream_t stream);
hipError_t rtk_launch_pack_bands(const float4 *accum, uint32_t *out, uint32_t W, uint32_t H,
                                 uint32_t band_begin, uint32_t band_stride, uint32_t n_bands,
                                 hipStream_t stream);
hipError_t rtk_launch_pack(const float4 *accum, uint32_t *out, uint32_t n_px, hipStream_t stream);

#endif
