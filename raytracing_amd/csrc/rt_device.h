// rt_device.h - structures shared by the kernels and the C-ABI layer.
// Internal to librt_hip.so; the public interface is include/rt_hip.h.
#ifndef RT_DEVICE_H
#define RT_DEVICE_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "rt_hip.h"

// primitive kinds of a run; values match the order the C ABI lists the groups
enum { DEV_SPHERE = 0, DEV_PLANE = 1, DEV_CUBE = 2, DEV_TRI = 3 };

// how a run is scanned
enum { RUN_EXACT = 0, RUN_FLAT = 1, RUN_TREE = 2 };
// sphere runs at least this long go through the conservative rejection test
#define DEV_CULL_MIN_RUN 32u
// sphere / triangle runs at least this long get the box tree
#define DEV_TREE_MIN_RUN 64u
#define DEV_MAX_LEVELS 8u

// One maximal run of same-kind primitives of the reference's insertion-ordered
// Vec<Box<dyn Hittable>> (src/geometry/mod.rs:228-238).  128 bytes, read with two
// scalar loads.  A RUN_FLAT run has its rejection records in DevScene::cull_rec (steps of
// eight records, 128 bytes).  A RUN_TREE run has an implicit 8-ary tree of boxes in
// DevScene::tree_rec: level 0 holds one box per primitive in steps of eight (DEV_NODE_F4 float4),
// level L+1 one bounding box per step of level L, again in steps of eight; the children of
// step s of level L+1 are steps 8s..8s+7 of level L.
struct DevRun {
    uint32_t kind;        // DEV_*
    uint32_t first;       // first slot in the kind's arrays
    uint32_t count;       // primitives in the run
    uint32_t id0;         // list position of the run's first primitive
    uint32_t mode;        // RUN_*
    uint32_t n_levels;    // RUN_FLAT: 1; RUN_TREE: levels of the tree
    uint32_t top_steps;   // steps of the top level (RUN_FLAT: all steps, an even number)
    uint32_t order_base;  // DevScene::order + order_base: list position in run -> slot offset; RUN_TREE: followed
                          // by the slot offsets of the n_unbounded primitives the tree does not hold
    uint32_t level_base[DEV_MAX_LEVELS];   // RUN_TREE: float4 offset of each level in tree_rec; RUN_FLAT: [0] = first step
    uint32_t level_aux[DEV_MAX_LEVELS];    // RUN_TREE: boxes at each level; RUN_FLAT: [0] = float bits of 1 - eps
    float reach;          // RUN_TREE, RUN_FLAT: at least |centre| + |radius| of every bounded ball (|x| of every box corner)
    float inv_2rho;       // RUN_TREE of spheres: 1 / (2 * smallest |radius|), +inf when that is 0 or for triangles
    uint32_t n_unbounded; // RUN_TREE: primitives with non-finite or huge coordinates, tested for every ray
    uint32_t pad[5];
};

// The primitive list in device memory.  All arrays are read with wave-uniform
// indices except where a lane fetches the primitive it finally hit.
struct DevScene {
    const DevRun *runs;
    uint32_t n_runs;
    uint32_t need_ff_shared;  // some Cube has a Dielectric material: its front_face
                              // is inherited from the previously accepted hit
    const float4 *sph_geom;   // cx, cy, cz, radius                  (by slot)
    const uint32_t *sph_mat;  //                                     (by slot)
    const uint32_t *sph_id;   // list position                       (by slot)
    const float4 *planes;     // 2 per plane: {n, material bits}, {point, 0}
    const float4 *cubes;      // 4 per cube: {center, material bits}, {w/2,h/2,d/2,0}, {corner0,0}, {corner1,0}
    const float4 *tris;       // 3 per triangle: {v0, material bits}, {e1,0}, {e2,0}
    const uint32_t *tri_id;   // list position                       (by slot)
    const float4 *cull_rec;   // RUN_FLAT rejection records, 8 float4 per step of eight balls:
                              // four pairs {cx0,cx1, cy0,cy1}{cz0,cz1, thr0,thr1}
    const float4 *tree_rec;   // RUN_TREE records, DEV_NODE_F4 float4 per step of eight boxes (72 floats):
                              // per pair of boxes {lo0,lo1,hi0,hi1,lo0,lo1} for x, y, z
    const uint32_t *order;    // see DevRun::order_base
    const float4 *mat_rec;    // 2 per material: {r, g, b, param}, {kind bits, 1/param, r0 front, r0 back}
                              // (Dielectric: ratio and reflectance at normal incidence for either face,
                              // src/geometry/mod.rs:313-317,327: functions of the material alone)
    // the top levels of ONE tree run are copied into LDS by every workgroup:
    // levels >= lds_level of run lds_run, tree_rec float4s [lds_first, +lds_count)
    uint32_t lds_run, lds_level, lds_first, lds_count;
    // the same for a launch with the deep material stack (less room): the launch copies them over the four above
    uint32_t lds_deep_level, lds_deep_first, lds_deep_count;
    uint32_t lds_kind;        // DEV_SPHERE or DEV_TRI: what run lds_run, the scene's largest tree, holds
    uint32_t modes;           // bit 0: some run is RUN_FLAT, bit 1: some run is RUN_TREE
    uint32_t cheap_ok;        // every sphere has finite |centre| <= 2^30 and 2^-30 <= |radius| <= 2^30: rays of
                              // moderate origin and length may take the short division and root sequences
};
// Samples per pixel one launch renders at most: what a ring slot holds for a work unit of one pixel.
// A launch has a fixed cost of 0.5 to 0.8 ms (its waves end unevenly), so a frame takes as few launches
// as it can: 256 samples of a 4K frame in one launch (units of two pixels) against four of 64 samples
// (units of eight): 10,000 spheres 387.8 against 395.8 ms, a million triangles 177.7 against 185.9.
#ifndef DEV_SAMPLE_CHUNK
#define DEV_SAMPLE_CHUNK 512u
#endif
// Staging ring: every wavefront of a render launch keeps the colours of its unfinished work
// units in a part of its own, DEV_STAGE_SLOTS units of at most 2^DEV_STAGE_ITEM_BITS (pixel,
// sample) items, 12 bytes each.  The ring's size depends on the launch's grid, not on the frame.
#define DEV_STAGE_ITEM_BITS 9u
#define DEV_STAGE_SLOTS 4u
struct StagedRgb { float r, g, b; };
#define DEV_STAGE_WAVE_BYTES ((size_t)sizeof(StagedRgb) * DEV_STAGE_SLOTS << DEV_STAGE_ITEM_BITS)   // 24 KiB
// float4s per step of eight boxes in tree_rec.  18 (DEV_NODE_MIRROR, the default): per pair of boxes
// and axis {lo0,lo1,hi0,hi1,lo0,lo1}, of which a lane reads the first or the last four by the sign
// of its ray's direction: near planes first, far planes second, without a comparison.
// 12 (DEV_NODE_MIRROR=0, for comparison): {lo0,lo1,hi0,hi1}, near and far sorted out by the test.
#ifndef DEV_NODE_MIRROR
#define DEV_NODE_MIRROR 1
#endif
#define DEV_NODE_F4 (DEV_NODE_MIRROR ? 18u : 12u)
// The render kernel of a scene with a tree exists for two depth budgets: up to DEV_STACK_SHALLOW
// bounces (the reference's 10, src/raytracer.rs:132) and up to RT_MAX_BOUNCES.  The material stack
// in LDS is sized by it, and what it leaves of a workgroup's 80 KB goes to the copy of a tree's
// top levels: DEV_TREE_LDS_F4 / DEV_TREE_LDS_F4_DEEP float4s.
#define DEV_STACK_SHALLOW 10u
#ifndef DEV_TREE_LDS_F4
#define DEV_TREE_LDS_F4 (DEV_NODE_MIRROR ? 3528u : 2352u)        // 56 KB
#endif
#ifndef DEV_TREE_LDS_F4_DEEP
#define DEV_TREE_LDS_F4_DEEP (DEV_NODE_MIRROR ? 2754u : 2748u)   // 44 KB
#endif

// Per-frame buffers and the lookup tables
struct DevFrame {
    const float *tbl_disc;    // 128
    const float *tbl_sphere;  // 768
    const float *tbl_screen;  // 16384
    float4 *accum;            // {r,g,b,n} per pixel, display order
    int32_t *primary_ids;     // per pixel, display order
    StagedRgb *stage;         // staging ring, DEV_STAGE_WAVE_BYTES per wavefront of the grid
    unsigned long long *segments;
    uint32_t *queue;          // work units of the running launch handed out so far (zero when it starts)
    uint32_t *queue_next;     // the counter the next launch on the stream will use: the running one zeroes it
    uint32_t disc_plain;      // every entry of tbl_disc has 2^-90 <= |value| < 1: pixel column + entry is zero or
                              // at least 2^-24, which lets the division by width - 1 take the short sequence
};

// stage_waves: wavefronts the staging ring behind fr.stage has room for (the grid is cut to it);
// unit_px_log2: pixels of a tile per work unit, or -1 to choose by launch size
// cams (device memory, may be null), n_frames: one launch over n_frames frames of p.width x p.height, frame
// f through cams[f] into frame f of fr.accum and fr.primary_ids (width*height elements apart); only scenes
// for which rtk_has_frames_kernel(modes) holds.  With cams == null: one frame through `cam`.
hipError_t rtk_launch_render(const DevScene &sc, const rt_camera &cam, const rt_params &p,
                             const DevFrame &fr, uint32_t n_bands, uint32_t n_cus, uint32_t stage_waves,
                             int unit_px_log2, hipStream_t stream, const rt_camera *cams = nullptr,
                             uint32_t n_frames = 1u);
bool rtk_has_frames_kernel(uint32_t modes);
// The per-GPU static scratch of rt_render_static (see rt_kernels.hip): device addresses of its
// parts on `device`, which must be the current device the first time it is asked for.
struct RtkStaticScratch {
    uint32_t *queue;                // two counters; head_bytes from `queue` on are cleared before a launch
    unsigned long long *segments;   // sink for the segment count when the caller wants none
    StagedRgb *stage;
    uint32_t stage_waves;
    uint32_t head_bytes;
};
hipError_t rtk_static_scratch(int device, RtkStaticScratch *out);
// wavefronts a render launch of a scene with these scan modes (DevScene::modes) puts on a device with
// n_cus compute units at most
uint32_t rtk_resident_waves(uint32_t n_cus, uint32_t modes);
// the instantiation render_tiles<scan_modes, depth_budget> a launch with this depth budget runs
void rtk_kernel_variant(uint32_t modes, uint32_t max_bounces, uint32_t *scan_modes, uint32_t *depth_budget);
hipError_t rtk_launch_pack_bands(const float4 *accum, uint32_t *out, uint32_t W, uint32_t H,
                                 uint32_t band_begin, uint32_t band_stride, uint32_t n_bands,
                                 hipStream_t stream);
hipError_t rtk_launch_pack(const float4 *accum, uint32_t *out, uint32_t n_px, hipStream_t stream);
// blocks x 256 threads x rounds operand triples through rtk::check_arith; bad[0..1] must be zeroed
hipError_t rtk_launch_check_arith(uint32_t blocks, uint32_t rounds, uint32_t seed, unsigned long long *bad,
                                  hipStream_t stream);
hipError_t rtk_launch_compose_bands(const uint32_t *blocks, uint32_t *out, uint32_t W, uint32_t H, uint32_t n_ctx,
                                    size_t block_px, hipStream_t stream);

#endif
