// rt_kernels.hip - the render loop as gfx950 (CDNA4, wave64) kernels.
//
// The launch fills the chip once with resident workgroups; every wavefront takes work units
// (a few rows of an 8x8 pixel tile x the samples of the launch) from one global queue.  Its
// work items are (pixel, sample) pairs: a lane whose path has ended (sky or depth budget) takes
// the next item, whatever pixel that belongs to - once a third of the wave stands idle, so that
// the steps in between run for many lanes at a time - and the wave starts its next unit while the
// longer paths of the last one are still on the way, so waves stay busy until the queue is
// empty; the per-sample colours are staged in memory and added to each pixel's sum in sample
// order when the pixel's unit is complete.
// The closest-hit scan follows the primitive list in the reference's insertion order, run by
// run (maximal stretches of one primitive kind).  Short runs: a wave-uniform loop, primitive
// data read through scalar loads, every lane tests the same primitive against its own ray - no
// divergence between lanes that are tracing.  Long runs of spheres or triangles: each lane
// walks an implicit 8-ary tree of bounding boxes for its own ray and runs the exact test only
// on what it cannot reject; equal hit distances are resolved by list position, so the result
// is that of the list-order scan.
// Arithmetic that decides a result is plain f32 mul/add/div/sqrt without contraction (the
// reference is scalar Rust f32, src/vector/mod.rs:141-189); fused multiply-add is used only in
// the conservative rejection tests (balls, boxes), which can never change a result.
//
// Reference lines each device function follows are cited at the function.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <mutex>

#include "rt_device.h"

// staged colours fetched at a time when a unit is added up: 16 where the tree walk sets the kernel's
// vector register count anyway, 8 elsewhere (with 16 this rare step set it: 96 registers instead of 64)
__host__ __device__ constexpr uint32_t sum_batch_of(int modes) { return (modes & 2) ? 16u : 8u; }
static_assert(DEV_STAGE_SLOTS == 4u, "UnitRing and the slot arithmetic assume four units in flight per wave");
static_assert(DEV_SAMPLE_CHUNK <= (1u << DEV_STAGE_ITEM_BITS), "the samples of a launch, for one pixel, must fit a ring slot");
#ifndef RTK_TREE_WAVES
#define RTK_TREE_WAVES 8
#endif

namespace rtk {

// ------------------------------------------------------------------ vec3
// src/vector/mod.rs:12-16 and the operator impls :107-189
struct V3 { float x, y, z; };
__device__ __forceinline__ V3 mk(float x, float y, float z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
__device__ __forceinline__ V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
__device__ __forceinline__ V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
__device__ __forceinline__ V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
__device__ __forceinline__ V3 operator*(V3 a, float t) { return mk(a.x * t, a.y * t, a.z * t); }
__device__ __forceinline__ V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
__device__ __forceinline__ V3 operator/(V3 a, float t) { return mk(a.x / t, a.y / t, a.z / t); }
// src/vector/mod.rs:34-36: (x*x + y*y) + z*z, three roundings for the products
__device__ __forceinline__ float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ float len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }
// src/vector/mod.rs:38-42
__device__ __forceinline__ V3 cross(V3 a, V3 b) {
    return mk(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
// src/vector/mod.rs:44-46: true division by the rounded length
__device__ __forceinline__ V3 unit(V3 a) { return a / __builtin_sqrtf(len2(a)); }
// src/vector/mod.rs:91-93
__device__ __forceinline__ V3 reflect(V3 v, V3 n) { return v - (n * 2.0f) * dot(v, n); }
// Rust f32::min (a NaN operand gives the other one): v_min_f32 in the IEEE mode kernels run in.
// Its callers pass a constant as b, so the order of -0 and +0, which neither defines, never matters.
__device__ __forceinline__ float rmin(float a, float b) { return __builtin_fminf(a, b); }
// Rust `as i32` and back: truncation towards zero, NaN -> 0, values beyond the range saturate
// ((i32::MAX as f32) rounds to 2^31).  v_cvt_i32_f32 is defined to do exactly that; written as the
// instruction because a C++ cast of such a value has no defined result.
__device__ __forceinline__ float trunc_i32_as_f32(float x) {
    int32_t i;
    asm("v_cvt_i32_f32 %0, %1" : "=v"(i) : "v"(x));
    return (float)i;
}

// ------------------------------------------------------------------ quotients and roots
// `x / y` and sqrtf(x) have to be the correctly rounded binary32 results (Rust's `/` and
// f32::sqrt).  The compiler's expansions of them take 11 and 16 instructions; a good part of
// that scales operands near the ends of the exponent range and patches up zeros, infinities
// and NaNs afterwards.  Away from those ends the steps in between, on their own, yield the
// same bits, and quotients by one denominator can share the reciprocal:
//     r = rcp(y);  r += r (1 - y r)                                  rcp_refined
//     q = n r;     q += r (n - y q);   q += r (n - y q)              div_refined
// for 2^-30 <= |y| <= 2^30 and 2^-90 <= |n| <= 2^60: no operand would have been scaled, the
// quotient is normal, and the residuals n - y q (multiples of 2^-47 |n| at least) are exact;
//     s = sqrt~(x), then one ulp down if x - s_down s <= 0, one up if x - s_up s > 0      sqrt_refined
// for 2^-96 <= x (+inf and NaN come out as themselves).  Callers check the ranges, per wave,
// and fall back to `/` and sqrtf; rtk::check_arith compares both on random operands.
__device__ __forceinline__ float rcp_refined(float y) {
    const float r = __builtin_amdgcn_rcpf(y);
    return __builtin_fmaf(__builtin_fmaf(-y, r, 1.0f), r, r);
}
__device__ __forceinline__ float div_refined(float n, float y, float r) {
    float q = n * r;
    q = __builtin_fmaf(__builtin_fmaf(-y, q, n), r, q);
    return __builtin_fmaf(__builtin_fmaf(-y, q, n), r, q);
}
__device__ __forceinline__ float sqrt_refined(float x) {
    const float s = __builtin_amdgcn_sqrtf(x);
    const float down = __uint_as_float(__float_as_uint(s) - 1u), up = __uint_as_float(__float_as_uint(s) + 1u);
    const float e_down = __builtin_fmaf(-down, s, x), e_up = __builtin_fmaf(-up, s, x);
    float t = e_down <= 0.0f ? down : s;
    t = e_up > 0.0f ? up : t;
    return t;
}
#ifndef RTK_SHORT_ARITH
#define RTK_SHORT_ARITH 1     // 0: `/` and sqrtf everywhere (for comparison)
#endif
constexpr float DIV_NUM_MIN = 0x1p-90f;    // smallest |numerator| of div_refined
constexpr float SQRT_ARG_MIN = 0x1p-96f;   // smallest argument of sqrt_refined
// three quotients by one denominator; `cheap` (wave-uniform) says that y is in div_refined's
// range and no numerator above it; numerators below it (zero included) send the wave to `/`
__device__ __forceinline__ V3 div3(V3 n, float y, bool cheap) {
    const float small = __builtin_fminf(__builtin_fminf(__builtin_fabsf(n.x), __builtin_fabsf(n.y)), __builtin_fabsf(n.z));
    if (cheap && __builtin_amdgcn_ballot_w64(!(small >= DIV_NUM_MIN)) == 0ull) {
        const float r = rcp_refined(y);
        return mk(div_refined(n.x, y, r), div_refined(n.y, y, r), div_refined(n.z, y, r));
    }
    return n / y;
}
// sqrtf for arguments that are rarely below sqrt_refined's range
__device__ __forceinline__ float sqrt_mostly(float x) {
    if (RTK_SHORT_ARITH && __builtin_amdgcn_ballot_w64(!(x >= SQRT_ARG_MIN)) == 0ull) return sqrt_refined(x);
    return __builtin_sqrtf(x);
}

// integer mixer of the sampling schedule (two multiply-xorshift rounds)
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du;
    x ^= x >> 15; x *= 0x846ca68bu;
    x ^= x >> 16;
    return x;
}
constexpr uint32_t GOLDEN32 = 0x9E3779B9u;

// Scene arrays are read through the constant address space: together with a
// wave-uniform index this makes the compiler fetch them with scalar loads
// (s_load_dwordx4/x16 into SGPRs) instead of 64 identical vector loads.  The
// arrays are never written while a render kernel runs.
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16 __attribute__((ext_vector_type(16)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef f4 f4a8 __attribute__((aligned(8)));   // four floats at an address that is a multiple of eight
typedef uint32_t u16v __attribute__((ext_vector_type(16)));
typedef uint32_t u8v __attribute__((ext_vector_type(8)));
typedef uint32_t u4v __attribute__((ext_vector_type(4)));
#define RT_CONST_AS __attribute__((address_space(4)))
template <typename T, typename S>
__device__ __forceinline__ const RT_CONST_AS T *const_ptr(const S *p) {
    return (const RT_CONST_AS T *)(uintptr_t)p;
}
// The scene's arrays.  Only the place of the DevScene in the kernel's argument block is kept in
// registers: a part of the scan fetches the address of an array where it starts to use it (a
// scalar load that hits the scalar cache), through a pointer the compiler cannot trace back, so
// that a dozen addresses do not sit in scalar registers through the whole scan - with the loop's
// own wave-uniform state they do not fit, and what does not fit is parked in lanes of vector
// registers at a vector instruction per use.
struct SceneView {
    const RT_CONST_AS DevScene *d;
    uint32_t n_runs;
    bool lazy;                   // false (the kernel for scenes without long runs, where they do fit): addresses are
                                 // fetched wherever the compiler likes, once
    unsigned long long *stats;   // fr.segments; slots 1..7 are instrumentation (RTK_STATS builds)
    __device__ __forceinline__ const RT_CONST_AS DevScene *at() const {
        const RT_CONST_AS DevScene *p = d;
        if (lazy) asm volatile("" : "+s"(p));
        return p;
    }
    __device__ __forceinline__ const RT_CONST_AS u8v *runs() const { return const_ptr<u8v>(at()->runs); }   // four per DevRun
    __device__ __forceinline__ const RT_CONST_AS f4 *sph_geom() const { return const_ptr<f4>(at()->sph_geom); }
    __device__ __forceinline__ const RT_CONST_AS f4 *planes() const { return const_ptr<f4>(at()->planes); }
    __device__ __forceinline__ const RT_CONST_AS f4 *cubes() const { return const_ptr<f4>(at()->cubes); }
    __device__ __forceinline__ const RT_CONST_AS f4 *tris() const { return const_ptr<f4>(at()->tris); }
    __device__ __forceinline__ const RT_CONST_AS f4 *mat_rec() const { return const_ptr<f4>(at()->mat_rec); }
    __device__ __forceinline__ const RT_CONST_AS f16 *cull_rec() const { return const_ptr<f16>(at()->cull_rec); }   // RUN_FLAT: two per step of eight records
    __device__ __forceinline__ const RT_CONST_AS f4 *tree_rec() const { return const_ptr<f4>(at()->tree_rec); }     // RUN_TREE: DEV_NODE_F4 per step of eight boxes
    __device__ __forceinline__ const RT_CONST_AS uint32_t *sph_mat() const { return const_ptr<uint32_t>(at()->sph_mat); }
    __device__ __forceinline__ const RT_CONST_AS uint32_t *sph_id() const { return const_ptr<uint32_t>(at()->sph_id); }
    __device__ __forceinline__ const RT_CONST_AS uint32_t *tri_id() const { return const_ptr<uint32_t>(at()->tri_id); }
    __device__ __forceinline__ const RT_CONST_AS uint32_t *order() const { return const_ptr<uint32_t>(at()->order); }
    __device__ __forceinline__ uint32_t need_ff_shared() const { return at()->need_ff_shared; }
    // lds_run, lds_level, lds_first, lds_count
    __device__ __forceinline__ u4v lds() const { return *(const RT_CONST_AS u4v *)&at()->lds_run; }
};
__device__ __forceinline__ SceneView view_of(const RT_CONST_AS DevScene *d, unsigned long long *stats, bool lazy) {
    SceneView v;
    v.d = d;
    v.lazy = lazy;
    v.n_runs = d->n_runs;
    v.stats = stats;
    return v;
}
// A DevRun: its first eight words in scalar registers (one scalar load); the others are read
// where a long run needs them (all 32 at once would push as many wave-uniform values of the
// loop out of the scalar register file for every run of the list)
struct RunView {
    uint32_t kind, first, count, id0, mode, n_levels, top_steps, order_base;
    const RT_CONST_AS uint32_t *words;
    __device__ __forceinline__ uint32_t level_base0() const { return words[8]; }
    __device__ __forceinline__ float level_keep0() const { return __uint_as_float(words[16]); }   // RUN_FLAT
    // reach, 1 / (2 smallest radius), primitives outside the tree
    __device__ __forceinline__ u4v tail() const { return *(const RT_CONST_AS u4v *)(words + 24); }
};
__device__ __forceinline__ RunView load_run(const SceneView &sc, uint32_t r) {
    RunView v;
    const RT_CONST_AS u8v *rec = sc.runs() + 4 * r;
    const u8v head = rec[0];
    v.words = (const RT_CONST_AS uint32_t *)rec;
    v.kind = head.s0; v.first = head.s1; v.count = head.s2; v.id0 = head.s3;
    v.mode = head.s4; v.n_levels = head.s5; v.top_steps = head.s6; v.order_base = head.s7;
    return v;
}

// wavefronts per workgroup: 8 when a tree copy in LDS is shared, else 4
__host__ __device__ constexpr int waves_of(int modes) { return (modes & 2) ? RTK_TREE_WAVES : 4; }
// wavefronts per SIMD the kernel is compiled for
#ifndef RTK_OCC
#define RTK_OCC 4
#endif
#ifndef RTK_OCC0
#define RTK_OCC0 8           // scenes without long runs: 60 VGPRs, no spills.  Default scene at 1080p, 64 samples:
#endif                       // 6.40 ms with four waves per SIMD, 5.70 with five, 5.48 with six and with eight
__host__ __device__ constexpr int occ_of(int modes) { return modes == 0 ? RTK_OCC0 : RTK_OCC; }
// What the compiler is told: one wave less for the kernel without long runs.  At "eight waves" it
// keeps to 80 scalar registers and parks 30 wave-uniform values in lanes of vector registers; told
// seven it takes 94 (14 parked) and still fits eight waves per SIMD (64 VGPRs, 96 SGPRs, 20 KB of LDS
// per workgroup): default scene -1.4 %.  The launch asks the runtime what really fits (blocks_per_cu).
__host__ __device__ constexpr int occ_hint_of(int modes) { return modes == 0 ? RTK_OCC0 - 1 : RTK_OCC; }
// idle lanes a wave waits for before it counts finished items and hands out new ones
// (64-sample launches of the whole frame, 1 / 16 / 24 / 32 lanes: default scene 1080p 4.97 / 4.70 / 4.57 /
// 4.58 ms; with a tree 8 / 16 / 20 / 24 / 32 lanes: 10,000 spheres 4K 107.8 / 105.6 / 105.6 / 106.4 /
// 107.6 ms, a million triangles 53.9 / 51.4 / 50.5 / 50.3 / 50.3, 960-triangle mesh 10.35 / 10.08 / 9.97 /
// 9.94 / 9.97; with the later kernel whose nodes store their planes twice, 20 / 24 lanes: 10,000 spheres 98.9 / 99.1 ms, a million
// triangles 46.8 / 46.2, mesh 9.27 / 9.20: a scene whose largest tree holds triangles waits for 24)
#ifndef RTK_FETCH_MIN0
#define RTK_FETCH_MIN0 24
#endif
#ifndef RTK_FETCH_MIN
#define RTK_FETCH_MIN 20
#endif
#ifndef RTK_FETCH_MIN_TRI
#define RTK_FETCH_MIN_TRI 24
#endif
constexpr int TILE = 8;   // 8x8 pixels per wavefront

struct TreeTable {
    uint32_t base[DEV_MAX_LEVELS];    // float4 offset of each level in tree_rec
    uint32_t count[DEV_MAX_LEVELS];   // boxes at each level
};

// The work units a wave has in flight (see render_tiles): column | row << 16 of the unit's first
// pixel (row counted from the bottom), and the number of items not finished yet (0 = slot free).
// `fr`: the frame of a launch over several frames the unit belongs to (render_tiles<.., FRAMES = true>).
struct UnitRing { uint32_t xy[4], left[4], fr[4]; };

// MODES: bit 0 = the scene has RUN_FLAT runs, bit 1 = it has RUN_TREE runs; the
// kernel is compiled once per combination so that scenes without a tree do not
// pay registers and LDS for it
// DEPTH: the largest depth budget the kernel takes (DEV_STACK_SHALLOW or RT_MAX_BOUNCES; with a tree
// the shallow kernel has room for more of it in LDS)
template <int MODES, int DEPTH>
struct Lds {
    // top levels of one tree (DevScene::lds_*); first, so that reads of two 8-byte halves reach a
    // whole node with their 8-bit offsets
    f4 tree_top[(MODES & 2) ? (DEPTH > (int)DEV_STACK_SHALLOW ? DEV_TREE_LDS_F4_DEEP : DEV_TREE_LDS_F4) : 1];
    float disc[128];                                    // src/denoising/mod.rs:55
    float sphere[768];                                  // src/denoising/mod.rs:59
    uint32_t mat_stack[waves_of(MODES)][DEPTH][64];     // materials of the path's hits so far, per lane
    TreeTable tree[waves_of(MODES)];                    // per-level constants of the run being walked
    UnitRing ring[waves_of(MODES)];                     // the units each wave has in flight
    float uv[4];                                        // width - 1, height - 1, their refined reciprocals
};

// two 8-wave workgroups of a tree kernel share a CU's 160 KiB of LDS
static_assert(2u * sizeof(Lds<2, (int)DEV_STACK_SHALLOW>) <= 160u * 1024u && 2u * sizeof(Lds<3, (int)DEV_STACK_SHALLOW>) <= 160u * 1024u &&
              2u * sizeof(Lds<2, (int)RT_MAX_BOUNCES>) <= 160u * 1024u && 2u * sizeof(Lds<3, (int)RT_MAX_BOUNCES>) <= 160u * 1024u,
              "DEV_TREE_LDS_F4 / DEV_TREE_LDS_F4_DEEP leave no room for two workgroups per CU");

struct Ray { V3 o, d; };

// Camera::get_ray, src/camera/mod.rs:75-79
template <class Cam>   // the camera in the kernel's arguments (scalar loads), or one of a launch's cameras in device memory
__device__ __forceinline__ Ray get_ray(const Cam &c, float s, float t, float da, float db) {
    V3 origin = mk(c.origin[0], c.origin[1], c.origin[2]);
    V3 llc = mk(c.lower_left_corner[0], c.lower_left_corner[1], c.lower_left_corner[2]);
    V3 hor = mk(c.horizontal[0], c.horizontal[1], c.horizontal[2]);
    V3 ver = mk(c.vertical[0], c.vertical[1], c.vertical[2]);
    V3 cu = mk(c.u[0], c.u[1], c.u[2]);
    V3 cv = mk(c.v[0], c.v[1], c.v[2]);
    V3 rd = mk(da, db, 0.0f) * c.lens_radius;
    V3 offset = cu * rd.x + cv * rd.y;
    Ray r;
    r.o = origin + offset;
    r.d = llc + hor * s + ver * t - origin - offset;
    return r;
}

// BlueNoise::random_in_unit_sphere, src/denoising/mod.rs:37-50: (v*sp_bn)*1.1
__device__ __forceinline__ V3 rius(const float *tbl_sphere, uint32_t seed, float sp_bn) {
    uint32_t k = seed & 255u;
    V3 t = mk(tbl_sphere[3 * k], tbl_sphere[3 * k + 1], tbl_sphere[3 * k + 2]);
    return (t * sp_bn) * 1.1f;
}

// f32::powi(5) as LLVM expands it
__device__ __forceinline__ float powi5(float x) { float x2 = x * x; float x4 = x2 * x2; return x * x4; }

// Optional instrumentation (build with -DRTK_STATS): wave-level event counts in
// fr.segments[1..7].  Not compiled into the shipped library.
#ifdef RTK_STATS
#define RTK_COUNT(sc, slot, n)                                                              \
    do {                                                                                    \
        unsigned long long n_ = (unsigned long long)(n);                                    \
        if (__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)) == 0u)       \
            atomicAdd((sc).stats + (slot), n_);                                             \
    } while (0)
// the same from divergent code: counted once per wave by its first active lane
#define RTK_COUNT_ANY(sc, slot, n)                                                          \
    do {                                                                                    \
        unsigned long long n_ = (unsigned long long)(n);                                    \
        if ((threadIdx.x & 63u) == (uint32_t)__builtin_ctzll(__builtin_amdgcn_ballot_w64(true))) \
            atomicAdd((sc).stats + (slot), n_);                                             \
    } while (0)
#else
#define RTK_COUNT(sc, slot, n) do { } while (0)
#define RTK_COUNT_ANY(sc, slot, n) do { } while (0)
#endif

struct Hit {
    float t;        // closest_so_far, src/geometry/mod.rs:242
    int kind;       // -1 = nothing hit
    uint32_t slot;  // position in the kind's device array
    int32_t id;     // position in the reference's list
    uint32_t ff;    // front_face (1 or 0) as the shared temp record held it at accept time; kept as a number:
                    // a bool carried across divergent branches lives in scalar mask registers, three scalar
                    // instructions per join
};

// ----------------------------------------------------------------------
// Exact primitive tests.  They update (h.t, h.kind, h.slot, h.id) like the
// reference's list loop updates closest_so_far and its record.  A scan that
// does not visit a run in list order (the tree walk) must still
// end with the primitive the list-order scan ends with: among equal hit
// distances the reference keeps the LATER sphere / plane / triangle
// (`t_max < root` rejects, equality accepts, src/geometry/mod.rs:74,116), so
// an equal-distance candidate with a smaller list position than the current
// holder is refused here.  In a list-order scan positions only grow and the
// extra condition never fires.
// The decision functions ask for list positions through callbacks, only when two distances
// tie: a list-order scan knows them, the tree walk reads them from the by-slot tables.
// ----------------------------------------------------------------------
// What the exact tests need to know about the ray beyond origin and direction: `a` is |dir|^2
// (Sphere::hit, src/geometry/mod.rs:63, the same for every sphere).  `cheap` is wave-uniform and
// says that every ray being traced has 2^-24 <= a <= 2^24 and |origin coordinates| <= 2^30, every
// sphere of the scene moderate coordinates (DevScene::cheap_ok) and t_min >= 2^-20: then the
// quotients by `a` may be div_refined with the shared reciprocal inv_a.  (Their numerators stay
// below 2^60; one below 2^-90 gives a root below 2^-66, which both this and `/` put in front of
// t_min, so its bits are never looked at.)
struct RayAux { float a, inv_a; bool cheap; };
// Sphere::hit, src/geometry/mod.rs:60-94, up to the decision: true, with the hit distance in
// `root`, when the sphere g replaces a holder at distance `closest`.  this_pos() and
// holder_pos() give the list positions of the two, asked for only when the distances tie.
template <class ThisPos, class HolderPos>
__device__ __forceinline__ bool sphere_closer(const f4 g, const Ray &ray, const RayAux &aux,
                                              float t_min, float closest, ThisPos this_pos, HolderPos holder_pos,
                                              float &root) {
    const float a = aux.a;            // g: cx, cy, cz, radius
    V3 oc = ray.o - mk(g.x, g.y, g.z);
    float half_b = dot(oc, ray.d);
    float c = len2(oc) - g.w * g.w;
    float disc = half_b * half_b - a * c;
    if (disc < 0.0f) return false;
    float d_sqrt = sqrt_mostly(disc);
    if (aux.cheap) {
        root = div_refined(-(half_b + d_sqrt), a, aux.inv_a);
        if (root < t_min || closest < root) {
            root = div_refined(-half_b + d_sqrt, a, aux.inv_a);
            if (root < t_min || closest < root) return false;
        }
    } else {
        root = -(half_b + d_sqrt) / a;
        if (root < t_min || closest < root) {
            root = (-half_b + d_sqrt) / a;
            if (root < t_min || closest < root) return false;
        }
    }
    if (root == closest && this_pos() < holder_pos()) return false;
    return true;
}
// front_face of a sphere hit at distance t, src/geometry/mod.rs:81-86,30-33
__device__ __forceinline__ bool sphere_front_face(const SceneView &sc, uint32_t slot, const Ray &ray, float t, bool cheap) {
    const f4 g = sc.sph_geom()[slot];
    V3 p = ray.o + ray.d * t;
    V3 outward = div3(p - mk(g.x, g.y, g.z), g.w, cheap);
    return dot(ray.d, outward) < 0.0f;
}
// one sphere of a list-order scan (g: its record)
__device__ __forceinline__ void sphere_exact(const SceneView &sc, const f4 g, uint32_t slot, int32_t id, const Ray &ray,
                                             const RayAux &aux, float t_min, Hit &h, uint32_t &ff_shared) {
    float root;
    if (!sphere_closer(g, ray, aux, t_min, h.t, [&]() { return id; }, [&]() { return h.id; }, root)) return;
    h.t = root; h.kind = DEV_SPHERE; h.slot = slot; h.id = id;
    if (sc.need_ff_shared()) ff_shared = sphere_front_face(sc, slot, ray, root, aux.cheap) ? 1u : 0u;
}
__device__ __forceinline__ void sphere_exact(const SceneView &sc, uint32_t slot, int32_t id, const Ray &ray,
                                             const RayAux &aux, float t_min, Hit &h, uint32_t &ff_shared) {
    sphere_exact(sc, sc.sph_geom()[slot], slot, id, ray, aux, t_min, h, ff_shared);
}
// same selections as `m = a > b ? a : b; m > c ? m : c` (and < for min): with a NaN operand the
// results may differ from a fused max3, so they are spelled out
__device__ __forceinline__ float max3(float a, float b, float c) { float m = a > b ? a : b; return m > c ? m : c; }
__device__ __forceinline__ float min3(float a, float b, float c) { float m = a < b ? a : b; return m < c ? m : c; }
// Triangle (defined by rt_hip.h, absent from the reference): Moller-Trumbore, then the
// bounding-box gate that keeps the f32 barycentric test from accepting points far outside
// a sliver or an edge-on triangle
template <class ThisPos, class HolderPos>
__device__ __forceinline__ bool tri_closer(const RT_CONST_AS f4 *tris, uint32_t slot, const Ray &ray, float t_min, float closest,
                                           ThisPos this_pos, HolderPos holder_pos, float &t) {
    const f4 v0 = tris[3 * slot];
    const f4 e1 = tris[3 * slot + 1];
    const f4 e2 = tris[3 * slot + 2];
    // all three fetched now: left to itself the compiler fetches v0 only after the determinant
    // test, a second round trip to memory for most triangles that get this far
    asm volatile("" ::"v"(v0.x), "v"(e1.x), "v"(e2.x));
    V3 E1 = mk(e1.x, e1.y, e1.z), E2 = mk(e2.x, e2.y, e2.z);
    V3 hh = cross(ray.d, E2);
    float det = dot(E1, hh);
    if (det == 0.0f) return false;
    float f = 1.0f / det;
    V3 s = ray.o - mk(v0.x, v0.y, v0.z);
    float u = f * dot(s, hh);
    if (u < 0.0f || u > 1.0f) return false;
    V3 q = cross(s, E1);
    float v = f * dot(ray.d, q);
    if (v < 0.0f || u + v > 1.0f) return false;
    t = f * dot(E2, q);
    if (t < t_min || closest < t) return false;
    // gate: the hit point must lie in the widened bounding box of the triangle
    {
        const V3 a0 = mk(v0.x, v0.y, v0.z), a1 = a0 + E1, a2 = a0 + E2, p = ray.o + ray.d * t;
        const V3 lo = mk(min3(a0.x, a1.x, a2.x), min3(a0.y, a1.y, a2.y), min3(a0.z, a1.z, a2.z));
        const V3 hi = mk(max3(a0.x, a1.x, a2.x), max3(a0.y, a1.y, a2.y), max3(a0.z, a1.z, a2.z));
        const float ext = max3(hi.x - lo.x, hi.y - lo.y, hi.z - lo.z);
        const float mag = max3(max3(__builtin_fabsf(lo.x), __builtin_fabsf(lo.y), __builtin_fabsf(lo.z)),
                               max3(__builtin_fabsf(hi.x), __builtin_fabsf(hi.y), __builtin_fabsf(hi.z)),
                               max3(__builtin_fabsf(ray.o.x), __builtin_fabsf(ray.o.y), __builtin_fabsf(ray.o.z)));
        const float delta = ext * 0.0078125f + mag * 9.5367431640625e-07f;
        if (!(p.x >= lo.x - delta && p.x <= hi.x + delta && p.y >= lo.y - delta && p.y <= hi.y + delta &&
              p.z >= lo.z - delta && p.z <= hi.z + delta)) return false;
    }
    if (t == closest && this_pos() < holder_pos()) return false;
    return true;
}
// front_face of a triangle hit: outward normal unit(cross(e1, e2))
__device__ __forceinline__ bool tri_front_face(const SceneView &sc, uint32_t slot, const Ray &ray) {
    const RT_CONST_AS f4 *tris = sc.tris();
    const f4 e1 = tris[3 * slot + 1];
    const f4 e2 = tris[3 * slot + 2];
    V3 outward = unit(cross(mk(e1.x, e1.y, e1.z), mk(e2.x, e2.y, e2.z)));
    return dot(ray.d, outward) < 0.0f;
}
// one triangle of a list-order scan
__device__ __forceinline__ void tri_exact(const SceneView &sc, uint32_t slot, int32_t id, const Ray &ray,
                                          float t_min, Hit &h, uint32_t &ff_shared) {
    float t;
    if (!tri_closer(sc.tris(), slot, ray, t_min, h.t, [&]() { return id; }, [&]() { return h.id; }, t)) return;
    h.t = t; h.kind = DEV_TRI; h.slot = slot; h.id = id;
    if (sc.need_ff_shared()) ff_shared = tri_front_face(sc, slot, ray) ? 1u : 0u;
}

// ----------------------------------------------------------------------
// Conservative ball rejection.  With u the unit direction, the squared
// distance from a centre c to the ray's line is |c-o|^2 - ((c-o).u)^2; a ball
// of radius r cannot be hit when that exceeds r^2.  Expanded so that no
// per-ball subtraction is needed:
//     g = |o|^2 (1-eps) - 2 o.c - ((c.u) - (o.u))^2        (7 fused ops, packed two
//                                                            balls per v_pk_fma_f32)
//     reject  <=>  g > r^2 (1+eps) - |c|^2 (1-eps) = thr   (1 compare)
// i.e. dist^2 - r^2 > eps (|o|^2 + |c|^2 + r^2).  Sphere::hit misses (its
// discriminant, evaluated in f32 without fusing, is negative) whenever
// dist^2 - r^2 > 19.5 ulp |o-c|^2; the rounding of the seven fused ops, of
// u, o.u and thr adds at most 30 ulp (|o|+|c|)^2, and (|o|+|c|)^2 <=
// 2(|o|^2+|c|^2), so eps = 2^-16 = 256 ulp covers both with a factor 2.6 to
// spare.  Bounding balls of the tree use larger eps (see the host-side
// builder); eps enters through thr and through the per-level factor `keep`.
// The test never ACCEPTS anything: survivors run the exact test itself.
// NaN intermediates compare false and survive.  The bounds above are relative
// ones and need finite, normal intermediates: a ray whose |d|^2 leaves
// (1e-12, 1e12) or whose |o|^2 reaches 1e36 gets |o|^2 = NaN and rejects
// nothing, and so does one with |d|^2 (|o| + extent of the run)^2 >= 1e37 (scan_flat:
// Sphere::hit may then work with infinities and accept NaN roots).
// ----------------------------------------------------------------------
struct Cull { float ux, uy, uz, nou, ax, ay, az, oo; };

__device__ __forceinline__ Cull cull_setup(const Ray &ray, float a) {
    Cull c;
    const float len = __builtin_sqrtf(a);
    c.ux = ray.d.x / len; c.uy = ray.d.y / len; c.uz = ray.d.z / len;
    c.nou = -__builtin_fmaf(ray.o.x, c.ux, __builtin_fmaf(ray.o.y, c.uy, ray.o.z * c.uz));
    c.ax = -2.0f * ray.o.x; c.ay = -2.0f * ray.o.y; c.az = -2.0f * ray.o.z;
    c.oo = __builtin_fmaf(ray.o.x, ray.o.x, __builtin_fmaf(ray.o.y, ray.o.y, ray.o.z * ray.o.z));
    if (!(a > 1e-12f && a < 1e12f && c.oo < 1e36f)) c.oo = __builtin_nanf("");
    return c;
}
// g of two balls at once: seven v_pk_fma_f32, each taking one SGPR pair (the
// same coordinate of both balls) and ray values broadcast from one VGPR.
// oo_keep = |o|^2 (1-eps) for participating lanes, +inf for lanes that must
// reject everything.
__device__ __forceinline__ f2 cull_pair(const Cull &c, float oo_keep, f2 cx, f2 cy, f2 cz) {
    const f2 ux = {c.ux, c.ux}, uy = {c.uy, c.uy}, uz = {c.uz, c.uz}, nou = {c.nou, c.nou};
    const f2 ax = {c.ax, c.ax}, ay = {c.ay, c.ay}, az = {c.az, c.az}, oo = {oo_keep, oo_keep};
    const f2 m = __builtin_elementwise_fma(cx, ux, __builtin_elementwise_fma(cy, uy, __builtin_elementwise_fma(cz, uz, nou)));
    const f2 w = __builtin_elementwise_fma(cx, ax, __builtin_elementwise_fma(cy, ay, __builtin_elementwise_fma(cz, az, oo)));
    return __builtin_elementwise_fma(-m, m, w);
}

// One step: eight balls given as four pair records {cx0,cx1, cy0,cy1, cz0,cz1,
// thr0,thr1} in ra and rb (SGPRs).  Returns, per lane, the bit mask of balls the
// lane could not reject; any_bits gets the wave-wide union (uniform).
__device__ __forceinline__ uint32_t cull_step(const Cull &cull, float oo_keep, const f16 ra, const f16 rb,
                                              uint32_t &any_bits) {
    const f2 g01 = cull_pair(cull, oo_keep, ra.s01, ra.s23, ra.s45);
    const f2 g23 = cull_pair(cull, oo_keep, ra.s89, ra.sab, ra.scd);
    const f2 g45 = cull_pair(cull, oo_keep, rb.s01, rb.s23, rb.s45);
    const f2 g67 = cull_pair(cull, oo_keep, rb.s89, rb.sab, rb.scd);
    const bool p0 = !(g01.x > ra.s6), p1 = !(g01.y > ra.s7);
    const bool p2 = !(g23.x > ra.se), p3 = !(g23.y > ra.sf);
    const bool p4 = !(g45.x > rb.s6), p5 = !(g45.y > rb.s7);
    const bool p6 = !(g67.x > rb.se), p7 = !(g67.y > rb.sf);
    // the compare masks already are SGPR pairs: the wave-level "does any lane need
    // more" is a chain of s_or_b64 and one scalar branch
    const unsigned long long m0 = __builtin_amdgcn_ballot_w64(p0), m1 = __builtin_amdgcn_ballot_w64(p1);
    const unsigned long long m2 = __builtin_amdgcn_ballot_w64(p2), m3 = __builtin_amdgcn_ballot_w64(p3);
    const unsigned long long m4 = __builtin_amdgcn_ballot_w64(p4), m5 = __builtin_amdgcn_ballot_w64(p5);
    const unsigned long long m6 = __builtin_amdgcn_ballot_w64(p6), m7 = __builtin_amdgcn_ballot_w64(p7);
    any_bits = 0u;
    if ((m0 | m1 | m2 | m3 | m4 | m5 | m6 | m7) == 0ull) return 0u;
    // rare (a few percent of the steps)
    any_bits = (m0 ? 1u : 0u) | (m1 ? 2u : 0u) | (m2 ? 4u : 0u) | (m3 ? 8u : 0u) |
               (m4 ? 16u : 0u) | (m5 ? 32u : 0u) | (m6 ? 64u : 0u) | (m7 ? 128u : 0u);
    return (p0 ? 1u : 0u) | (p1 ? 2u : 0u) | (p2 ? 4u : 0u) | (p3 ? 8u : 0u) |
           (p4 ? 16u : 0u) | (p5 ? 32u : 0u) | (p6 ? 64u : 0u) | (p7 ? 128u : 0u);
}

// exact test of the primitives of one leaf step that some lane could not reject
template <int KIND>
__device__ __forceinline__ void leaf_exact(const SceneView &sc, const RunView &run, uint32_t first_in_run,
                                           uint32_t any_bits, uint32_t lane_bits, bool list_order,
                                           const Ray &ray, const RayAux &aux, float t_min, Hit &h, uint32_t &ff_shared) {
    if (first_in_run >= run.count) return;          // a step of padding only
    const uint32_t left = run.count - first_in_run;
#pragma unroll 1
    for (uint32_t k = 0; k < 8u; ++k) {
        if (((any_bits >> k) & 1u) == 0u || k >= left) continue;
        RTK_COUNT(sc, 4, 1);
        const uint32_t slot = run.first + first_in_run + k;
        const int32_t id = list_order ? (int32_t)(run.id0 + first_in_run + k)
                                      : (int32_t)(KIND == DEV_SPHERE ? sc.sph_id()[slot] : sc.tri_id()[slot]);
        if ((lane_bits >> k) & 1u) {
            if (KIND == DEV_SPHERE) sphere_exact(sc, slot, id, ray, aux, t_min, h, ff_shared);
            else tri_exact(sc, slot, id, ray, t_min, h, ff_shared);
        }
    }
}

// ----------------------------------------------------------------------
// RUN_FLAT: a long sphere run in list order, eight spheres per step through
// the rejection test.  Two steps per trip in ping-pong register sets: the
// records of the next step are requested BEFORE the current step is computed,
// so the scalar-cache / L2 latency of the s_load_dwordx16 pair overlaps the
// arithmetic.  Scalar loads return out of order, the only wait is "all of
// them"; it sits right before the next request, when the awaited data has had
// a whole step to arrive.
// ----------------------------------------------------------------------
__device__ __forceinline__ void scan_flat(const SceneView &sc, const RunView &run, const Cull &cull, bool tracing,
                                          const Ray &ray, const RayAux &aux, float t_min, Hit &h, uint32_t &ff_shared) {
    // Every intermediate of Sphere::hit stays below 2 a S^2, S = |o| + reach >= |o - c| + |r| for the
    // run's bounded spheres.  Where that may overflow, the discriminant can come out as inf - inf and
    // the reference accepts the NaN root of a sphere the ray passes far from: nothing is rejected then.
    const float S = __builtin_amdgcn_sqrtf(cull.oo) * 1.000001f + __uint_as_float(run.tail().x);
    const float oo = aux.a * (S * S) < 1e37f ? cull.oo : __builtin_nanf("");
    const float oo_keep = tracing ? oo * run.level_keep0() : __builtin_inff();
    const RT_CONST_AS f16 *rec = sc.cull_rec() + 2u * run.level_base0();
    const uint32_t trips = run.top_steps / 2u;
    f16 a0 = rec[0], a1 = rec[1];
    for (uint32_t b = 0; b < trips; ++b) {
        uint32_t any_bits, lane_bits;
        asm volatile("; records of the even step have arrived" ::"s"(a0.s0), "s"(a1.s0));
        __builtin_amdgcn_sched_barrier(0);
        const f16 b0 = rec[4 * b + 2], b1 = rec[4 * b + 3];
        __builtin_amdgcn_sched_barrier(0);
        RTK_COUNT(sc, 2, 2);
        lane_bits = cull_step(cull, oo_keep, a0, a1, any_bits);
        if (any_bits) {
            RTK_COUNT(sc, 3, 1);
            leaf_exact<DEV_SPHERE>(sc, run, 16u * b, any_bits, lane_bits, true, ray, aux, t_min, h, ff_shared);
        }
        asm volatile("; records of the odd step have arrived" ::"s"(b0.s0), "s"(b1.s0));
        __builtin_amdgcn_sched_barrier(0);
        const uint32_t nb = (b + 1u < trips) ? b + 1u : b;   // the last trip re-reads itself
        a0 = rec[4 * nb]; a1 = rec[4 * nb + 1];
        __builtin_amdgcn_sched_barrier(0);
        lane_bits = cull_step(cull, oo_keep, b0, b1, any_bits);
        if (any_bits) {
            RTK_COUNT(sc, 3, 1);
            leaf_exact<DEV_SPHERE>(sc, run, 16u * b + 8u, any_bits, lane_bits, true, ray, aux, t_min, h, ff_shared);
        }
    }
}

// ----------------------------------------------------------------------
// RUN_TREE: the run's primitives are stored in a spatially hierarchical order under an
// implicit 8-ary tree of axis-aligned boxes.  Level 0 holds one box per primitive in steps of
// eight (a sphere's extent, a triangle's gate box), level L+1 one bounding box per step of
// level L (the children of step s of level L+1 are the steps 8s..8s+7 one level down).  Every
// lane walks the tree for its own ray.  No stack: the path is the step index itself
// (parent = s >> 3), the children still to visit are one byte per level in a 64-bit register.
//
// Per trip a lane fetches ITS node, tests the eight child boxes and descends into
// the survivor the ray enters first; on level 0 the survivors get the exact test instead.
// When no child survives, or after a leaf step, the lane moves on to its next pending node;
// the walk ends when a ballot says no lane has a node left.
//
// A child box [lo, hi] is skipped when the ray misses the box widened by `pad` on every
// side, lies behind the origin, or starts beyond closest_so_far (slab test:
// t = (plane - o) / d per axis, with 1/d and the offsets folded into one FMA per plane).
// `pad` covers (a) how far the reference's f32 arithmetic lets a primitive reach beyond its
// geometric extent and (b) the rounding of the slab test, both proportional to
// S = |o| + reach >= |o - x| for every point x of the run:
//   (a) spheres: the discriminant of Sphere::hit is uncertain by kappa a |o-c|^2 with
//       kappa < 2^-19, so a root lies up to sqrt(r^2 + kappa |o-c|^2) from the centre:
//       at most min(sqrt(kappa) S, kappa S^2 / (2 r_min)) beyond r;
//       triangles: the gate box grows by 2^-20 |o|, the hit point o + t*d it is applied
//       to is off by < 4 ulp S;
//   (b) each t is off by a few ulp of (|plane| + |o| + pad) |1/d|, i.e. the plane by a few
//       ulp of S.  2^-17 S covers (b) and the triangles' (a) three times over.
// A node stores, per pair of child boxes and axis, {lo0, lo1, hi0, hi1, lo0, lo1}: a lane reads
// four of the six, from the first if its ray runs up that axis and from the third if it runs
// down, and so gets the planes it reaches first in the first two and the planes it leaves
// through in the last two without comparing anything (192 of the node's 288 bytes; the entry
// distance is then one v_max3 and one v_max, the exit distance one v_min3 and one v_min).
// A lane leaves the walk as soon as its closest_so_far turns NaN (see scan_tree).
// ----------------------------------------------------------------------
// The walk keeps the closest hit so far as a distance and a slot of the run (NO_SLOT: still
// the hit the ray came with, at list position entry_pos); the hit record is written once, after
// the walk.  Exact test of the primitive in `slot` against that state.
constexpr uint32_t NO_SLOT = 0xffffffffu;
template <int KIND>
__device__ __forceinline__ bool walk_test(const RT_CONST_AS f4 *geom, const RT_CONST_AS uint32_t *by_slot, uint32_t slot,
                                          const Ray &ray, const RayAux &aux,
                                          float t_min, int32_t entry_pos, float best_t, uint32_t best_slot, float &t) {
    auto this_pos = [&]() { return (int32_t)by_slot[slot]; };
    auto holder_pos = [&]() { return best_slot == NO_SLOT ? entry_pos : (int32_t)by_slot[best_slot]; };
    if (KIND == DEV_SPHERE) return sphere_closer(geom[slot], ray, aux, t_min, best_t, this_pos, holder_pos, t);
    return tri_closer(geom, slot, ray, t_min, best_t, this_pos, holder_pos, t);
}

template <int KIND>
__device__ __forceinline__ void walk_tree(const SceneView &sc, const RunView &run, uint32_t r, bool walking,
                                          TreeTable &tbl, uint32_t &tbl_run, const f4 *tree_top, uint32_t lds_from_level,
                                          const RT_CONST_AS f4 *geom, const RT_CONST_AS uint32_t *by_slot,
                                          const Ray &ray, const RayAux &aux, float t_min, int32_t entry_pos,
                                          float &best_t, uint32_t &best_slot) {
    // per-level constants where a lane can index them by its own level (tbl_run: the run whose
    // constants the wave's table holds; a scene with one tree writes them once per launch)
    if (tbl_run != r) {
        // (lane L copies the two words of level L)
        const uint32_t L = (uint32_t)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        const uint32_t *w = (const uint32_t *)(uintptr_t)run.words;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        if (L < DEV_MAX_LEVELS) { tbl.base[L] = w[8u + L]; tbl.count[L] = w[16u + L]; }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        tbl_run = r;
    }
    const u4v tail = run.tail();
    // (S only has to be an upper bound and 1/d only feeds the widened slab test: the hardware's
    // 1-ulp square root and reciprocal do, the square root scaled up by 8 ulp)
    const float S = __builtin_amdgcn_sqrtf(len2(ray.o)) * 1.000001f + __uint_as_float(tail.x);
    float reach_pad = 0.0f;
    if (KIND == DEV_SPHERE) {
        const float kS = S * (1.0f / 524288.0f);
        reach_pad = __builtin_fminf(S * 1.3811e-3f, kS * S * __uint_as_float(tail.y));
    }
    const float pad = S * (1.0f / 131072.0f) + reach_pad;
    // slab test constants: t = plane * inv + k, k = -(o -+ pad) * inv.
    // 1/d clamped to +-1e19: a zero component would make plane * inv and k infinities of
    // opposite sign.  With |d| in (1e-6, 1e6) (see scan_tree) a component below 1e-19 moves
    // the ray by less than 2e-13 S over any distance that matters, far inside the pad.
    const float ix = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(ray.d.x), -1e19f, 1e19f);
    const float iy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(ray.d.y), -1e19f, 1e19f);
    const float iz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(ray.d.z), -1e19f, 1e19f);
    const float klx = -((ray.o.x + pad) * ix), kly = -((ray.o.y + pad) * iy), klz = -((ray.o.z + pad) * iz);
    const float khx = -((ray.o.x - pad) * ix), khy = -((ray.o.y - pad) * iy), khz = -((ray.o.z - pad) * iz);
#if DEV_NODE_MIRROR
    const uint32_t offx = ix < 0.0f ? 2u : 0u, offy = iy < 0.0f ? 2u : 0u, offz = iz < 0.0f ? 2u : 0u;
    const float knx = ix < 0.0f ? khx : klx, kfx = ix < 0.0f ? klx : khx;
    const float kny = iy < 0.0f ? khy : kly, kfy = iy < 0.0f ? kly : khy;
    const float knz = iz < 0.0f ? khz : klz, kfz = iz < 0.0f ? klz : khz;
#endif
    const uint32_t top = run.n_levels - 1u;
    const RT_CONST_AS f4 *nodes = sc.tree_rec();
    const uint32_t lds_first = sc.lds().z;
    // no hit is accepted in front of t_min; the bound lies a little below it, so that a box
    // holding a hit at exactly t_min (and at closest_so_far, a tie) keeps an interval of its own
    // (0 for the usual positive t_min: entry distances that are not negative order like their
    // bit patterns, which the choice of the nearest child relies on)
    const float t_low = t_min > 0.0f ? 0.0f : t_min - __builtin_fmaxf(__builtin_fabsf(t_min) * 9.5367431640625e-07f, 1e-30f);

    for (uint32_t ts = 0; ts < run.top_steps; ++ts) {
        uint32_t L = top, st = ts;
        unsigned long long pend = 0ull;
        bool act = walking;
#ifdef RTK_STATS
        while (__builtin_amdgcn_ballot_w64(act) != 0ull) {
            RTK_COUNT(sc, 2, 1);
            RTK_COUNT(sc, 5, __builtin_popcountll(__builtin_amdgcn_ballot_w64(act)));
            if (act) {
#else
        // (a lane leaves the loop when it has no node left; the wave goes on while any lane is inside)
        while (act) {
            {
#endif
#if DEV_NODE_MIRROR
                // the node's records as floats; a lane reads four from each, starting at the first
                // (near planes = low planes) or the third (near planes = high planes)
                const uint32_t at = 4u * tbl.base[L] + 72u * st;
                f4 nd[12];
                if (L >= lds_from_level) {
                    const float *src = (const float *)tree_top + (at - 4u * lds_first);
                    const float *sx = src + offx, *sy = src + offy, *sz = src + offz;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        nd[3 * pr] = *(const f4a8 *)(sx + 18 * pr);
                        nd[3 * pr + 1] = *(const f4a8 *)(sy + 18 * pr + 6);
                        nd[3 * pr + 2] = *(const f4a8 *)(sz + 18 * pr + 12);
                    }
                } else {
                    const RT_CONST_AS float *src = (const RT_CONST_AS float *)nodes + (size_t)at;
                    const RT_CONST_AS float *sx = src + offx, *sy = src + offy, *sz = src + offz;
#pragma unroll
                    for (int pr = 0; pr < 4; ++pr) {
                        nd[3 * pr] = *(const RT_CONST_AS f4a8 *)(sx + 18 * pr);
                        nd[3 * pr + 1] = *(const RT_CONST_AS f4a8 *)(sy + 18 * pr + 6);
                        nd[3 * pr + 2] = *(const RT_CONST_AS f4a8 *)(sz + 18 * pr + 12);
                    }
                }
                uint32_t live = 0u;               // children the ray may hit
                uint32_t nearest = 0xffffffffu;   // min over them of (entry distance | index)
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const f4 X = nd[3 * pr], Y = nd[3 * pr + 1], Z = nd[3 * pr + 2];   // near0, near1, far0, far1
                    const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz};
                    const f2 tnx = __builtin_elementwise_fma(X.xy, ix2, (f2){knx, knx}), tfx = __builtin_elementwise_fma(X.zw, ix2, (f2){kfx, kfx});
                    const f2 tny = __builtin_elementwise_fma(Y.xy, iy2, (f2){kny, kny}), tfy = __builtin_elementwise_fma(Y.zw, iy2, (f2){kfy, kfy});
                    const f2 tnz = __builtin_elementwise_fma(Z.xy, iz2, (f2){knz, knz}), tfz = __builtin_elementwise_fma(Z.zw, iz2, (f2){kfz, kfz});
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        // [enter, leave] = [t_low, closest_so_far] cut down by the three slabs
                        float enter = __builtin_fmaxf(__builtin_fmaxf(c ? tnx.y : tnx.x, c ? tny.y : tny.x), c ? tnz.y : tnz.x);
                        float leave = __builtin_fminf(__builtin_fminf(c ? tfx.y : tfx.x, c ? tfy.y : tfy.x), c ? tfz.y : tfz.x);
                        enter = __builtin_fmaxf(enter, t_low);
                        leave = __builtin_fminf(leave, best_t);
                        const uint32_t m = (uint32_t)((int32_t)__float_as_uint(enter - leave) >> 31);
                        const uint32_t child = (uint32_t)(2 * pr + c);
                        live |= m & (1u << child);
                        const uint32_t key = (__float_as_uint(enter) & ~7u) | child;
                        nearest = min(nearest, (key & m) | ~m);
                    }
                }
#else
                const uint32_t at = tbl.base[L] + 12u * st;
                f4 nd[12];
                if (L >= lds_from_level) {
                    const f4 *src = tree_top + (at - lds_first);
#pragma unroll
                    for (int q = 0; q < 12; ++q) nd[q] = src[q];
                } else {
                    const RT_CONST_AS f4 *src = nodes + (size_t)at;
#pragma unroll
                    for (int q = 0; q < 12; ++q) nd[q] = src[q];
                }
                uint32_t live = 0u;               // children the ray may hit
                uint32_t nearest = 0xffffffffu;   // min over them of (entry distance | index)
#pragma unroll
                for (int pr = 0; pr < 4; ++pr) {
                    const f4 X = nd[3 * pr], Y = nd[3 * pr + 1], Z = nd[3 * pr + 2];   // lo0, lo1, hi0, hi1
                    const f2 ix2 = {ix, ix}, iy2 = {iy, iy}, iz2 = {iz, iz};
                    const f2 tlx = __builtin_elementwise_fma(X.xy, ix2, (f2){klx, klx}), thx = __builtin_elementwise_fma(X.zw, ix2, (f2){khx, khx});
                    const f2 tly = __builtin_elementwise_fma(Y.xy, iy2, (f2){kly, kly}), thy = __builtin_elementwise_fma(Y.zw, iy2, (f2){khy, khy});
                    const f2 tlz = __builtin_elementwise_fma(Z.xy, iz2, (f2){klz, klz}), thz = __builtin_elementwise_fma(Z.zw, iz2, (f2){khz, khz});
#pragma unroll
                    for (int c = 0; c < 2; ++c) {
                        const float ax_ = c ? tlx.y : tlx.x, bx_ = c ? thx.y : thx.x;
                        const float ay_ = c ? tly.y : tly.x, by_ = c ? thy.y : thy.x;
                        const float az_ = c ? tlz.y : tlz.x, bz_ = c ? thz.y : thz.x;
                        // [enter, leave] = [t_low, closest_so_far] cut down by the three slabs.  The
                        // median of a slab's two plane distances and a bound is that bound clamped
                        // into the slab: where the slab overlaps [enter, leave] both come out as
                        // max(enter, near) and min(leave, far); where it does not they meet in one
                        // point and stay together through the other slabs.  A box is kept only if
                        // enter < leave; an interval of length zero is at most a touch of the widened
                        // box, which no primitive inside reaches.
                        float enter = t_low, leave = best_t;
                        enter = __builtin_amdgcn_fmed3f(ax_, bx_, enter); leave = __builtin_amdgcn_fmed3f(ax_, bx_, leave);
                        enter = __builtin_amdgcn_fmed3f(ay_, by_, enter); leave = __builtin_amdgcn_fmed3f(ay_, by_, leave);
                        enter = __builtin_amdgcn_fmed3f(az_, bz_, enter); leave = __builtin_amdgcn_fmed3f(az_, bz_, leave);
                        // all ones where enter < leave
                        const uint32_t m = (uint32_t)((int32_t)__float_as_uint(enter - leave) >> 31);
                        const uint32_t child = (uint32_t)(2 * pr + c);
                        live |= m & (1u << child);
                        // enter >= 0 orders like its bit pattern; the low three bits carry the child
                        // index, a rejected child becomes all ones
                        const uint32_t key = (__float_as_uint(enter) & ~7u) | child;
                        nearest = min(nearest, (key & m) | ~m);
                    }
                }
#endif
                // children past the end of the level below do not exist
                const uint32_t have = min(8u, tbl.count[L] - 8u * st);
                uint32_t bits = live & ((1u << have) - 1u);
                if (L != 0u && bits != 0u) {
                    // descend into the surviving child the ray enters first (so that a near
                    // hit bounds the rest of the walk early), remember the others
                    uint32_t k = nearest & 7u;
                    if (((bits >> k) & 1u) == 0u) k = (uint32_t)__builtin_ctz(bits);   // nearest was a masked-out child
                    L -= 1u;
                    pend |= (unsigned long long)(bits & ~(1u << k)) << (8u * L);
                    st = 8u * st + k;
                } else {
                    // level 0: the exact test of the primitives this lane could not reject, in
                    // slot order
                    // (a NaN distance ends the walk at once: the next primitive of the step would
                    // replace it, and it need not be one that follows in the list)
                    while (bits != 0u) {
                        const uint32_t k = (uint32_t)__builtin_ctz(bits);
                        bits &= bits - 1u;
                        const uint32_t slot = run.first + 8u * st + k;
                        RTK_COUNT(sc, 4, 1);
                        RTK_COUNT_ANY(sc, 7, 1);
                        float t;
                        const bool closer = walk_test<KIND>(geom, by_slot, slot, ray, aux, t_min, entry_pos, best_t, best_slot, t);
                        best_t = closer ? t : best_t;
                        best_slot = closer ? slot : best_slot;
                        bits = best_t != best_t ? 0u : bits;
                    }
                    // next pending sibling, of this node or of its nearest ancestor that has one
                    const unsigned long long rest = pend >> (8u * L);
                    if (rest == 0ull || best_t != best_t) {
                        act = false;
                    } else {
                        const uint32_t z = (uint32_t)__builtin_ctzll(rest);
                        const uint32_t up = z >> 3, k = z & 7u;
                        L += up;
                        st = ((st >> (3u * up)) & ~7u) | k;
                        pend &= ~(1ull << (8u * L + k));
                    }
                }
            }
        }
    }
}

// ----------------------------------------------------------------------
// One RUN_TREE run for the lanes of a wave.
// The reference's result depends on the visiting order only while a hit
// distance is NaN (src/geometry/mod.rs:74: both comparisons fail and the
// sphere is accepted, whatever is held).  Rays that enter with such numbers
// (non-finite direction or one outside 1e-6 .. 1e6 in length, NaN closest_so_far
// inherited from an earlier run, huge origins, and for spheres any |d| (|o| + extent of the run) at which
// the discriminant could overflow) are kept out of the walk; a lane whose closest_so_far
// turns NaN (a primitive with NaN/overflowing coordinates) drops out of it.  Both kinds then
// take the run in list order, the second from the state it entered with.  Primitives that
// cannot be bounded are not in the tree: the run lists them, and every ray tests them first.
// ----------------------------------------------------------------------
template <int KIND>
__device__ __forceinline__ void scan_tree(const SceneView &sc, const RunView &run, uint32_t r, bool tracing,
                                          TreeTable &tbl, uint32_t &tbl_run, const f4 *tree_top, uint32_t lds_from_level,
                                          const Ray &ray, const RayAux &aux, float t_min, Hit &h, uint32_t &ff_shared) {
    const float big = 1e18f;
    const float a = aux.a;
    bool tame = a > 1e-12f && a < 1e12f && h.t == h.t &&
                __builtin_fabsf(ray.o.x) < big && __builtin_fabsf(ray.o.y) < big && __builtin_fabsf(ray.o.z) < big;
    if (KIND == DEV_SPHERE) {
        // Every intermediate of Sphere::hit stays below 2 a S^2 (S as in walk_tree).  Where that may
        // overflow, the discriminant can come out as inf - inf and the reference accepts the NaN root
        // of a sphere the ray passes far from, which no box can stand for.  (The triangle test cannot
        // accept such a distance: its gate compares the hit point with finite bounds.)
        // At the other end, below 1e-20 its products may underflow and the relative bound on the
        // discriminant that the pad of the boxes is made from no longer holds.  (Above it a sphere
        // with a |o - c|^2 < 1e-36 has the origin inside its padded box and is never skipped.)
        const float S = __builtin_amdgcn_sqrtf(len2(ray.o)) * 1.000001f + __uint_as_float(run.tail().x);
        const float aSS = a * (S * S);
        tame = tame && aSS > 1e-20f && aSS < 1e37f;
    }
    const bool walking = tracing && tame;
    bool listed = tracing && !tame;   // lanes that take the run in list order
    if (__builtin_amdgcn_ballot_w64(walking) != 0ull) {
        // the hit record stays as the ray came with it until the walk is over (a lane that has to
        // start the run over in list order starts from it)
        float best_t = h.t;
        uint32_t best_slot = NO_SLOT;
        // the run's primitives and their list positions, by slot
        const RT_CONST_AS f4 *geom = KIND == DEV_SPHERE ? sc.sph_geom() : sc.tris();
        const RT_CONST_AS uint32_t *by_slot = KIND == DEV_SPHERE ? sc.sph_id() : sc.tri_id();
        // primitives the tree does not hold (non-finite or huge coordinates): every ray tests them
        const uint32_t n_unbounded = run.tail().z;
        for (uint32_t k = 0; k < n_unbounded; ++k) {
            const uint32_t slot = run.first + sc.order()[run.order_base + run.count + k];
            if (walking && best_t == best_t) {
                float t;
                const bool closer = walk_test<KIND>(geom, by_slot, slot, ray, aux, t_min, h.id, best_t, best_slot, t);
                best_t = closer ? t : best_t;
                best_slot = closer ? slot : best_slot;
            }
        }
        walk_tree<KIND>(sc, run, r, walking && best_t == best_t, tbl, tbl_run, tree_top, lds_from_level, geom, by_slot,
                        ray, aux, t_min, h.id, best_t, best_slot);
        if (walking && best_t != best_t) {
            listed = true;
        } else if (walking && best_slot != NO_SLOT) {
            h.t = best_t; h.kind = KIND; h.slot = best_slot;
            h.id = (int32_t)by_slot[best_slot];
            // temp_hr.front_face as the last accepted primitive of the run left it: the run's closest
            if (sc.need_ff_shared())
                ff_shared = (KIND == DEV_SPHERE ? sphere_front_face(sc, best_slot, ray, best_t, aux.cheap)
                                                : tri_front_face(sc, best_slot, ray)) ? 1u : 0u;
        }
    }
    if (__builtin_amdgcn_ballot_w64(listed) != 0ull) {
        for (uint32_t k = 0; k < run.count; ++k) {
            const uint32_t slot = run.first + sc.order()[run.order_base + k];
            if (listed) {
                if (KIND == DEV_SPHERE) sphere_exact(sc, slot, (int32_t)(run.id0 + k), ray, aux, t_min, h, ff_shared);
                else tri_exact(sc, slot, (int32_t)(run.id0 + k), ray, t_min, h, ff_shared);
            }
        }
    }
}

// ----------------------------------------------------------------------
// HittableList::hit, src/geometry/mod.rs:240-258, without the scatter calls:
// scatter results of hits that do not end up closest are discarded by the
// reference, so only (t, primitive, front_face) is tracked here and the record
// of the final hit is rebuilt afterwards from the same inputs.
// ----------------------------------------------------------------------
template <int MODES>
__device__ __forceinline__ void scan_list(const SceneView &sc, const Ray &ray, const RayAux &aux, float t_min, bool tracing,
                                          TreeTable &tbl, uint32_t &tbl_run, const f4 *tree_top, Hit &h) {
    h.t = __builtin_inff();
    h.kind = -1;
    h.slot = 0;
    h.id = -1;
    h.ff = 1u;
    uint32_t ff_shared = 1u;   // temp_hr.front_face, default true (src/geometry/mod.rs:36-39,241)
    const float a = aux.a;
    Cull cull;
    bool have_cull = false;
    RTK_COUNT(sc, 1, 1);
    for (uint32_t r = 0; r < sc.n_runs; ++r) {
        const RunView run = load_run(sc, r);
        if (MODES != 0 && run.mode != RUN_EXACT) {
            // levels >= from are read from the LDS copy (only the cached run has any)
            const u4v in_lds = sc.lds();
            const uint32_t from = (r == in_lds.x) ? in_lds.y : 0xffffffffu;
            if ((MODES & 1) && run.mode == RUN_FLAT) {
                // the ball rejection test needs the unit direction and |o|^2
                if (!have_cull) { cull = cull_setup(ray, a); have_cull = true; }
                scan_flat(sc, run, cull, tracing, ray, aux, t_min, h, ff_shared);
            } else if ((MODES & 2) && run.kind == DEV_SPHERE) {
                scan_tree<DEV_SPHERE>(sc, run, r, tracing, tbl, tbl_run, tree_top, from, ray, aux, t_min, h, ff_shared);
            } else if (MODES & 2) {
                scan_tree<DEV_TRI>(sc, run, r, tracing, tbl, tbl_run, tree_top, from, ray, aux, t_min, h, ff_shared);
            }
        } else if (run.kind == DEV_SPHERE) {
            // two per trip, both records requested before the first is used: the loop's own scalar
            // instructions (13 beside 20 vector ones for a sphere that is missed, at a quarter of the
            // rate) and the wait for the record are shared
            uint32_t k = 0;
            const RT_CONST_AS f4 *geom = sc.sph_geom() + run.first;
            for (; k + 2u <= run.count; k += 2u) {
                const f4 g0 = geom[k], g1 = geom[k + 1u];
                sphere_exact(sc, g0, run.first + k, (int32_t)(run.id0 + k), ray, aux, t_min, h, ff_shared);
                sphere_exact(sc, g1, run.first + k + 1u, (int32_t)(run.id0 + k + 1u), ray, aux, t_min, h, ff_shared);
            }
            if (k < run.count)
                sphere_exact(sc, geom[k], run.first + k, (int32_t)(run.id0 + k), ray, aux, t_min, h, ff_shared);
        } else if (run.kind == DEV_PLANE) {
            // Plane::hit, src/geometry/mod.rs:112-130
            const RT_CONST_AS f4 *planes = sc.planes() + 2 * run.first;
            for (uint32_t k = 0; k < run.count; ++k) {
                const f4 n4 = planes[2 * k];
                const f4 p4 = planes[2 * k + 1];
                V3 n = mk(n4.x, n4.y, n4.z);
                float t = dot(mk(p4.x, p4.y, p4.z) - ray.o, n) / dot(n, ray.d);
                if (t < t_min || t > h.t) continue;
                h.t = t; h.kind = DEV_PLANE; h.slot = run.first + k; h.id = (int32_t)(run.id0 + k);
                if (sc.need_ff_shared()) ff_shared = dot(ray.d, n) < 0.0f ? 1u : 0u;
            }
        } else if (run.kind == DEV_CUBE) {
            // Cube::hit, src/geometry/mod.rs:158-223
            const RT_CONST_AS f4 *cubes = sc.cubes() + 4 * run.first;
            for (uint32_t k = 0; k < run.count; ++k) {
                const f4 c0 = cubes[4 * k + 2];
                const f4 c1 = cubes[4 * k + 3];
                float tmin, tmax, tymin, tymax, tzmin, tzmax;
                if (ray.d.x >= 0.0f) { tmin = (c0.x - ray.o.x) / ray.d.x; tmax = (c1.x - ray.o.x) / ray.d.x; }
                else                 { tmin = (c1.x - ray.o.x) / ray.d.x; tmax = (c0.x - ray.o.x) / ray.d.x; }
                if (ray.d.y >= 0.0f) { tymin = (c0.y - ray.o.y) / ray.d.y; tymax = (c1.y - ray.o.y) / ray.d.y; }
                else                 { tymin = (c1.y - ray.o.y) / ray.d.y; tymax = (c0.y - ray.o.y) / ray.d.y; }
                if ((tmin > tymax) || (tymin > tmax)) continue;
                if (tymin > tmin) tmin = tymin;
                if (tymax < tmax) tmax = tymax;
                if (ray.d.z >= 0.0f) { tzmin = (c0.z - ray.o.z) / ray.d.z; tzmax = (c1.z - ray.o.z) / ray.d.z; }
                else                 { tzmin = (c1.z - ray.o.z) / ray.d.z; tzmax = (c0.z - ray.o.z) / ray.d.z; }
                if ((tmin > tzmax) || (tzmin > tmax)) continue;
                if (tzmin > tmin) tmin = tzmin;
                if (tzmax < tmax) tmax = tzmax;
                if (tmin < h.t && tmax > t_min) {
                    h.t = tmin; h.kind = DEV_CUBE; h.slot = run.first + k; h.id = (int32_t)(run.id0 + k);
                    h.ff = ff_shared;   // no set_face_normal in Cube::hit
                }
            }
        } else {
            for (uint32_t k = 0; k < run.count; ++k)
                tri_exact(sc, run.first + k, (int32_t)(run.id0 + k), ray, t_min, h, ff_shared);
        }
    }
}

// Rebuild the HitRecord of the closest hit (p, normal, front_face) exactly as
// the primitive's hit() filled it, then run its material's scatter.
// Returns the material index; ray is replaced by the scattered ray.
// (path_key() is asked for only where a Dielectric draws its coin)
template <class PathKey>
__device__ __forceinline__ uint32_t shade_hit(const SceneView &sc, const float *tbl_sphere, Ray &ray, V3 ud, const Hit &h,
                                              bool cheap, float sp_bn, uint32_t &path_counter, PathKey path_key,
                                              uint32_t bounce) {
    V3 p = ray.o + ray.d * h.t;   // Ray::at, src/ray/mod.rs:19-21
    V3 normal;
    bool ff = h.ff != 0u;
    uint32_t mat;
    if (h.kind == DEV_SPHERE) {
        // src/geometry/mod.rs:81-86
        const f4 g = sc.sph_geom()[h.slot];
        V3 outward = div3(p - mk(g.x, g.y, g.z), g.w, cheap);   // |p - c| < 2^60 by the bounds `cheap` stands for
        ff = dot(ray.d, outward) < 0.0f;
        normal = ff ? outward : -outward;
        mat = sc.sph_mat()[h.slot];
    } else if (h.kind == DEV_PLANE) {
        // src/geometry/mod.rs:120-123
        const f4 n4 = sc.planes()[2 * h.slot];
        V3 outward = mk(n4.x, n4.y, n4.z);
        ff = dot(ray.d, outward) < 0.0f;
        normal = ff ? outward : -outward;
        mat = __float_as_uint(n4.w);
    } else if (h.kind == DEV_CUBE) {
        // src/geometry/mod.rs:208-216
        const RT_CONST_AS f4 *cube = sc.cubes() + 4 * h.slot;
        const f4 cc = cube[0];
        const f4 hw = cube[1];   // w/2, h/2, d/2
        const float eps = 1.0001f;
        V3 pc = p - mk(cc.x, cc.y, cc.z);
        V3 nn;
        nn.x = trunc_i32_as_f32(eps * pc.x / hw.x);
        nn.y = trunc_i32_as_f32(eps * pc.y / hw.y);
        nn.z = trunc_i32_as_f32(eps * pc.z / hw.z);
        // unit_vector(nn): a hit on a face (not within 1e-4 of an edge) has one component of +-1 and two
        // zeros (+0: they come from integers): length 1, and x / 1 = x
        if (__builtin_amdgcn_ballot_w64(len2(nn) != 1.0f) == 0ull) normal = nn;
        else normal = unit(nn);
        mat = __float_as_uint(cc.w);
    } else {
        const RT_CONST_AS f4 *tri = sc.tris() + 3 * h.slot;
        const f4 v0 = tri[0];
        const f4 e1 = tri[1];
        const f4 e2 = tri[2];
        V3 outward = unit(cross(mk(e1.x, e1.y, e1.z), mk(e2.x, e2.y, e2.z)));
        ff = dot(ray.d, outward) < 0.0f;
        normal = ff ? outward : -outward;
        mat = __float_as_uint(v0.w);
    }
    const RT_CONST_AS f4 *material = sc.mat_rec() + 2u * mat;
    const f4 ma = material[0];   // r, g, b, param
    const f4 mb = material[1];   // kind, 1/param, r0 for a front face, r0 for a back face
    const int mk_ = (int)__float_as_uint(mb.x);
    V3 dir;
    // `ud` is unit_vector(r_in.direction) (src/geometry/mod.rs:276,326), the same value for
    // whichever material needs it; the unit-sphere draw of Metal and Lambertian is one lookup
    V3 rv = mk(0.0f, 0.0f, 0.0f);
    if (mk_ != RT_DIELECTRIC) {
        path_counter += 1;
        rv = rius(tbl_sphere, path_counter, sp_bn);
    }
    if (mk_ == RT_METAL) {
        // Metal::scatter, src/geometry/mod.rs:274-281
        V3 reflected = reflect(ud, normal);
        dir = reflected + rv * ma.w;
    } else if (mk_ == RT_LAMBERTIAN) {
        // Lambertian::scatter, src/geometry/mod.rs:287-299
        dir = normal + rv;
        if (dir.x < 0.0f && dir.y < 0.0f && dir.z < 0.0f) dir = normal;   // is_near_zero, src/vector/mod.rs:29-32
    } else {
        // Dielectric::scatter, src/geometry/mod.rs:323-346
        float ratio = ff ? mb.y : ma.w;   // 1.0 / ir or ir
        float cos_theta = rmin(dot(-ud, normal), 1.0f);
        float sin_theta = sqrt_mostly(1.0f - cos_theta * cos_theta);
        bool refl = ratio * sin_theta > 1.0f;
        if (!refl) {
            // reflectance, src/geometry/mod.rs:313-317 (r0 = ((1-ratio)/(1+ratio))^2 comes with the
            // material); coin: hash of (path, bounce)
            float r0 = ff ? mb.z : mb.w;
            float refl_p = r0 + (1.0f - r0) * powi5(1.0f - cos_theta);
            uint32_t rb = mix32(path_key() + GOLDEN32 * (bounce + 1u));
            float coin = (float)(rb >> 8) * (1.0f / 16777216.0f);
            refl = refl_p > coin;
        }
        if (refl) {
            dir = reflect(ud, normal);
        } else {
            // refract, src/geometry/mod.rs:307-312
            float ct = rmin(dot(-ud, normal), 1.0f);
            V3 perp = (ud + normal * ct) * ratio;
            V3 par = (-normal) * sqrt_mostly(__builtin_fabsf(1.0f - len2(perp)));
            dir = perp + par;
        }
    }
    ray.o = p;
    ray.d = dir;
    return mat;
}

// sky branch of ray_colour, src/raytracer.rs:81-91
// (for a unit direction the three channels lie in 0.25 .. 1, far inside sqrt_refined's range)
__device__ __forceinline__ V3 sky(V3 ud, bool gamma) {
    float t = 0.5f * ud.y + 1.0f;
    V3 c = mk(1.0f, 1.0f, 1.0f) * (1.0f - t) + mk(0.5f, 0.7f, 1.0f) * t;
    if (gamma) {
        const float low = __builtin_fminf(__builtin_fminf(c.x, c.y), c.z);
        if (RTK_SHORT_ARITH && __builtin_amdgcn_ballot_w64(!(low >= SQRT_ARG_MIN)) == 0ull) {
            c.x = sqrt_refined(c.x); c.y = sqrt_refined(c.y); c.z = sqrt_refined(c.z);
        } else {
            c.x = __builtin_sqrtf(c.x); c.y = __builtin_sqrtf(c.y); c.z = __builtin_sqrtf(c.z);
        }
    }
    return c;
}

// ----------------------------------------------------------------------
// calculate_some_pxls + ray_colour, src/raytracer.rs:232-271 and :68-92.
//
// The launch is a fixed number of resident workgroups; every wavefront on its own takes work
// units from one global queue until the queue is empty.  A unit is 1 to 64 pixels of an 8x8 tile
// (part of a row, or whole rows; chosen per launch, see rtk_launch_render) x all samples of the
// launch; its work items
// (pixel, sample), sample-major, are handed to the wave's lanes one at a time: a lane whose
// path ends takes the next item, whatever pixel it belongs to, and when a unit runs out of
// items the wave starts the next unit at once while its longer paths are still on the way (up
// to four units in flight per wave, UnitRing).  Pixels differ a lot in segments per sample;
// with one lane per pixel a wave would wait for its slowest pixel.
//
// The colour of every sample goes to the wave's part of the staging ring (fr.stage).  When the last item of a unit is finished
// the wave adds the unit's samples to the pixel sums IN SAMPLE ORDER: `px_colour += ...`
// (src/raytracer.rs:262) is an ordered f32 sum and must not depend on which lane finished
// first.  Nothing is shared between waves.
// ----------------------------------------------------------------------
constexpr uint32_t ITEM_BITS = DEV_STAGE_ITEM_BITS;   // item index inside a unit: pixels of the unit x samples of the launch <= 512

// the arguments of render_tiles as they lie in the kernel's argument block
// (cams, n_frames: a launch over several frames of one size, frame f seen through cams[f], its per-pixel
// buffers frame f of fr.accum / fr.primary_ids - only the kernels compiled with FRAMES read them)
struct KernArgs {
    DevScene dsc; rt_camera cam; rt_params p; DevFrame fr; uint32_t n_bands, unit_px_log2, stretch_unit, fetch_min;
    const rt_camera *cams; uint32_t n_frames;
    uint32_t unit_batch;   // consecutive units a wave takes by position at first, and per request to the counter
};

// FRAMES: the launch covers n_frames frames of one size with a camera each (the reference renders ten
// in a row, moving the camera in between, src/raytracer.rs:32,165,190): the queue's units run over the
// tiles of all of them, so that frames too small to fill the chip - the reference's own 600 x 337 at one
// sample is 3,192 tiles for 8,192 wavefronts - share a launch instead of paying its start and its
// uneven end once each.
template <int MODES, int DEPTH, bool FRAMES = false>
// Two 8-wave workgroups per CU: four waves per SIMD, at most 128 VGPRs each (some per-item
// state is spilled around the walk).  Measured alternatives on the 10,000-sphere frame: three
// waves per SIMD without spills (one 12-wave workgroup per CU) -14 %, 4-wave workgroups with
// a 16 KB tree copy -5 %, 16-wave workgroups the same, 96 VGPRs for five waves per SIMD -40 %;
// 6- or 10-wave workgroups fit only once per CU.
__global__ __launch_bounds__(waves_of(MODES) * 64, occ_hint_of(MODES))
void render_tiles(KernArgs by_value) {
    __shared__ Lds<MODES, DEPTH> lds;
#ifdef RTK_WAVE_ENDS
    const unsigned long long t_begin = wall_clock64();
#endif
    // The arguments (90 dwords) are not kept in registers: with the loop's own wave-uniform
    // state they would overflow the scalar register file, and the compiler's way out, parking
    // them in lanes of a vector register, costs a vector instruction per use (a quarter of the
    // vector instructions of a small scene).  Each part of the loop reads what it needs from the
    // argument block in memory instead (scalar loads from a line that stays in the scalar cache),
    // through a pointer the compiler cannot trace back (args()), so that the loads stay where
    // they are written.
    const RT_CONST_AS KernArgs *const args0 = (const RT_CONST_AS KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    auto args = [&]() { const RT_CONST_AS KernArgs *k = args0; asm volatile("" : "+s"(k)); return k; };
    // (the wave's number as a scalar: what is indexed by it - its part of the staging ring, its
    // rows in LDS - then has wave-uniform addresses that need no vector registers of their own)
    const uint32_t wave = (uint32_t)__builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), lane = threadIdx.x & 63u;
    {
        const RT_CONST_AS KernArgs *k = args0;
        for (uint32_t q = threadIdx.x; q < 128; q += blockDim.x) lds.disc[q] = k->fr.tbl_disc[q];
        for (uint32_t q = threadIdx.x; q < 768; q += blockDim.x) lds.sphere[q] = k->fr.tbl_sphere[q];
        if (lane < 4u) lds.ring[wave].left[lane] = 0u;
        if (threadIdx.x == 0u) {
            // width - 1, height - 1 and their refined reciprocals, for start_sample
            const float w1 = (float)(k->p.width - 1u), h1 = (float)(k->p.height - 1u);
            lds.uv[0] = w1; lds.uv[1] = h1; lds.uv[2] = rcp_refined(w1); lds.uv[3] = rcp_refined(h1);
        }
        if (blockIdx.x == 0u && threadIdx.x == 0u) *k->fr.queue_next = 0u;     // the next launch starts at zero
        if (MODES & 2) {
            const RT_CONST_AS f4 *src = const_ptr<f4>(k->dsc.tree_rec) + (size_t)k->dsc.lds_first;
            for (uint32_t q = threadIdx.x; q < k->dsc.lds_count; q += blockDim.x) lds.tree_top[q] = src[q];
        }
    }
    __syncthreads();

    // what every part of the loop uses stays in registers
    const uint32_t W = args0->p.width, H = args0->p.height;
    const uint32_t unit_px_log2 = args0->unit_px_log2, sample_begin = args0->p.sample_begin;
    const uint32_t max_bounces = args0->p.max_bounces;
    UnitRing &ring = lds.ring[wave];
    // this wave's part of the staging ring: DEV_STAGE_SLOTS units of up to 2^ITEM_BITS colours;
    // an item's place is its own number, ring slot << ITEM_BITS | index in the unit
    StagedRgb *const stage = args0->fr.stage + ((size_t)(blockIdx.x * waves_of(MODES) + wave) << (ITEM_BITS + 2u));

    // units of this launch: tile by tile along the band, the parts of a tile one after the other
    const uint32_t unit_px = 1u << unit_px_log2;                  // 1 .. 64 pixels: part of a tile row, or whole rows
    const uint32_t unit_cols = unit_px < 8u ? unit_px : 8u, unit_rows = unit_px < 8u ? 1u : unit_px >> 3;

    // Per-lane state that lives across the scan is kept small (registers are what limits the
    // kernel): the work item, the ray, the depth budget and the unit-sphere counter.  Pixel,
    // sample, screen-space noise value and the path's hash key are derived from the item
    // again where they are needed.
    uint32_t item = 0;        // ring slot << ITEM_BITS | pixel of the unit + pixels of a unit * (sample - sample_begin)
    bool alive = false;
    Ray ray;
    ray.o = mk(0, 0, 0); ray.d = mk(0, 0, 1);
    uint32_t depth_left = max_bounces, path_counter = 0;
    float sp_bn = 0.0f;
    uint32_t n_seg = 0;
    // wave-uniform: the unit items are being handed out from
    uint32_t cur_slot = 3u, cur_next = 0u, cur_n = 0u;
    bool queue_empty = false;
    // the queue is read one unit ahead, so that the answer is there when it is needed
    // (a wave's first unit is the one with its own number: thousands of waves asking the counter
    // in the launch's first microsecond would queue up behind one another - the counter hands out
    // the units after those)
    // Units come in batches of unit_batch consecutive ones (1 unless the launch's units are short: at one
    // sample per pixel a unit, a whole tile, is 64 items instead of 512, and single units of that size
    // would be asked for faster than the counter answers, about 90 requests per microsecond: ten frames
    // of 600 x 337 at one sample took 0.50 ms that way and 0.27 ms of it was the counter).  Only the
    // kernels that hand out stretches at all (those with a tree, and FRAMES) do this.
    constexpr bool STRETCHES = (MODES & 2) != 0 || FRAMES;
    const uint32_t unit_batch = STRETCHES ? args0->unit_batch : 1u;
    const uint32_t grid_waves = gridDim.x * (uint32_t)waves_of(MODES);
    uint32_t unit_ahead = (blockIdx.x * (uint32_t)waves_of(MODES) + wave) * unit_batch;
    // units the wave may still take from the stretch it was given last (wave-uniform)
    uint32_t stretch_next = 0u, stretch_end = 0u, asked = unit_batch;   // asked: units of the request on the way

    struct Where { uint32_t i, j, s, f; };      // column, row from the bottom, sample, frame of the launch
    auto where = [&](uint32_t it) {
        const uint32_t slot = it >> ITEM_BITS, q = it & ((1u << ITEM_BITS) - 1u);
        const uint32_t xy = ring.xy[slot];
        Where w;
        w.i = (xy & 0xffffu) + (q & (unit_cols - 1u));
        w.j = (xy >> 16) + ((q >> 3) & (unit_rows - 1u));
        w.s = sample_begin + (q >> unit_px_log2);
        w.f = FRAMES ? ring.fr[slot] : 0u;
        return w;
    };
    const size_t frame_px = FRAMES ? (size_t)W * H : 0u;    // pixels between a frame's buffers and the next one's
    auto path_key_at = [&](const Where &w, uint32_t seed) { return mix32((w.j * W + w.i) ^ mix32(w.s ^ (seed * GOLDEN32))); };

    auto start_sample = [&](const Where &w) {
        const RT_CONST_AS KernArgs *k = args();
        // two disc draws per sample: jitter then lens (src/raytracer.rs:254,258);
        // sample n of a single-thread run uses entries 2n+1 and 2n+2 of the
        // pre-incremented counter (src/denoising/mod.rs:13-17)
        const uint32_t pix = w.j * W + w.i;
        uint32_t n = pix * k->p.spp + w.s;   // only n mod 32 matters, wrap is harmless
        uint32_t k1 = (2u * n + 1u) & 63u, k2 = (2u * n + 2u) & 63u;
        float ox = lds.disc[2 * k1], oy = lds.disc[2 * k1 + 1];
        float la = lds.disc[2 * k2], lb = lds.disc[2 * k2 + 1];
        // (column + disc entry) / (W - 1), src/raytracer.rs:255-256: the denominators are 1 .. 65535 and the
        // numerators, for a table without tiny entries, zero-free and in 2^-90 .. 2^17
        const bool cheap_uv = RTK_SHORT_ARITH && k->fr.disc_plain != 0u && W >= 2u && H >= 2u;
        const float w1 = lds.uv[0], h1 = lds.uv[1], inv_w1 = lds.uv[2], inv_h1 = lds.uv[3];
        float u, v;
        if (cheap_uv) { u = div_refined((float)w.i + ox, w1, inv_w1); v = div_refined((float)w.j + oy, h1, inv_h1); }
        else { u = ((float)w.i + ox) / w1; v = ((float)w.j + oy) / h1; }
        if (FRAMES && k->n_frames > 1u) ray = get_ray(k->cams[w.f], u, v, la, lb);
        else ray = get_ray(k->cam, u, v, la, lb);
        path_counter = path_key_at(w, k->p.seed);
        // BlueNoise::get_screenspace, src/denoising/mod.rs:19-23: fixed per pixel, kept for the path
        sp_bn = k->fr.tbl_screen[128u * (w.i % 128u) + (w.j % 128u)];
        depth_left = max_bounces;
    };

    // All items of the unit in `slot` are finished and their colours stored (by lanes of
    // this wave): its pixels are added up, one pixel per lane.
    // Ring slots whose items are all finished and stored but not added up yet (bit = slot,
    // wave-uniform).  They are added up together, as late as possible: when one of them is needed
    // for a new unit, or at the end.  One unit alone keeps as many lanes busy as it has pixels (8 of
    // 64 at 64 samples per launch) for as many dependent additions as the launch has samples.
    uint32_t unsummed = 0u;
    // (RTK_NO_STAGE, a measurement build: colours are neither staged nor read back, the sums stay zero -
    // what the staging traffic costs is the difference in time)
    auto staged = [&](uint32_t it) {
#ifdef RTK_NO_STAGE
        StagedRgb c; c.r = c.g = c.b = __uint_as_float(it & 0u); return c;
#else
        return stage[it];
#endif
    };
    auto finish_units = [&](uint32_t slots) {
        const RT_CONST_AS KernArgs *k = args();
        float4 *const accum = k->fr.accum;
        const uint32_t sample_count = k->p.sample_count;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // the slots in ascending order
        uint32_t rest = slots;
        const uint32_t n_slots = (uint32_t)__builtin_popcount(slots);
        const uint32_t s0 = (uint32_t)__builtin_ctz(rest | 16u); rest &= rest - 1u;
        const uint32_t s1 = (uint32_t)__builtin_ctz(rest | 16u); rest &= rest - 1u;
        const uint32_t s2 = (uint32_t)__builtin_ctz(rest | 16u); rest &= rest - 1u;
        const uint32_t s3 = (uint32_t)__builtin_ctz(rest | 16u);
        const uint32_t per_pass = 64u >> unit_px_log2;            // units the lanes of the wave cover at once
        for (uint32_t first = 0; first < n_slots; first += per_pass) {
            const uint32_t u = first + (lane >> unit_px_log2);    // this lane's unit among them,
            const uint32_t q = lane & (unit_px - 1u);             // and its pixel of the unit (see where()): column q % 8, row q / 8
            const uint32_t slot = u == 0u ? s0 : u == 1u ? s1 : u == 2u ? s2 : s3;
            const uint32_t xy = ring.xy[slot & 3u];
            const uint32_t pi = (xy & 0xffffu) + (q & 7u), pj = (xy >> 16) + (q >> 3);
            if (u < n_slots && pi < W && pj < H) {
                size_t pix_out = (size_t)(H - 1u - pj) * W + pi;   // src/raytracer.rs:265
                if (FRAMES) pix_out += frame_px * ring.fr[slot & 3u];
                float4 acc = accum[pix_out];
                V3 px = mk(acc.x, acc.y, acc.z);
                // with a depth budget of 0 every sample is black and nothing was traced
                const uint32_t n_add = max_bounces > 0 ? sample_count : 0u;
                // sample n of pixel q of the unit in `slot` is item q + pixels of the unit * n of that slot
                const uint32_t item0 = (slot << ITEM_BITS) + q;
                uint32_t n = 0;
                // sum_batch_of(MODES) loads on the way at a time; the additions keep their order (src/raytracer.rs:262)
                for (; n + sum_batch_of(MODES) <= n_add; n += sum_batch_of(MODES)) {
                    StagedRgb c[sum_batch_of(MODES)];
#pragma unroll
                    for (uint32_t m = 0; m < sum_batch_of(MODES); ++m) c[m] = staged(item0 + ((n + m) << unit_px_log2));
#pragma unroll
                    for (uint32_t m = 0; m < sum_batch_of(MODES); ++m) px = px + mk(c[m].r, c[m].g, c[m].b);
                }
                for (; n < n_add; ++n) {
                    const StagedRgb c = staged(item0 + (n << unit_px_log2));
                    px = px + mk(c.r, c.g, c.b);
                }
                accum[pix_out] = make_float4(px.x, px.y, px.z, acc.w + (float)sample_count);
            }
        }
    };

    // lanes with `need` take the next items; when the current unit has none left the wave
    // takes the next unit from the queue.  Lanes left without an item try again after the
    // next segment (a ring slot may still be held by a unit with paths on the way).
    auto fetch = [&](bool need) {
        for (;;) {
            const unsigned long long mask = __builtin_amdgcn_ballot_w64(need);
            if (mask == 0ull) break;
            if (cur_next >= cur_n) {
                if (queue_empty) break;
                const uint32_t slot = (cur_slot + 1u) & 3u;
                if (ring.left[slot] != 0u) break;
                if (unsummed & (1u << slot)) { finish_units(unsummed); unsummed = 0u; }
                const RT_CONST_AS KernArgs *k = args();
                const uint32_t tiles_x = (W + TILE - 1u) / TILE;
                const uint32_t sub_log2 = 6u - unit_px_log2;                  // units per tile, log2
                const uint32_t n_units = (tiles_x * k->n_bands * (FRAMES ? k->n_frames : 1u)) << sub_log2;
                const uint32_t sample_count = k->p.sample_count;
                uint32_t u;
                // (only the kernels with a tree, and the one for short units or several frames: a scene
                // without a tree has short launches, and the three wave-uniform values cost its
                // kernel more than the requests they save)
                if (STRETCHES && stretch_next < stretch_end) {
                    u = stretch_next++;
                } else {
                    u = (uint32_t)__builtin_amdgcn_readfirstlane((int)unit_ahead);
                    if (u >= n_units) { queue_empty = true; break; }
                    // The counter hands out stretches of consecutive units, one request per stretch (it
                    // answers about 90 requests per microsecond; with one unit per request a million
                    // triangles at 4K ask 21 times per microsecond): a few units while much is left,
                    // single units afterwards (see rtk_launch_render).  A wave's first unit, by
                    // position, is a stretch of one.
                    if (STRETCHES) { stretch_next = u + 1u; stretch_end = min(u + asked, n_units); }
                    if (MODES & 2) {
                        const uint32_t left = n_units - u, per = k->stretch_unit;   // per: 32 batches for every wave
                        asked = (left >= 8u * per ? 8u : left >= 4u * per ? 4u : left >= 2u * per ? 2u : 1u) * unit_batch;
                    }
                    // (a launch with no more batches than waves never asks the counter: thousands of
                    // requests for nothing take 30 us to answer)
                    if (grid_waves * unit_batch >= n_units) unit_ahead = n_units;
                    else if (lane == 0u) unit_ahead = grid_waves * unit_batch + atomicAdd(k->fr.queue, asked);
                }
                const uint32_t t = u >> sub_log2, sub = u & ((1u << sub_log2) - 1u);
                uint32_t by = t / tiles_x;
                const uint32_t tx = t - by * tiles_x;
                uint32_t frame = 0u;
                if (FRAMES) { frame = by / k->n_bands; by -= frame * k->n_bands; }   // frame after frame, each band by band
                const uint32_t band = k->p.band_begin + by * k->p.band_stride;
                const uint32_t first = sub << unit_px_log2;                 // the unit's first pixel of the tile, row-major
                const uint32_t x0 = tx * TILE + (first & 7u), y0 = band * TILE + (first >> 3);
                const uint32_t vw = x0 >= W ? 0u : (W - x0 < unit_cols ? W - x0 : unit_cols);
                const uint32_t vh = y0 >= H ? 0u : (H - y0 < unit_rows ? H - y0 : unit_rows);
                const uint32_t left = max_bounces > 0u ? vw * vh * sample_count : 0u;
                if (lane == 0u) { ring.xy[slot] = x0 | (y0 << 16); ring.left[slot] = left; if (FRAMES) ring.fr[slot] = frame; }
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                cur_slot = slot; cur_next = 0u; cur_n = max_bounces > 0u ? sample_count << unit_px_log2 : 0u;
                if (left == 0u) { unsummed |= 1u << slot; continue; }
            }
            const uint32_t rank = __builtin_amdgcn_mbcnt_hi((uint32_t)(mask >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)mask, 0u));
            const uint32_t k = cur_next + rank;
            cur_next += (uint32_t)__builtin_popcountll(mask);
            if (need && k < cur_n) {
                const uint32_t it = (cur_slot << ITEM_BITS) | k;
                const Where w = where(it);
                if (w.i < W && w.j < H) { item = it; start_sample(w); need = false; alive = true; }
            }
        }
    };
#ifdef RTK_TIMES
    unsigned long long tm_scan = 0, tm_rest = 0, tm_done = 0, tm_fetch = 0, tm_fin = 0;
#endif
    fetch(true);
    uint32_t tbl_run = 0xffffffffu;   // the run whose per-level constants lds.tree[wave] holds (wave-uniform)
    bool parked = false;   // the lane's path is over, its colour not worked out and its item not counted yet

    while (__builtin_amdgcn_ballot_w64(alive) != 0ull) {
        Hit h;
#ifdef RTK_TIMES
        const unsigned long long tm0 = __builtin_readcyclecounter();
#endif
        const RT_CONST_AS KernArgs *k = args();
        const SceneView sc = view_of(&k->dsc, k->fr.segments, MODES != 0);
        const float t_min = k->p.t_min;
        // |direction|^2 and whether the short division and root sequences apply (see RayAux)
        const bool cheap_frame = RTK_SHORT_ARITH && k->dsc.cheap_ok != 0u && t_min >= 0x1p-20f;
        RayAux aux;
        aux.a = len2(ray.d);
        {
            const float far = __builtin_fmaxf(__builtin_fmaxf(__builtin_fabsf(ray.o.x), __builtin_fabsf(ray.o.y)), __builtin_fabsf(ray.o.z));
            const bool moderate = aux.a >= 0x1p-24f && aux.a <= 0x1p24f && far <= 0x1p30f;
            aux.cheap = cheap_frame && __builtin_amdgcn_ballot_w64(alive && !moderate) == 0ull;
            aux.inv_a = rcp_refined(aux.a);
        }
        scan_list<MODES>(sc, ray, aux, t_min, alive, lds.tree[wave], tbl_run, lds.tree_top, h);
#ifdef RTK_TIMES
        const unsigned long long tm1 = __builtin_readcyclecounter();
        tm_scan += tm1 - tm0;
#endif
        n_seg += (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(alive));
        bool done = false;
        if (alive) {
            // (the arguments the shading needs are read after the scan, not kept across it)
            const RT_CONST_AS KernArgs *ks = args();
            const SceneView sc = view_of(&ks->dsc, ks->fr.segments, MODES != 0);
            const uint32_t n_stack = max_bounces - depth_left;   // hits of this path so far
            if (n_stack == 0u) {
                const Where at = where(item);
                if (at.s == 0u) ks->fr.primary_ids[frame_px * at.f + (size_t)(H - 1u - at.j) * W + at.i] = h.id;   // src/raytracer.rs:265
            }
            // unit_vector(direction): the sky (src/raytracer.rs:82), Metal and Dielectric all start
            // from it; computed once for the wave instead of once per branch
            // (cheap: 2^-12 <= |d| <= 2^12, every component at most that)
            const V3 ud = div3(ray.d, aux.cheap ? sqrt_refined(aux.a) : __builtin_sqrtf(aux.a), aux.cheap);
            uint32_t n_hits = n_stack;
            if (h.kind >= 0) {
                uint32_t mat = shade_hit(sc, lds.sphere, ray, ud, h, aux.cheap, sp_bn, path_counter,
                                         [&]() { return path_key_at(where(item), ks->p.seed); }, n_stack);
                lds.mat_stack[wave][n_stack][lane] = mat;
                n_hits = n_stack + 1u;
                depth_left -= 1;
                done = depth_left == 0u;   // `if ray_bounces <= 0 { return 0 }`, src/raytracer.rs:69
            } else {
                done = true;
            }
#ifdef RTK_TIMES
            const unsigned long long tm2 = __builtin_readcyclecounter();
#endif
            if (done) {
                // the path is over; its colour is worked out with those of the other lanes that stand
                // idle by the time the wave hands out new items (below).  Until then the ray's origin
                // holds the unit direction it left the scene with and the depth budget the number of
                // its hits (top bit: it did leave the scene; else it ran out of depth and is black).
                ray.o = ud;
                depth_left = n_hits | (h.kind >= 0 ? 0u : 0x80000000u);
                alive = false;
            }
#ifdef RTK_TIMES
            tm_done += __builtin_readcyclecounter() - tm2;
#endif
        }
#ifdef RTK_TIMES
        const unsigned long long tm3 = __builtin_readcyclecounter();
#endif
        // Finished items are counted against their units, and their lanes given new items, once
        // fetch_min lanes stand idle (or all that are left): both steps cost the wave the
        // same whether one lane takes part or all.
        parked = parked || done;
        const unsigned long long idle = __builtin_amdgcn_ballot_w64(!alive);
        const uint32_t fetch_min = MODES == 0 ? (uint32_t)RTK_FETCH_MIN0 : args()->fetch_min;   // chosen per launch, see rtk_launch_render
        const bool refill = idle != 0ull && ((uint32_t)__builtin_popcountll(idle) >= fetch_min || (idle | ~__builtin_amdgcn_ballot_w64(true)) == ~0ull);
        if (refill && parked) {
            // attenuation * ray_colour(...), innermost product first (src/raytracer.rs:79)
            // (four entries per round: their fetches overlap, the products keep their order)
            const RT_CONST_AS KernArgs *ks = args();
            const RT_CONST_AS f4 *mat_rec = const_ptr<f4>(ks->dsc.mat_rec);
            // the sky (src/raytracer.rs:82-91), or black
            V3 c = mk(0.0f, 0.0f, 0.0f);
            if (depth_left >> 31) c = sky(ray.o, ks->p.gamma != 0u);
            uint32_t k = depth_left & 0x7fffffffu;
            while (k != 0u) {
                const uint32_t m0 = lds.mat_stack[wave][k - 1u][lane];
                const uint32_t m1 = lds.mat_stack[wave][k >= 2u ? k - 2u : 0u][lane];
                const uint32_t m2 = lds.mat_stack[wave][k >= 3u ? k - 3u : 0u][lane];
                const uint32_t m3 = lds.mat_stack[wave][k >= 4u ? k - 4u : 0u][lane];
                const f4 a0 = mat_rec[2u * m0], a1 = mat_rec[2u * m1], a2 = mat_rec[2u * m2], a3 = mat_rec[2u * m3];
                c = mk(a0.x, a0.y, a0.z) * c;
                if (k >= 2u) c = mk(a1.x, a1.y, a1.z) * c;
                if (k >= 3u) c = mk(a2.x, a2.y, a2.z) * c;
                if (k >= 4u) c = mk(a3.x, a3.y, a3.z) * c;
                k = k >= 4u ? k - 4u : 0u;
            }
            StagedRgb out; out.r = c.x; out.g = c.y; out.b = c.z;
#ifdef RTK_NO_STAGE
            asm volatile("" ::"v"(out.r), "v"(out.g), "v"(out.b));
#else
            stage[item] = out;
#endif
        }
        // (at most four different units in a wave)
        unsigned long long dm = refill ? __builtin_amdgcn_ballot_w64(parked) : 0ull;
        while (dm != 0ull) {
            const uint32_t slot = (uint32_t)__builtin_amdgcn_readlane((int)(item >> ITEM_BITS), __builtin_ctzll(dm));
            const unsigned long long same = dm & __builtin_amdgcn_ballot_w64((item >> ITEM_BITS) == slot);
            dm &= ~same;
            const uint32_t left = ring.left[slot] - (uint32_t)__builtin_popcountll(same);
            if (lane == 0u) ring.left[slot] = left;
            __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
            if (left == 0u) unsummed |= 1u << slot;
        }
#ifdef RTK_TIMES
        const unsigned long long tm4 = __builtin_readcyclecounter();
        tm_fin += tm4 - tm3;
#endif
        if (refill) { parked = false; fetch(!alive); }
#ifdef RTK_TIMES
        tm_fetch += __builtin_readcyclecounter() - tm4;
        tm_rest += __builtin_readcyclecounter() - tm1;
#endif
    }
    if (unsummed != 0u) finish_units(unsummed);
#ifdef RTK_TIMES
    // development aid: wave time inside the closest-hit scan and outside it
    if (lane == 0) {
        unsigned long long *const seg = args0->fr.segments;
        atomicAdd(seg + 4, tm_scan); atomicAdd(seg + 7, tm_rest);
        atomicAdd(seg + 5, tm_done); atomicAdd(seg + 6, tm_fetch);
        atomicAdd(seg + 3, tm_fin);
    }
#endif
#ifdef RTK_WAVE_ENDS
    // development aid (tools/wave_ends.py): when each wave started and ended, in ticks of the
    // 100 MHz clock, at the start of the wave's part of the staging ring (free by now)
    if (lane == 0) {
        unsigned long long *log = (unsigned long long *)stage;
        log[0] = t_begin | ((unsigned long long)n_seg << 40); log[1] = wall_clock64();
    }
#endif
    if (lane == 0 && n_seg != 0) atomicAdd(args0->fr.segments, (unsigned long long)n_seg);
}

// write_colour + from_u8_rgb, src/raytracer.rs:35-55
__device__ __forceinline__ uint32_t quantise(float x) {
    if (x < 0.0f) x = 0.0f;
    else if (x > 0.999f) x = 0.999f;
    float y = 256.0f * x;
    if (y != y) return 0u;          // Rust `as u8`: NaN -> 0, saturating
    if (y >= 255.0f) return 255u;
    if (y <= 0.0f) return 0u;
    return (uint32_t)(int32_t)y;
}
__global__ __launch_bounds__(256)
void pack_0rgb(const float4 *__restrict__ accum, uint32_t *__restrict__ out, uint32_t n_px) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n_px) return;
    float4 a = accum[k];
    float scale = 1.0f / a.w;
    uint32_t r = quantise(a.x * scale), g = quantise(a.y * scale), b = quantise(a.z * scale);
    out[k] = (r << 16) | (g << 8) | b;
}

// pack_0rgb for the bands of one GPU, written as a compact block (8 rows per
// band, lowest row first); see rt_export_packed_bands in rt_hip.h
__global__ __launch_bounds__(256)
void pack_bands(const float4 *__restrict__ accum, uint32_t *__restrict__ out, uint32_t W, uint32_t H,
                uint32_t band_begin, uint32_t band_stride, uint32_t n_bands) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t m = blockIdx.y;                 // row of the block
    if (x >= W || m >= n_bands * TILE) return;
    const uint32_t j = (band_begin + (m / TILE) * band_stride) * TILE + (m % TILE);
    uint32_t v = 0u;
    if (j < H) {
        float4 a = accum[(size_t)(H - 1u - j) * W + x];
        float scale = 1.0f / a.w;
        v = (quantise(a.x * scale) << 16) | (quantise(a.y * scale) << 8) | quantise(a.z * scale);
    }
    out[(size_t)m * W + x] = v;
}

// The inverse of pack_bands for a whole node: `blocks` holds the compact blocks of n_ctx
// contexts one after the other (block_px pixels each; context g rendered the bands
// b % n_ctx == g); the frame is rebuilt in display order, like the receive loop that fills
// img_buffer from the workers' messages (src/raytracer.rs:176-185).
__global__ __launch_bounds__(256)
void compose_bands(const uint32_t *__restrict__ blocks, uint32_t *__restrict__ out, uint32_t W, uint32_t H,
                   uint32_t n_ctx, size_t block_px) {
    const uint32_t x = blockIdx.x * blockDim.x + threadIdx.x;
    const uint32_t j = blockIdx.y;                 // row from the bottom
    if (x >= W || j >= H) return;
    const uint32_t band = j / TILE, g = band % n_ctx, m = (band / n_ctx) * TILE + (j % TILE);
    out[(size_t)(H - 1u - j) * W + x] = blocks[(size_t)g * block_px + (size_t)m * W + x];
}

// div_refined and sqrt_refined against `/` and sqrtf on random operands of their stated ranges
// (exponents uniform over the range, one mantissa in four all zeros, all ones or one bit off
// either): bad[0] counts quotients that differ in any bit, bad[1] roots.
__global__ __launch_bounds__(256)
void check_arith(uint32_t rounds, uint32_t seed, unsigned long long *bad) {
    uint32_t state = mix32((blockIdx.x * blockDim.x + threadIdx.x) ^ mix32(seed));
    auto next = [&]() { state = state * 747796405u + 2891336453u; return mix32(state); };
    auto operand = [&](int e_min, int e_max, bool negative_too) {
        const uint32_t w = next(), pick = next();
        uint32_t mant = w & 0x7fffffu;
        if ((pick & 3u) == 0u) {
            const uint32_t edge[4] = {0u, 0x7fffffu, 1u, 0x7ffffeu};
            mant = edge[(pick >> 2) & 3u];
        }
        const uint32_t e = (uint32_t)(127 + e_min) + (pick >> 8) % (uint32_t)(e_max - e_min + 1);
        return __uint_as_float((negative_too ? (w & 0x80000000u) : 0u) | (e << 23) | mant);
    };
    uint32_t bad_div = 0, bad_sqrt = 0;
    for (uint32_t k = 0; k < rounds; ++k) {
        const float y = operand(-30, 29, true), n = operand(-90, 59, true), x = operand(-96, 127, false);
        const float q = div_refined(n, y, rcp_refined(y));
        bad_div += __float_as_uint(q) != __float_as_uint(n / y);
        bad_sqrt += __float_as_uint(sqrt_refined(x)) != __float_as_uint(__builtin_sqrtf(x));
    }
    if (bad_div) atomicAdd(bad, (unsigned long long)bad_div);
    if (bad_sqrt) atomicAdd(bad + 1, (unsigned long long)bad_sqrt);
}

}  // namespace rtk

// ------------------------------------------------------------ static scratch
// Staging ring and queue counters of rt_render_static, the entry point for callers that
// cannot allocate: one array per GPU in the code object's own device memory, 96 MiB + 256 B,
// held from the first use of the library on that GPU until the process ends.  Bytes 0..7: the
// two queue counters, 64..71: segment count sink, from 256: the ring (DEV_STAGE_WAVE_BYTES
// for each of RTK_STATIC_WAVES wavefronts).  In host code the array's name is only a handle:
// its address on a device comes from hipGetSymbolAddress, once per device.
#define RTK_STATIC_WAVES RT_STATIC_MAX_WAVES
#define RTK_STATIC_HEAD 256u
static __device__ __align__(256) unsigned char scratch[RT_STATIC_SCRATCH_BYTES];
static_assert(RT_STATIC_SCRATCH_BYTES == RTK_STATIC_HEAD + (size_t)RTK_STATIC_WAVES * DEV_STAGE_WAVE_BYTES,
              "rt_hip.h states the size of the static scratch");

hipError_t rtk_static_scratch(int device, RtkStaticScratch *out) {
    constexpr int MAX_DEVICES = 64;
    static std::mutex guard;
    static unsigned char *base[MAX_DEVICES] = {};
    if (device < 0 || device >= MAX_DEVICES || !out) return hipErrorInvalidValue;
    std::lock_guard<std::mutex> lock(guard);
    if (!base[device]) {
        int current = -1;
        hipError_t e = hipGetDevice(&current);
        if (e != hipSuccess) return e;
        if (current != device) return hipErrorInvalidDevice;   // the symbol resolves on the current device
        void *ptr = nullptr;
        e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
        if (e != hipSuccess) return e;
        if (!ptr) return hipErrorInvalidSymbol;
        base[device] = static_cast<unsigned char *>(ptr);
    }
    out->queue = reinterpret_cast<uint32_t *>(base[device]);
    out->segments = reinterpret_cast<unsigned long long *>(base[device] + 64);
    out->stage = reinterpret_cast<StagedRgb *>(base[device] + RTK_STATIC_HEAD);
    out->stage_waves = RTK_STATIC_WAVES;
    out->head_bytes = RTK_STATIC_HEAD;
    return hipSuccess;
}

// ------------------------------------------------------------ launchers
hipError_t rtk_launch_pack_bands(const float4 *accum, uint32_t *out, uint32_t W, uint32_t H,
                                 uint32_t band_begin, uint32_t band_stride, uint32_t n_bands,
                                 hipStream_t stream) {
    dim3 grid((W + 255u) / 256u, n_bands * rtk::TILE, 1), block(256, 1, 1);
    hipLaunchKernelGGL(rtk::pack_bands, grid, block, 0, stream, accum, out, W, H, band_begin, band_stride,
                       n_bands);
    return hipGetLastError();
}
// The instantiations: scenes without a tree have one kernel for every depth budget, scenes with
// one a kernel for up to DEV_STACK_SHALLOW bounces and one for more (index: modes, + 4 for the latter).
static const void *render_kernel(uint32_t which) {
    switch (which) {
    case 0: return (const void *)rtk::render_tiles<0, RT_MAX_BOUNCES>;
    case 1: return (const void *)rtk::render_tiles<1, RT_MAX_BOUNCES>;
    case 2: return (const void *)rtk::render_tiles<2, DEV_STACK_SHALLOW>;
    case 3: return (const void *)rtk::render_tiles<3, DEV_STACK_SHALLOW>;
    case 6: return (const void *)rtk::render_tiles<2, RT_MAX_BOUNCES>;
    case 7: return (const void *)rtk::render_tiles<3, RT_MAX_BOUNCES>;
    case 8: return (const void *)rtk::render_tiles<0, RT_MAX_BOUNCES, true>;    // several frames per launch
    default: return nullptr;
    }
}
static uint32_t kernel_index(uint32_t modes, uint32_t max_bounces) {
    return (modes & 3u) + (((modes & 2u) && max_bounces > DEV_STACK_SHALLOW) ? 4u : 0u);
}
// workgroups of a render kernel that fit one CU, asked once per process (the same on every GPU of a node)
static uint32_t blocks_per_cu(uint32_t which) {
    static std::once_flag once;
    static int fit[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    std::call_once(once, [] {
        for (int m = 0; m < 9; ++m) {
            const void *kernel = render_kernel((uint32_t)m);
            if (!kernel) continue;
            int n = 0;
            if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&n, kernel, rtk::waves_of(m & 3) * 64, 0) != hipSuccess || n <= 0)
                n = 4 * rtk::occ_hint_of(m & 3) / rtk::waves_of(m & 3);      // what the compiler was told
            fit[m] = n;
        }
    });
    return (uint32_t)fit[which < 9u ? which : 0u];
}
uint32_t rtk_resident_waves(uint32_t n_cus, uint32_t modes) {
    (void)blocks_per_cu(0u);   // asked here, before the first launch, so that no launch has to
    return n_cus * 4u * (uint32_t)rtk::occ_of((int)(modes & 3u));
}
void rtk_kernel_variant(uint32_t modes, uint32_t max_bounces, uint32_t *scan_modes, uint32_t *depth_budget) {
    *scan_modes = modes & 3u;
    *depth_budget = ((modes & 2u) && max_bounces <= DEV_STACK_SHALLOW) ? DEV_STACK_SHALLOW : RT_MAX_BOUNCES;
}
bool rtk_has_frames_kernel(uint32_t modes) { return (modes & 3u) == 0u; }
hipError_t rtk_launch_render(const DevScene &sc, const rt_camera &cam, const rt_params &p,
                             const DevFrame &fr, uint32_t n_bands, uint32_t n_cus, uint32_t stage_waves,
                             int unit_px_log2, hipStream_t stream, const rt_camera *cams, uint32_t n_frames) {
    const uint32_t waves = (uint32_t)rtk::waves_of((int)(sc.modes & 3u));
    const uint32_t tiles_x = (p.width + rtk::TILE - 1u) / rtk::TILE;
    bool frames = cams != nullptr;
    if (frames && (n_frames == 0u || !rtk_has_frames_kernel(sc.modes))) return hipErrorInvalidValue;
    if (!frames) n_frames = 1u;
    // (a launch of a scene without long runs whose units are short goes to the same kernel, as one frame)
    if (!frames && rtk_has_frames_kernel(sc.modes) && p.sample_count * 64u <= (1u << DEV_STAGE_ITEM_BITS) / 2u) frames = true;
    const uint64_t n_tiles = (uint64_t)tiles_x * n_bands * n_frames;
    // workgroups the chip holds at once (what the kernel is written for, and no more than the runtime
    // says fit a CU: a workgroup that is not resident would start its first unit when another one ends),
    // and the staging ring has room for
    const uint64_t resident = std::min<uint64_t>((uint64_t)n_cus * std::min<uint32_t>(4u * rtk::occ_of((int)(sc.modes & 3u)) / waves, blocks_per_cu(frames ? 8u : kernel_index(sc.modes, p.max_bounces))),
                                                 stage_waves / waves);
    if (p.sample_count > DEV_SAMPLE_CHUNK || p.width > 65536u || p.height > 65536u || resident == 0u) return hipErrorInvalidValue;
    const uint64_t max_items = 1ull << DEV_STAGE_ITEM_BITS;    // a unit's colours must fit one ring slot
    // Pixels of a tile per work unit (one pixel, part of a row, or whole rows): as many as a ring slot's
    // 512 work items (eight per lane) allow with the samples of this launch, 64 at most.  Fewer items per
    // unit are worse: measured on 4K frames at 64 samples per launch, whole
    // frame / one eighth of the bands, 10,000 spheres: 4 pixels 135.2 / 17.4 ms, 8 pixels 132.2 /
    // 17.8, 16 pixels 132.1 / 19.1, 32 pixels 135.3 / 21.9, 64 pixels 139.1 / 26.5; the
    // million-triangle scene 69.8 / 9.2 ms with 4, 64.9 / 9.4 with 8, 75.9 / 22.1 with 64.
    if (unit_px_log2 < 0 || unit_px_log2 > 6) {
        unit_px_log2 = 6;
        while (unit_px_log2 > 0 && ((uint64_t)p.sample_count << unit_px_log2) > max_items) --unit_px_log2;
        // With a tree and many samples, half of that: the rays of a wave then come from fewer pixels and walk
        // more alike, which is worth more than the longer unit (whole 4K frame, 256 samples per launch, units
        // of one / two pixels: 10,000 spheres 387.0 / 390.4 ms, a million triangles 177.1 / 177.7, mesh scene
        // at 1080p 36.03 / 36.05; 128 samples, two / four pixels: 196.5 / 197.1, 90.4 / 90.7, 18.33 / 18.24)
        if ((sc.modes & 2u) && unit_px_log2 > 0 && p.sample_count >= 128u) --unit_px_log2;
        // a small launch (one GPU's share of a frame): half as long units, so that a wave still
        // gets a few dozen of them.  Scenes without a tree have short segments and feel the cost of
        // a unit more (default scene at 64 samples, 256- against 512-item units: whole 1080p frame
        // 6.1 / 4.4 ms, half of its bands 3.2 / 2.5, a quarter 1.7 / 1.4, an eighth 1.02 / 0.98), so
        // there units are halved only below three per wave.
        const uint64_t want = (sc.modes & 2u) ? 48u : 3u;
        if (unit_px_log2 > 0 && (n_tiles << (6 - unit_px_log2)) < want * resident * waves && ((uint64_t)p.sample_count << unit_px_log2) >= 512u)
            --unit_px_log2;
    }
    if (((uint64_t)p.sample_count << unit_px_log2) > max_items) return hipErrorInvalidValue;
    const uint64_t n_units = n_tiles << (6 - unit_px_log2);
    if (n_units >= 0xffffffffull - 64u * resident * waves) return hipErrorInvalidValue;   // unit numbers handed out: below n_units + waves x (1 + stretch)
    // as many workgroups as the chip holds, fewer when there are not that many work units
    uint64_t groups = std::max<uint64_t>(1u, std::min<uint64_t>(resident, (n_units + waves - 1u) / waves));
    // development aid: RT_HIP_GROUPS=n fixes the number of workgroups (at most what is resident)
    if (const char *env = std::getenv("RT_HIP_GROUPS"))
        if (std::atoi(env) > 0) groups = std::min<uint64_t>(resident, (uint64_t)std::atoi(env));
    dim3 grid((uint32_t)groups, 1, 1), block(waves * 64, 1, 1);
    const uint32_t r = (uint32_t)unit_px_log2;
    // Units per request to the queue counter: 8, 4 or 2 while every wave still has 8, 4 or 2 times 32
    // units to come, else 1.  Long stretches end a launch unevenly even when they shrink towards its
    // end (the default scene at 1080p, 32 units per wave: 8.0 ms with stretches of up to 8 from the
    // start, 4.5 without; a wave's stretch on the way was asked for a stretch earlier), so they are
    // used only while much is left, and only by the kernels with a tree: 10,000 spheres 4K -0.4 %, a
    // million triangles -1 %.  (The kernel for short units or several frames asks for one batch at a time.)
    // Short units (a launch of less than eight samples: whole tiles with fewer than 512 items) come in
    // batches that make up for it, as far as every wave still gets a batch
    const uint64_t unit_items = (uint64_t)p.sample_count << unit_px_log2;
    uint64_t unit_batch = 1u;
    if ((frames || (sc.modes & 2u)) && unit_items != 0u)
        unit_batch = std::max<uint64_t>(1u, std::min<uint64_t>(max_items / unit_items, (n_units + groups * waves - 1u) / (groups * waves)));
    const uint64_t stretch_unit = std::min<uint64_t>(32u * groups * waves * unit_batch, 0x1fffffffu);
    // idle lanes a wave waits for before it hands out new work items
    const uint32_t fetch_min = (sc.modes & 3u) == 0u ? RTK_FETCH_MIN0
                               : ((sc.modes & 2u) && sc.lds_kind == DEV_TRI) ? RTK_FETCH_MIN_TRI : RTK_FETCH_MIN;
    rtk::KernArgs ka{sc, cam, p, fr, n_bands, r, (uint32_t)stretch_unit, fetch_min, cams, n_frames, (uint32_t)unit_batch};
    if (p.max_bounces > RT_MAX_BOUNCES) return hipErrorInvalidValue;
    const uint32_t which = frames ? 8u : kernel_index(sc.modes, p.max_bounces);
    if (which >= 4u && which < 8u) {
        // the deep material stack leaves less room for the tree's top levels
        ka.dsc.lds_level = sc.lds_deep_level; ka.dsc.lds_first = sc.lds_deep_first; ka.dsc.lds_count = sc.lds_deep_count;
        if (sc.lds_deep_level == 0xffffffffu) ka.dsc.lds_run = 0xffffffffu;
    }
    switch (which) {
    case 0: hipLaunchKernelGGL((rtk::render_tiles<0, RT_MAX_BOUNCES>), grid, block, 0, stream, ka); break;
    case 1: hipLaunchKernelGGL((rtk::render_tiles<1, RT_MAX_BOUNCES>), grid, block, 0, stream, ka); break;
    case 2: hipLaunchKernelGGL((rtk::render_tiles<2, DEV_STACK_SHALLOW>), grid, block, 0, stream, ka); break;
    case 3: hipLaunchKernelGGL((rtk::render_tiles<3, DEV_STACK_SHALLOW>), grid, block, 0, stream, ka); break;
    case 6: hipLaunchKernelGGL((rtk::render_tiles<2, RT_MAX_BOUNCES>), grid, block, 0, stream, ka); break;
    case 8: hipLaunchKernelGGL((rtk::render_tiles<0, RT_MAX_BOUNCES, true>), grid, block, 0, stream, ka); break;
    default: hipLaunchKernelGGL((rtk::render_tiles<3, RT_MAX_BOUNCES>), grid, block, 0, stream, ka); break;
    }
    return hipGetLastError();
}
hipError_t rtk_launch_compose_bands(const uint32_t *blocks, uint32_t *out, uint32_t W, uint32_t H, uint32_t n_ctx,
                                    size_t block_px, hipStream_t stream) {
    dim3 grid((W + 255u) / 256u, H, 1), block(256, 1, 1);
    hipLaunchKernelGGL(rtk::compose_bands, grid, block, 0, stream, blocks, out, W, H, n_ctx, block_px);
    return hipGetLastError();
}
hipError_t rtk_launch_check_arith(uint32_t blocks, uint32_t rounds, uint32_t seed, unsigned long long *bad,
                                  hipStream_t stream) {
    hipLaunchKernelGGL(rtk::check_arith, dim3(blocks, 1, 1), dim3(256, 1, 1), 0, stream, rounds, seed, bad);
    return hipGetLastError();
}
hipError_t rtk_launch_pack(const float4 *accum, uint32_t *out, uint32_t n_px, hipStream_t stream) {
    dim3 grid((n_px + 255u) / 256u, 1, 1), block(256, 1, 1);
    hipLaunchKernelGGL(rtk::pack_0rgb, grid, block, 0, stream, accum, out, n_px);
    return hipGetLastError();
}
