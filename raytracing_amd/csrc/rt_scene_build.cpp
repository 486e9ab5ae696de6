// rt_scene_build.cpp - turns the structure-of-arrays scene of the C ABI into the
// device layout the render kernel scans (host side only, no HIP calls).
//
// The reference's Vec<Box<dyn Hittable>> (src/geometry/mod.rs:228-238) is cut
// into maximal runs of one primitive kind, in list order.  A run is scanned in
// one of three ways:
//   RUN_EXACT  every primitive gets its hit() arithmetic, in list order;
//   RUN_FLAT   (spheres) list order, but eight spheres per step first go through
//              a conservative rejection test on packed FMAs;
//   RUN_TREE   (spheres, triangles) the run is re-ordered by recursive median
//              splits and an implicit 8-ary tree of bounding boxes is laid over it
//              (spheres: the rejection test on their own balls at the leaves;
//              triangles: their gate boxes).  Ties between equal hit distances are
//              resolved by list position explicitly, so the result equals the
//              list-order scan.
#include "rt_scene_build.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <numeric>

namespace {

inline float bits_to_float(uint32_t u) { float f; std::memcpy(&f, &u, 4); return f; }
inline uint32_t float_to_bits(float f) { uint32_t u; std::memcpy(&u, &f, 4); return u; }

struct Ball { double x, y, z, r; };

// smallest float >= v
float round_up(double v) {
    float f = (float)v;
    if ((double)f < v) f = std::nextafterf(f, INFINITY);
    return f;
}

// largest float <= v
float round_down(double v) {
    float f = (float)v;
    if ((double)f > v) f = std::nextafterf(f, -INFINITY);
    return f;
}

// A ball the kernel's rejection tests cannot speak for: NaN or infinite numbers, or
// magnitudes at which the tests' intermediates could overflow.  The rejection scan never
// rejects it; a tree does not hold it at all: such primitives are listed with their run and
// tested for every ray before the walk (Sphere::hit may accept them with a NaN distance
// whatever the closest hit so far is, which no bounding volume could express).
const double BALL_RANGE = 1e15;
const Ball UNBOUNDED = {0.0, 0.0, 0.0, INFINITY};
bool unbounded(const Ball &b) { return !(std::abs(b.r) < BALL_RANGE); }
Ball bounded_or_unbounded(const Ball &b) {
    if (std::abs(b.x) < BALL_RANGE && std::abs(b.y) < BALL_RANGE && std::abs(b.z) < BALL_RANGE && std::abs(b.r) < BALL_RANGE)
        return b;   // comparisons with NaN are false
    return UNBOUNDED;
}

// Bounding box of a leaf or node, in double.  EVERYWHERE marks a primitive that cannot be
// bounded; NOWHERE is what the tree holds in its place, and any node without bounded
// primitives below it.
struct Box { double lo[3], hi[3]; };
const Box EVERYWHERE = {{-INFINITY, -INFINITY, -INFINITY}, {INFINITY, INFINITY, INFINITY}};
const Box NOWHERE = {{INFINITY, INFINITY, INFINITY}, {-INFINITY, -INFINITY, -INFINITY}};
bool nowhere(const Box &b) { return b.lo[0] > b.hi[0]; }

// Gate of the triangle test (rt_hip.h): the bounding box of v0, v0+e1, v0+e2 widened by
// 2^-7 of its largest extent plus 2^-20 of the largest |coordinate|, every step in
// binary32 from the f32 edges the kernel uses.  The kernel's gate also counts the ray
// origin among the coordinates; that part is covered by the pad of the slab test.
Box tri_gate_box(const rt_scene *s, uint32_t k) {
    const float v0[3] = {s->tri_v0x[k], s->tri_v0y[k], s->tri_v0z[k]};
    const float e1[3] = {s->tri_v1x[k] - v0[0], s->tri_v1y[k] - v0[1], s->tri_v1z[k] - v0[2]};
    const float e2[3] = {s->tri_v2x[k] - v0[0], s->tri_v2y[k] - v0[1], s->tri_v2z[k] - v0[2]};
    float lo[3], hi[3], ext = 0.0f, mag = 0.0f;
    for (int ax = 0; ax < 3; ++ax) {
        const float a = v0[ax], b = v0[ax] + e1[ax], c = v0[ax] + e2[ax];
        lo[ax] = std::min(a, std::min(b, c));
        hi[ax] = std::max(a, std::max(b, c));
        if (!(a == a && b == b && c == c)) return EVERYWHERE;   // NaN vertex
        ext = std::max(ext, hi[ax] - lo[ax]);
        mag = std::max(mag, std::max(std::abs(lo[ax]), std::abs(hi[ax])));
    }
    if (!(mag < BALL_RANGE)) return EVERYWHERE;
    const float delta = ext * 0.0078125f + mag * 9.5367431640625e-07f;
    Box bx;
    for (int ax = 0; ax < 3; ++ax) { bx.lo[ax] = lo[ax] - delta; bx.hi[ax] = hi[ax] + delta; }
    return bx;
}

// Rejection record {cx,cy,cz,thr} of the ball (c, r) at margin eps:
//   thr = r^2 (1+eps) - |c|^2 (1-eps), rounded up; the kernel rejects when
//   |o|^2 (1-eps) - 2 o.c - ((c-o).u)^2 > thr.
// Balls whose magnitudes leave the range in which the kernel's error bound holds
// are never rejected (thr = +inf).
float4 reject_record(const Ball &b, double eps) {
    const double c2 = b.x * b.x + b.y * b.y + b.z * b.z;
    const double r2 = b.r * b.r;
    float thr = INFINITY;
    if (!unbounded(b) && c2 + r2 > 1e-24 && eps <= 1.0 / 16.0)
        thr = round_up(r2 * (1.0 + eps) - c2 * (1.0 - eps));
    return make_float4((float)b.x, (float)b.y, (float)b.z, thr);
}
const float4 NEVER_PASS = {0.f, 0.f, 0.f, -INFINITY};   // g > -inf: always rejected

// Slot order for the implicit 8-ary tree: order[k] = index (within the run) of the
// primitive placed at slot k.  The tree's shape is fixed by the slot numbers alone
// (eight consecutive slots form a leaf step, eight consecutive steps a node, ...),
// so the order decides how compact every node is.  Top-down: the slots a node
// covers are cut into its (up to eight) children's slot ranges by recursive
// bisection at the median along the longest axis of the centres' bounding box;
// then each child is arranged the same way.
struct Arranger {
    const std::vector<Ball> &balls;
    std::vector<uint32_t> &idx;

    double coord(uint32_t i, int axis) const { return axis == 0 ? balls[i].x : axis == 1 ? balls[i].y : balls[i].z; }

    // cut idx[a, a+n) into `groups` consecutive chunks of `cap` primitives (the last may be short)
    void split(size_t a, size_t n, size_t groups, size_t cap) {
        if (groups <= 1) return;
        double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
        for (size_t k = a; k < a + n; ++k)
            for (int ax = 0; ax < 3; ++ax) {
                const double v = coord(idx[k], ax);
                if (v < lo[ax]) lo[ax] = v;
                if (v > hi[ax]) hi[ax] = v;
            }
        int axis = 0;
        if (hi[1] - lo[1] > hi[axis] - lo[axis]) axis = 1;
        if (hi[2] - lo[2] > hi[axis] - lo[axis]) axis = 2;
        const size_t gl = groups / 2, left = gl * cap;
        std::nth_element(idx.begin() + a, idx.begin() + a + left, idx.begin() + a + n, [&](uint32_t p, uint32_t q) {
            const double u = coord(p, axis), v = coord(q, axis);
            return u < v || (u == v && p < q);   // total order: centres are finite, ties by index
        });
        split(a, left, gl, cap);
        split(a + left, n - left, groups - gl, cap);
    }
    // arrange idx[a, a+n), whose node's children each cover `cap` primitives
    void arrange(size_t a, size_t n, size_t cap) {
        if (n <= 8) return;
        const size_t groups = (n + cap - 1) / cap;
        split(a, n, groups, cap);
        if (cap > 8)
            for (size_t g = 0; g < groups; ++g) arrange(a + g * cap, std::min(cap, n - g * cap), cap / 8);
    }
};

std::vector<uint32_t> tree_order(const std::vector<Ball> &balls) {
    std::vector<uint32_t> order(balls.size());
    std::iota(order.begin(), order.end(), 0u);
    size_t steps = (balls.size() + 7) / 8, cap = 8;
    while (steps > 8) { steps = (steps + 7) / 8; cap *= 8; }   // cap = primitives under one child of the top node
    Arranger ar{balls, order};
    ar.arrange(0, balls.size(), cap);
    return order;
}

// pair layout the kernel reads with s_load_dwordx16: two records interleaved as
// {cx0,cx1, cy0,cy1, cz0,cz1, thr0,thr1}
void append_paired(std::vector<float4> &out, const std::vector<float4> &recs) {
    for (size_t k = 0; k + 1 < recs.size(); k += 2) {
        const float4 a = recs[k], b = recs[k + 1];
        out.push_back(make_float4(a.x, b.x, a.y, b.y));
        out.push_back(make_float4(a.z, b.z, a.w, b.w));
    }
}

// Appends one level of boxes (one per entry of `boxes`, in steps of eight, DEV_NODE_F4 float4
// per step: per pair of boxes {lo0,lo1,hi0,hi1,lo0,lo1} for x, for y, for z - a ray reads the first
// four or the last four, whichever has its near planes first), rounded outwards.
void append_box_level(std::vector<float4> &out, const std::vector<Box> &boxes) {
    const size_t m = boxes.size(), n_steps = (m + 7) / 8;
    for (size_t st = 0; st < n_steps; ++st) {
        float lo[8][3], hi[8][3];
        for (size_t c = 0; c < 8; ++c)
            for (int ax = 0; ax < 3; ++ax) {
                // children past the end are masked out by the kernel; a box that holds nothing
                // becomes one far-away point: its three slabs have no thickness, so the interval
                // the slab test cuts down ends up empty whatever the ray
                const bool have = 8 * st + c < m;
                const bool empty = !have || nowhere(boxes[8 * st + c]);
                lo[c][ax] = empty ? 3.0e38f : round_down(boxes[8 * st + c].lo[ax]);
                hi[c][ax] = empty ? 3.0e38f : round_up(boxes[8 * st + c].hi[ax]);
            }
#if DEV_NODE_MIRROR
        float rec[72];
        for (int pr = 0; pr < 4; ++pr)
            for (int ax = 0; ax < 3; ++ax) {
                float *r = rec + 6 * (3 * pr + ax);
                r[0] = lo[2 * pr][ax]; r[1] = lo[2 * pr + 1][ax]; r[2] = hi[2 * pr][ax]; r[3] = hi[2 * pr + 1][ax];
                r[4] = r[0]; r[5] = r[1];
            }
        for (int q = 0; q < 18; ++q) out.push_back(make_float4(rec[4 * q], rec[4 * q + 1], rec[4 * q + 2], rec[4 * q + 3]));
#else
        for (int pr = 0; pr < 4; ++pr)
            for (int ax = 0; ax < 3; ++ax)
                out.push_back(make_float4(lo[2 * pr][ax], lo[2 * pr + 1][ax], hi[2 * pr][ax], hi[2 * pr + 1][ax]));
#endif
    }
}

// Builds the implicit 8-ary tree over the leaves of a run (already in slot order) and
// appends its records to scene.tree_rec.
//   level 0   one box per primitive (a sphere's extent, a triangle's gate box), in steps of
//             eight, DEV_NODE_F4 float4 per step;
//   level L>0 one box per step of level L-1, in steps of eight.
// Boxes hold plain extents rounded outwards; the kernel widens them per ray by a pad that
// covers what the reference's f32 arithmetic can add to a primitive's reach (for a sphere:
// the root of a discriminant that is uncertain by kappa |o-c|^2 lies up to
// sqrt(r^2 + kappa |o-c|^2) from the centre; rho_min, the smallest radius of the run, bounds
// that) plus the rounding of the slab test itself.  Boxes fit flat or elongated groups
// (spheres resting on a plane, a terrain mesh) far better than balls do.  Leaves given as
// NOWHERE (primitives that cannot be bounded) take no part.
void build_tree(HostScene &sc, DevRun &run, const std::vector<Box> &leaf_boxes, double rho_min) {
    for (const Box &b : leaf_boxes) {
        if (nowhere(b)) continue;
        double far2 = 0.0;
        for (int ax = 0; ax < 3; ++ax) {
            const double m = std::max(std::abs(b.lo[ax]), std::abs(b.hi[ax]));
            far2 += m * m;
        }
        run.reach = std::max(run.reach, round_up(std::sqrt(far2)));
    }
    run.inv_2rho = rho_min > 0 && rho_min < INFINITY ? round_up(0.5 / rho_min) : INFINITY;
    std::vector<Box> cur = leaf_boxes;
    for (uint32_t level = 0;; ++level) {
        run.level_base[level] = (uint32_t)sc.tree_rec.size();
        run.level_aux[level] = (uint32_t)cur.size();
        append_box_level(sc.tree_rec, cur);
        const size_t n_steps = (cur.size() + 7) / 8;
        if (level > 0 && (n_steps == 1 || level + 1 == DEV_MAX_LEVELS)) {
            run.n_levels = level + 1;
            run.top_steps = (uint32_t)n_steps;
            return;
        }
        // one box per step (eight boxes) of this level
        std::vector<Box> up(n_steps);
        for (size_t k = 0; k < n_steps; ++k) {
            Box bx = NOWHERE;
            for (size_t c = 8 * k; c < std::min(cur.size(), 8 * k + 8); ++c)
                for (int ax = 0; ax < 3 && !nowhere(cur[c]); ++ax) {
                    bx.lo[ax] = std::min(bx.lo[ax], cur[c].lo[ax]);
                    bx.hi[ax] = std::max(bx.hi[ax], cur[c].hi[ax]);
                }
            up[k] = bx;
        }
        cur.swap(up);
    }
}

}  // namespace

int build_host_scene(const rt_scene *s, bool use_tree, HostScene &sc, std::string &err) {
    const uint64_t n64 = (uint64_t)s->n_spheres + s->n_planes + s->n_cubes + s->n_tris;
    if (n64 > 0x7fffffffu) { err = "too many primitives"; return RT_E_BADARG; }
    const uint32_t n = (uint32_t)n64;
    if ((s->n_spheres && !(s->sph_cx && s->sph_cy && s->sph_cz && s->sph_r && s->sph_mat && s->sph_id)) ||
        (s->n_planes && !(s->pl_nx && s->pl_ny && s->pl_nz && s->pl_px && s->pl_py && s->pl_pz && s->pl_mat && s->pl_id)) ||
        (s->n_cubes && !(s->cb_cx && s->cb_cy && s->cb_cz && s->cb_w && s->cb_h && s->cb_d && s->cb_mat && s->cb_id)) ||
        (s->n_tris && !(s->tri_v0x && s->tri_v0y && s->tri_v0z && s->tri_v1x && s->tri_v1y && s->tri_v1z &&
                        s->tri_v2x && s->tri_v2y && s->tri_v2z && s->tri_mat && s->tri_id)) ||
        (s->n_materials && !(s->mat_kind && s->mat_r && s->mat_g && s->mat_b && s->mat_param))) {
        err = "null array in a non-empty group";
        return RT_E_BADARG;
    }
    if (n > 0 && s->n_materials == 0) { err = "primitives but no materials"; return RT_E_BADARG; }
    for (uint32_t m = 0; m < s->n_materials; ++m)
        if (s->mat_kind[m] < RT_METAL || s->mat_kind[m] > RT_DIELECTRIC) { err = "unknown material kind"; return RT_E_BADARG; }

    // list order: the ids of all groups must be a permutation of 0..n-1
    struct Entry { int kind = -1; uint32_t k = 0; };
    std::vector<Entry> list(n);
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
    if (!ok) { err = "ids are not a permutation of the list, or a material index is out of range"; return RT_E_BADARG; }

    for (uint32_t m = 0; m < s->n_materials; ++m) {
        sc.mat_rec.push_back(make_float4(s->mat_r[m], s->mat_g[m], s->mat_b[m], s->mat_param[m]));
        // Dielectric::scatter, src/geometry/mod.rs:327: refraction_ratio = 1.0/ir for a front face, ir
        // otherwise; reflectance :313-317: r0 = ((1-ratio)/(1+ratio))^2.  Functions of the material alone,
        // evaluated here once in the same f32 operations.
        const float ir = s->mat_param[m], inv_ir = 1.0f / ir;
        float r0_front = (1.0f - inv_ir) / (1.0f + inv_ir), r0_back = (1.0f - ir) / (1.0f + ir);
        r0_front = r0_front * r0_front;
        r0_back = r0_back * r0_back;
        sc.mat_rec.push_back(make_float4(bits_to_float((uint32_t)s->mat_kind[m]), inv_ir, r0_front, r0_back));
    }
    for (uint32_t k = 0; k < s->n_spheres; ++k) {
        const float lim = 1073741824.0f, r = std::abs(s->sph_r[k]);
        if (!(std::abs(s->sph_cx[k]) <= lim && std::abs(s->sph_cy[k]) <= lim && std::abs(s->sph_cz[k]) <= lim &&
              r <= lim && r >= 1.0f / lim))
            sc.cheap_ok = 0;
    }

    const double LEAF_EPS_SPHERE = 1.0 / 65536.0;   // proven bound of the kernel's test, see cull_pair

    // Sphere runs of DEV_CULL_MIN_RUN and more get a tree.  (The list-order rejection scan, RUN_FLAT, is
    // kept for measurements - RT_HIP_FLAT_SCAN=1 - because it does the reference's work per segment;
    // its kernels are the ones that do not fit the register file without spills, and from 32 spheres
    // on a small tree costs a ray less than the rejection scan of the whole run.)
    const uint32_t sphere_tree_min = DEV_CULL_MIN_RUN;

    for (uint32_t id = 0; id < n;) {
        const int kind = list[id].kind;
        uint32_t end = id;
        while (end < n && list[end].kind == kind) ++end;
        const uint32_t count = end - id;
        DevRun run{};
        run.kind = (uint32_t)kind;
        run.count = count;
        run.id0 = id;
        run.mode = RUN_EXACT;

        if (kind == DEV_SPHERE) {
            run.first = (uint32_t)sc.sph_geom.size();
            std::vector<uint32_t> order(count);
            std::iota(order.begin(), order.end(), 0u);
            std::vector<Ball> balls(count);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + k].k;
                balls[k] = bounded_or_unbounded({s->sph_cx[src], s->sph_cy[src], s->sph_cz[src], s->sph_r[src]});
            }
            if (count >= sphere_tree_min && use_tree) {
                run.mode = RUN_TREE;
                order = tree_order(balls);
            } else if (count >= DEV_CULL_MIN_RUN) {
                run.mode = RUN_FLAT;
            }
            std::vector<Ball> sorted(count);
            run.order_base = (uint32_t)sc.order.size();
            sc.order.resize(sc.order.size() + count);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + order[k]].k;
                sorted[k] = balls[order[k]];
                sc.sph_geom.push_back(make_float4(s->sph_cx[src], s->sph_cy[src], s->sph_cz[src], s->sph_r[src]));
                sc.sph_mat.push_back(s->sph_mat[src]);
                sc.sph_id.push_back(id + order[k]);
                sc.order[run.order_base + order[k]] = k;   // list position within the run -> slot offset
            }
            if (run.mode == RUN_TREE) {
                std::vector<Box> boxes(count);
                double rho_min = INFINITY;
                for (uint32_t k = 0; k < count; ++k) {
                    const Ball &b = sorted[k];
                    const double r = std::abs(b.r);
                    boxes[k] = unbounded(b) ? NOWHERE : Box{{b.x - r, b.y - r, b.z - r}, {b.x + r, b.y + r, b.z + r}};
                    if (!unbounded(b)) rho_min = std::min(rho_min, r);
                    else sc.order.push_back(k);      // the run's unbounded primitives follow its order table
                }
                run.n_unbounded = (uint32_t)(sc.order.size() - run.order_base - count);
                build_tree(sc, run, boxes, rho_min);
            } else if (run.mode == RUN_FLAT) {
                std::vector<float4> recs;
                for (const Ball &b : sorted) {
                    recs.push_back(reject_record(b, LEAF_EPS_SPHERE));
                    if (!unbounded(b))
                        run.reach = std::max(run.reach, round_up(std::sqrt(b.x * b.x + b.y * b.y + b.z * b.z) + std::abs(b.r)));
                }
                while (recs.size() % 16) recs.push_back(NEVER_PASS);   // two steps per loop trip
                run.level_base[0] = (uint32_t)(sc.cull_rec.size() / 8);
                run.level_aux[0] = float_to_bits((float)(1.0 - LEAF_EPS_SPHERE));
                run.n_levels = 1;
                run.top_steps = (uint32_t)(recs.size() / 8);
                append_paired(sc.cull_rec, recs);
            }
        } else if (kind == DEV_PLANE) {
            run.first = (uint32_t)(sc.planes.size() / 2);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + k].k;
                sc.planes.push_back(make_float4(s->pl_nx[src], s->pl_ny[src], s->pl_nz[src], bits_to_float(s->pl_mat[src])));
                sc.planes.push_back(make_float4(s->pl_px[src], s->pl_py[src], s->pl_pz[src], 0.f));
            }
        } else if (kind == DEV_CUBE) {
            run.first = (uint32_t)(sc.cubes.size() / 4);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + k].k;
                // Cube::new, src/geometry/mod.rs:146-154: corners from center and extents;
                // w/2 is also what Cube::hit divides by (:213-215)
                const float cx = s->cb_cx[src], cy = s->cb_cy[src], cz = s->cb_cz[src];
                const float w = s->cb_w[src], h = s->cb_h[src], d = s->cb_d[src];
                const float hw = w / 2.0f, hh = h / 2.0f, hd = d / 2.0f;
                const float c0x = cx + (-w / 2.0f), c0y = cy + (-h / 2.0f), c0z = cz + (-d / 2.0f);
                const float c1x = cx + hw, c1y = cy + hh, c1z = cz + hd;
                sc.cubes.push_back(make_float4(cx, cy, cz, bits_to_float(s->cb_mat[src])));
                sc.cubes.push_back(make_float4(hw, hh, hd, 0.f));
                sc.cubes.push_back(make_float4(c0x, c0y, c0z, 0.f));
                sc.cubes.push_back(make_float4(c1x, c1y, c1z, 0.f));
                if (s->mat_kind[s->cb_mat[src]] == RT_DIELECTRIC) sc.need_ff_shared = 1;
            }
        } else {
            run.first = (uint32_t)(sc.tris.size() / 3);
            std::vector<uint32_t> order(count);
            std::iota(order.begin(), order.end(), 0u);
            std::vector<Ball> balls(count);    // box centres, for the slot order only
            std::vector<Box> gate(count);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + k].k;
                gate[k] = tri_gate_box(s, src);
                const bool open = gate[k].lo[0] == -INFINITY;
                balls[k] = open ? UNBOUNDED
                                : Ball{(gate[k].lo[0] + gate[k].hi[0]) / 2, (gate[k].lo[1] + gate[k].hi[1]) / 2,
                                       (gate[k].lo[2] + gate[k].hi[2]) / 2, 0.0};
            }
            if (count >= DEV_TREE_MIN_RUN && use_tree) {
                run.mode = RUN_TREE;
                order = tree_order(balls);
            }
            std::vector<Box> sorted(count);
            run.order_base = (uint32_t)sc.order.size();
            sc.order.resize(sc.order.size() + count);
            for (uint32_t k = 0; k < count; ++k) {
                const uint32_t src = list[id + order[k]].k;
                sorted[k] = gate[order[k]];
                const float ax = s->tri_v0x[src], ay = s->tri_v0y[src], az = s->tri_v0z[src];
                const float e1x = s->tri_v1x[src] - ax, e1y = s->tri_v1y[src] - ay, e1z = s->tri_v1z[src] - az;
                const float e2x = s->tri_v2x[src] - ax, e2y = s->tri_v2y[src] - ay, e2z = s->tri_v2z[src] - az;
                sc.tris.push_back(make_float4(ax, ay, az, bits_to_float(s->tri_mat[src])));
                sc.tris.push_back(make_float4(e1x, e1y, e1z, 0.f));
                sc.tris.push_back(make_float4(e2x, e2y, e2z, 0.f));
                sc.tri_id.push_back(id + order[k]);
                sc.order[run.order_base + order[k]] = k;
            }
            if (run.mode == RUN_TREE) {
                for (uint32_t k = 0; k < count; ++k)
                    if (sorted[k].lo[0] == -INFINITY) { sorted[k] = NOWHERE; sc.order.push_back(k); }
                run.n_unbounded = (uint32_t)(sc.order.size() - run.order_base - count);
                build_tree(sc, run, sorted, INFINITY);
            }
        }
        sc.runs.push_back(run);
        id = end;
    }
    // Which tree keeps its top levels in LDS: the largest one; as many levels from
    // the top as fit.  A tree's levels are laid out bottom-up, so levels >= L are the
    // contiguous tail [level_base[L], end of the top level).
    uint32_t best = 0;
    for (uint32_t r = 0; r < sc.runs.size(); ++r) {
        const DevRun &run = sc.runs[r];
        if (run.mode == RUN_FLAT) sc.modes |= 1u;
        if (run.mode != RUN_TREE) continue;
        sc.modes |= 2u;
        if (run.count <= best) continue;
        best = run.count;
        const uint32_t top = run.n_levels - 1u;
        const uint32_t end = run.level_base[top] + run.top_steps * DEV_NODE_F4;   // top >= 1: a box level
        auto levels_that_fit = [&](uint32_t room) {
            uint32_t L = top;
            while (L > 0 && end - run.level_base[L - 1] <= room) --L;
            return end - run.level_base[L] > room ? 0xffffffffu : L;   // even the top level alone may be too big
        };
        const uint32_t L = levels_that_fit(DEV_TREE_LDS_F4), Ld = levels_that_fit(DEV_TREE_LDS_F4_DEEP);
        if (L == 0xffffffffu) continue;
        sc.lds_run = r; sc.lds_level = L; sc.lds_kind = run.kind;
        sc.lds_first = run.level_base[L];
        sc.lds_count = end - run.level_base[L];
        sc.lds_deep_level = 0xffffffffu; sc.lds_deep_first = 0; sc.lds_deep_count = 0;
        if (Ld != 0xffffffffu) {
            sc.lds_deep_level = Ld;
            sc.lds_deep_first = run.level_base[Ld];
            sc.lds_deep_count = end - run.level_base[Ld];
        }
    }
    return RT_OK;
}
