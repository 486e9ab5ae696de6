// scene.hpp - C++ host-side mirror of the reference's scene API, for programs
// that drive librt_hip.so without Rust.  Names and argument meaning follow
// ojferro/raytracing: HittableList/Sphere/Plane/Cube and the materials
// (src/geometry/mod.rs:47-57,97-109,134-155,228-238,268-304), Scene
// (src/scene/mod.rs:15-88), Camera::new (src/camera/mod.rs:35-73).
// Building the list is all that happens here; no intersection or shading code.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_hip.h"

namespace rt {

struct vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    vec3() = default;
    vec3(float a, float b, float c) : x(a), y(b), z(c) {}
};
using point3 = vec3;
using colour = vec3;

// ---- materials -------------------------------------------------------------
struct Material {
    int kind;
    colour albedo;
    float param;  // Metal: fuzz, Dielectric: index_of_refraction
};
inline Material Metal(colour albedo, float fuzz) { return {RT_METAL, albedo, fuzz}; }
inline Material Lambertian(colour albedo) { return {RT_LAMBERTIAN, albedo, 0.f}; }
inline Material Dielectric(colour albedo, float index_of_refraction) {
    return {RT_DIELECTRIC, albedo, index_of_refraction};
}

// ---- primitives ------------------------------------------------------------
struct Sphere { point3 center; float radius; Material material; };
struct Plane { vec3 normal; point3 point; Material material; bool single_sided; };
struct Cube { point3 center; float w, h, d; Material material; };
struct Triangle { point3 v0, v1, v2; Material material; };  // not in the reference sources

// Insertion-ordered list (src/geometry/mod.rs:228-238) kept directly in the
// structure-of-arrays form rt_upload_scene() takes; the position in the list
// is recorded per primitive and decides ties like the reference's scan order.
class HittableList {
public:
    void add(const Sphere& s) {
        uint32_t m = push_material(s.material);
        sph_[0].push_back(s.center.x); sph_[1].push_back(s.center.y); sph_[2].push_back(s.center.z);
        sph_[3].push_back(s.radius);
        sph_mat_.push_back(m); sph_id_.push_back(next_id_++);
    }
    void add(const Plane& p) {
        uint32_t m = push_material(p.material);
        pl_[0].push_back(p.normal.x); pl_[1].push_back(p.normal.y); pl_[2].push_back(p.normal.z);
        pl_[3].push_back(p.point.x); pl_[4].push_back(p.point.y); pl_[5].push_back(p.point.z);
        pl_mat_.push_back(m); pl_id_.push_back(next_id_++);
    }
    void add(const Cube& c) {
        uint32_t m = push_material(c.material);
        cb_[0].push_back(c.center.x); cb_[1].push_back(c.center.y); cb_[2].push_back(c.center.z);
        cb_[3].push_back(c.w); cb_[4].push_back(c.h); cb_[5].push_back(c.d);
        cb_mat_.push_back(m); cb_id_.push_back(next_id_++);
    }
    void add(const Triangle& t) {
        uint32_t m = push_material(t.material);
        const point3* v[3] = {&t.v0, &t.v1, &t.v2};
        for (int k = 0; k < 3; ++k) {
            tri_[3 * k + 0].push_back(v[k]->x); tri_[3 * k + 1].push_back(v[k]->y);
            tri_[3 * k + 2].push_back(v[k]->z);
        }
        tri_mat_.push_back(m); tri_id_.push_back(next_id_++);
    }
    size_t len() const { return next_id_; }

    // Pointers stay valid until the next add().
    rt_scene flatten() const {
        rt_scene s;
        std::memset(&s, 0, sizeof s);
        s.n_spheres = (uint32_t)sph_id_.size();
        s.sph_cx = sph_[0].data(); s.sph_cy = sph_[1].data(); s.sph_cz = sph_[2].data();
        s.sph_r = sph_[3].data(); s.sph_mat = sph_mat_.data(); s.sph_id = sph_id_.data();
        s.n_planes = (uint32_t)pl_id_.size();
        s.pl_nx = pl_[0].data(); s.pl_ny = pl_[1].data(); s.pl_nz = pl_[2].data();
        s.pl_px = pl_[3].data(); s.pl_py = pl_[4].data(); s.pl_pz = pl_[5].data();
        s.pl_mat = pl_mat_.data(); s.pl_id = pl_id_.data();
        s.n_cubes = (uint32_t)cb_id_.size();
        s.cb_cx = cb_[0].data(); s.cb_cy = cb_[1].data(); s.cb_cz = cb_[2].data();
        s.cb_w = cb_[3].data(); s.cb_h = cb_[4].data(); s.cb_d = cb_[5].data();
        s.cb_mat = cb_mat_.data(); s.cb_id = cb_id_.data();
        s.n_tris = (uint32_t)tri_id_.size();
        s.tri_v0x = tri_[0].data(); s.tri_v0y = tri_[1].data(); s.tri_v0z = tri_[2].data();
        s.tri_v1x = tri_[3].data(); s.tri_v1y = tri_[4].data(); s.tri_v1z = tri_[5].data();
        s.tri_v2x = tri_[6].data(); s.tri_v2y = tri_[7].data(); s.tri_v2z = tri_[8].data();
        s.tri_mat = tri_mat_.data(); s.tri_id = tri_id_.data();
        s.n_materials = (uint32_t)mat_kind_.size();
        s.mat_kind = mat_kind_.data();
        s.mat_r = mat_[0].data(); s.mat_g = mat_[1].data(); s.mat_b = mat_[2].data();
        s.mat_param = mat_[3].data();
        return s;
    }

private:
    uint32_t push_material(const Material& m) {
        mat_kind_.push_back(m.kind);
        mat_[0].push_back(m.albedo.x); mat_[1].push_back(m.albedo.y); mat_[2].push_back(m.albedo.z);
        mat_[3].push_back(m.param);
        return (uint32_t)mat_kind_.size() - 1;
    }
    uint32_t next_id_ = 0;
    std::vector<float> sph_[4], pl_[6], cb_[6], tri_[9], mat_[4];
    std::vector<uint32_t> sph_mat_, sph_id_, pl_mat_, pl_id_, cb_mat_, cb_id_, tri_mat_, tri_id_;
    std::vector<int32_t> mat_kind_;
};

// ---- camera ----------------------------------------------------------------
struct Camera {
    rt_camera c;
    uint32_t samples_per_px;
    // Camera::new, src/camera/mod.rs:35-73 (arithmetic done by the library)
    static Camera make(float v_fov, float aspect_ratio, float aperture, float focus_dist, point3 origin,
                       point3 look_at, vec3 v_up, uint32_t samples_per_px) {
        Camera cam;
        const float o[3] = {origin.x, origin.y, origin.z}, l[3] = {look_at.x, look_at.y, look_at.z},
                    u[3] = {v_up.x, v_up.y, v_up.z};
        if (rt_camera_new(v_fov, aspect_ratio, aperture, focus_dist, o, l, u, &cam.c) != RT_OK)
            throw std::runtime_error("rt_camera_new failed");
        cam.samples_per_px = samples_per_px;
        return cam;
    }
};

// ---- Scene -----------------------------------------------------------------
// The reference keeps the dolly position in a `static mut COUNTER` that every
// worker reads while main() increments it (src/scene/mod.rs:61-63,78-87); here
// it is a member the frame loop advances between frames.
class Scene {
public:
    uint32_t counter = 0;

    static HittableList get_scene() {
        HittableList scene;
        scene.add(Sphere{{0.80f, 0.5f, -1.0f}, 0.5f, Metal({0.8f, 0.6f, 0.2f}, 0.25f)});
        scene.add(Sphere{{-0.10f, 0.25f, -0.10f}, 0.25f, Lambertian({0.7f, 0.3f, 0.3f})});
        scene.add(Sphere{{-0.80f, 0.5f, -1.0f}, 0.5f, Metal({0.8f, 0.8f, 0.8f}, 0.0f)});
        scene.add(Sphere{{0.25f, 0.75f, -0.5f}, 0.1f, Dielectric({1.0f, 1.0f, 1.0f}, 1.5f)});
        scene.add(Sphere{{-0.25f, 0.75f, -0.42f}, 0.14f, Dielectric({0.95f, 0.95f, 1.0f}, 1.5f)});
        scene.add(Sphere{{-0.25f, 0.75f, -0.42f}, -0.13f, Dielectric({0.95f, 0.95f, 1.0f}, 1.5f)});
        scene.add(Cube{{0.0f, 0.5f, -1.0f}, 0.50f, 0.50f, 0.50f, Lambertian({0.7f, 0.3f, 0.7f})});
        scene.add(Plane{{0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, Lambertian({0.3f, 0.3f, 0.3f}), true});
        return scene;
    }
    Camera get_camera(float aperture = 0.0f, uint32_t samples_per_px = 1) const {
        Camera cam;
        if (rt_scene_camera(counter, aperture, &cam.c) != RT_OK) throw std::runtime_error("rt_scene_camera failed");
        cam.samples_per_px = samples_per_px;
        return cam;
    }
    void move_cam() { counter = (counter + 1) % 100; }
};

// ---- synthetic benchmark scene ------------------------------------------------
// PCG-XSH-RR 64/32 (O'Neill), the generator the benchmark scenes draw from
class Pcg32 {
public:
    explicit Pcg32(uint64_t seed, uint64_t seq = 54) : state_(0), inc_((seq << 1) | 1) {
        next_u32();
        state_ += seed;
        next_u32();
    }
    uint32_t next_u32() {
        const uint64_t old = state_;
        state_ = old * 6364136223846793005ull + inc_;
        const uint32_t xs = (uint32_t)(((old >> 18) ^ old) >> 27);
        const uint32_t rot = (uint32_t)(old >> 59);
        return (xs >> rot) | (xs << ((32 - rot) & 31));
    }
    double uniform(double lo = 0.0, double hi = 1.0) { return lo + (hi - lo) * (double)(next_u32() >> 8) * (1.0 / 16777216.0); }

private:
    uint64_t state_, inc_;
};

// n non-overlapping spheres resting on the plane y = 0: centres uniform in x, z in
// [-extent, extent], radius uniform [0.1, 0.4]; 60 % Lambertian, 30 % Metal (fuzz U[0, 0.5]),
// 10 % Dielectric 1.5; albedo U[0.1, 0.9]^3.  List order: the spheres, then the ground plane
// (Lambertian 0.5 grey).  The same draws in the same order as the Python workload generator.
inline HittableList random_spheres(uint32_t n = 10000, uint64_t seed = 1234, double extent = 30.0) {
    Pcg32 rng(seed);
    std::vector<double> cx, cz, rr;
    HittableList list;
    while (cx.size() < n) {
        const double x = rng.uniform(-extent, extent), z = rng.uniform(-extent, extent), r = rng.uniform(0.1, 0.4);
        bool overlap = false;
        for (size_t k = 0; k < cx.size() && !overlap; ++k) {
            const double d2 = (cx[k] - x) * (cx[k] - x) + (cz[k] - z) * (cz[k] - z) + (rr[k] - r) * (rr[k] - r);
            overlap = d2 < (rr[k] + r) * (rr[k] + r);
        }
        if (overlap) continue;
        cx.push_back(x); cz.push_back(z); rr.push_back(r);
        const double m = rng.uniform();
        const double ar = rng.uniform(0.1, 0.9), ag = rng.uniform(0.1, 0.9), ab = rng.uniform(0.1, 0.9);
        const colour albedo{(float)ar, (float)ag, (float)ab};
        Material mat = Lambertian(albedo);
        if (m >= 0.9) mat = Dielectric(albedo, 1.5f);
        else if (m >= 0.6) mat = Metal(albedo, (float)rng.uniform(0.0, 0.5));
        list.add(Sphere{{(float)x, (float)r, (float)z}, (float)r, mat});
    }
    list.add(Plane{{0.0f, 1.0f, 0.0f}, {0.0f, 0.0f, 0.0f}, Lambertian({0.5f, 0.5f, 0.5f}), true});
    return list;
}
inline Camera random_spheres_camera(uint32_t samples_per_px) {
    const float focus = (float)std::sqrt(0.0 * 0.0 + 6.0 * 6.0 + 25.0 * 25.0);
    return Camera::make(27.0f, 16.0f / 9.0f, 0.0f, focus, {0.0f, 6.0f, 25.0f}, {0.0f, 0.0f, 0.0f}, {0.0f, 1.0f, 0.0f}, samples_per_px);
}

// ---- Wavefront .obj --------------------------------------------------------
// `v x y z` and `f a b c ...` records; indices are 1-based, negative indices
// count back from the last vertex read so far, `a/b/c` forms use the part
// before the first slash, polygons are split as a fan around their first
// vertex.  Everything else is skipped.  Triangles are appended to `list` in
// file order with `material`, each vertex scaled then translated.
inline size_t load_obj(const std::string& path, HittableList& list, const Material& material,
                       float scale = 1.f, vec3 translate = {}) {
    std::ifstream in(path);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::vector<point3> verts;
    std::string line;
    size_t added = 0, lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        std::istringstream ss(line);
        std::string tag;
        if (!(ss >> tag)) continue;
        if (tag == "v") {
            float x, y, z;
            if (!(ss >> x >> y >> z)) throw std::runtime_error(path + ":" + std::to_string(lineno) + ": bad vertex");
            verts.push_back({x * scale + translate.x, y * scale + translate.y, z * scale + translate.z});
        } else if (tag == "f") {
            std::vector<size_t> idx;
            std::string tok;
            while (ss >> tok) {
                long k = std::strtol(tok.c_str(), nullptr, 10);
                long n = (long)verts.size();
                long at = k > 0 ? k - 1 : n + k;
                if (k == 0 || at < 0 || at >= n)
                    throw std::runtime_error(path + ":" + std::to_string(lineno) + ": vertex index out of range");
                idx.push_back((size_t)at);
            }
            if (idx.size() < 3) throw std::runtime_error(path + ":" + std::to_string(lineno) + ": face with < 3 vertices");
            for (size_t k = 1; k + 1 < idx.size(); ++k) {
                list.add(Triangle{verts[idx[0]], verts[idx[k]], verts[idx[k + 1]], material});
                ++added;
            }
        }
    }
    return added;
}

// ---- the three lookup tables (src/denoising/mod.rs:55,59,61) as text files --
inline std::vector<float> load_table(const std::string& path, size_t expect) {
    std::ifstream in(path, std::ios::binary);
    if (!in) throw std::runtime_error("cannot open " + path);
    std::stringstream buf;
    buf << in.rdbuf();
    const std::string text = buf.str();
    std::vector<float> out(expect);
    size_t n = rt_parse_floats(text.data(), text.size(), out.data(), expect);
    if (n != expect) throw std::runtime_error(path + ": expected " + std::to_string(expect) + " values, found " + std::to_string(n));
    return out;
}

}  // namespace rt
