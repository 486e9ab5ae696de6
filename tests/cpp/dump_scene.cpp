// Dumps what the C++ host mirror (csrc/host/scene.hpp) hands to rt_upload_scene,
// as text, so a test can compare it with the Python mirror field by field.
//   dump_scene [file.obj scale tx ty tz]
#include <cinttypes>
#include <cstdio>
#include <cstdlib>

#include "scene.hpp"

static void fl(const char* name, const float* p, uint32_t n) {
    std::printf("%s", name);
    for (uint32_t i = 0; i < n; ++i) {
        uint32_t bits;
        std::memcpy(&bits, p + i, 4);
        std::printf(" %08" PRIx32, bits);
    }
    std::printf("\n");
}
static void ul(const char* name, const uint32_t* p, uint32_t n) {
    std::printf("%s", name);
    for (uint32_t i = 0; i < n; ++i) std::printf(" %" PRIu32, p[i]);
    std::printf("\n");
}

int main(int argc, char** argv) {
    rt::HittableList list = (argc == 3 && std::string(argv[1]) == "spheres")
                                ? rt::random_spheres((uint32_t)std::strtoul(argv[2], nullptr, 10))
                                : rt::Scene::get_scene();
    if (argc == 6) {
        rt::load_obj(argv[1], list, rt::Metal({0.5f, 0.25f, 0.125f}, 0.3f), std::strtof(argv[2], nullptr),
                     {std::strtof(argv[3], nullptr), std::strtof(argv[4], nullptr), std::strtof(argv[5], nullptr)});
    }
    const rt_scene s = list.flatten();
    fl("sph_cx", s.sph_cx, s.n_spheres); fl("sph_cy", s.sph_cy, s.n_spheres); fl("sph_cz", s.sph_cz, s.n_spheres);
    fl("sph_r", s.sph_r, s.n_spheres); ul("sph_mat", s.sph_mat, s.n_spheres); ul("sph_id", s.sph_id, s.n_spheres);
    fl("pl_nx", s.pl_nx, s.n_planes); fl("pl_ny", s.pl_ny, s.n_planes); fl("pl_nz", s.pl_nz, s.n_planes);
    fl("pl_px", s.pl_px, s.n_planes); fl("pl_py", s.pl_py, s.n_planes); fl("pl_pz", s.pl_pz, s.n_planes);
    ul("pl_mat", s.pl_mat, s.n_planes); ul("pl_id", s.pl_id, s.n_planes);
    fl("cb_cx", s.cb_cx, s.n_cubes); fl("cb_cy", s.cb_cy, s.n_cubes); fl("cb_cz", s.cb_cz, s.n_cubes);
    fl("cb_w", s.cb_w, s.n_cubes); fl("cb_h", s.cb_h, s.n_cubes); fl("cb_d", s.cb_d, s.n_cubes);
    ul("cb_mat", s.cb_mat, s.n_cubes); ul("cb_id", s.cb_id, s.n_cubes);
    fl("tri_v0x", s.tri_v0x, s.n_tris); fl("tri_v0y", s.tri_v0y, s.n_tris); fl("tri_v0z", s.tri_v0z, s.n_tris);
    fl("tri_v1x", s.tri_v1x, s.n_tris); fl("tri_v1y", s.tri_v1y, s.n_tris); fl("tri_v1z", s.tri_v1z, s.n_tris);
    fl("tri_v2x", s.tri_v2x, s.n_tris); fl("tri_v2y", s.tri_v2y, s.n_tris); fl("tri_v2z", s.tri_v2z, s.n_tris);
    ul("tri_mat", s.tri_mat, s.n_tris); ul("tri_id", s.tri_id, s.n_tris);
    ul("mat_kind", reinterpret_cast<const uint32_t*>(s.mat_kind), s.n_materials);
    fl("mat_r", s.mat_r, s.n_materials); fl("mat_g", s.mat_g, s.n_materials); fl("mat_b", s.mat_b, s.n_materials);
    fl("mat_param", s.mat_param, s.n_materials);
    {
        rt::Camera cam = rt::random_spheres_camera(256);
        fl("spheres_camera", reinterpret_cast<const float*>(&cam.c), 19);
    }
    // camera of the dolly at COUNTER = 0, 1, 99 and after wrapping
    rt::Scene world;
    for (int k = 0; k < 101; ++k) {
        if (k == 0 || k == 1 || k == 99 || k == 100) {
            rt::Camera cam = world.get_camera(0.1f, 4);
            std::printf("counter %u", world.counter);
            fl("", reinterpret_cast<const float*>(&cam.c), 19);
        }
        world.move_cam();
    }
    return 0;
}
