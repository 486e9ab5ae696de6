// rt_scene_build.h - host-side construction of the device scene layout.
#ifndef RT_SCENE_BUILD_H
#define RT_SCENE_BUILD_H

#include <string>
#include <vector>

#include "rt_device.h"

struct HostScene {
    std::vector<DevRun> runs;
    std::vector<float4> sph_geom;     // cx, cy, cz, radius              (slot order)
    std::vector<uint32_t> sph_mat, sph_id;
    std::vector<float4> planes, cubes, tris;
    std::vector<uint32_t> tri_id;
    std::vector<float4> cull_rec;     // RUN_FLAT rejection records, 8 float4 per step
    std::vector<float4> tree_rec;     // RUN_TREE records, DEV_NODE_F4 float4 per step
    std::vector<uint32_t> order;      // per tree/flat run: list position -> slot offset
    std::vector<float4> mat_rec;      // 2 per material, see DevScene::mat_rec
    uint32_t need_ff_shared = 0;
    uint32_t cheap_ok = 1;            // see DevScene::cheap_ok
    uint32_t lds_run = 0xffffffffu, lds_level = 0, lds_first = 0, lds_count = 0;
    uint32_t lds_deep_level = 0xffffffffu, lds_deep_first = 0, lds_deep_count = 0;   // see DevScene
    uint32_t lds_kind = 0;
    uint32_t modes = 0;
};

// Validates `s` and fills `out`.  use_tree = false keeps long runs in list order
// (RUN_FLAT for spheres, RUN_EXACT for triangles).  Returns RT_OK or RT_E_BADARG
// with a message in err.
int build_host_scene(const rt_scene *s, bool use_tree, HostScene &out, std::string &err);

#endif
