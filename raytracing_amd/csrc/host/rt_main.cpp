// rt_main - headless stand-in for the reference's main() (src/raytracer.rs:96-208)
// on top of librt_hip.so: builds the scene and camera, renders
// NUM_FRAMES_TO_RENDER frames with the camera dolly between them
// (Scene::move_cam, src/scene/mod.rs:78-83), and writes each frame as a binary
// PPM (src/image/mod.rs:61-67) instead of blitting it to a window.  The worker
// threads, the channel and the receive loop of the reference are what the
// library call replaces.
//
//   rt_main [--width 600] [--spp 1] [--frames 10] [--aperture 0] [--bounces 10]
//           [--no-gamma] [--seed 0] [--device 0] [--data DIR] [--out PREFIX]
//           [--obj FILE [--obj-scale S] [--obj-at X Y Z]] [--spheres N] [--chunk N] [--quiet] [--png]
//           [--gpus G [--same-device]] [--batch]
//
// PREFIX_0000.ppm ... (.png with --png) are written when --out is given.  --chunk N renders the
// samples of a frame N at a time (progressive); the picture is the same.
// --gpus G: one context and one host thread per GPU (devices --device .. --device+G-1), context g
// renders the 8-row bands b with b % G == g - the reference's `(thread_id..height).step_by(
// num_threads)` (src/raytracer.rs:245) at band height - and rt_gather_packed puts the picture
// together on the first GPU, like the receive loop (:176-185).  --same-device keeps all G contexts
// on --device (a rehearsal of the same code on a one-GPU box); the picture is the same either way.
// --batch (one GPU): all frames go to the GPU in one rt_render_frames call, each with the camera position
// the dolly has reached by then, and are read back afterwards; the pictures are the same.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "scene.hpp"

namespace {

struct Options {
    uint32_t width = 600, spp = 1, frames = 10, bounces = 10, seed = 0, chunk = 0, spheres = 0, gpus = 1;
    int device = 0;
    bool same_device = false, batch = false;
    float aperture = 0.f, obj_scale = 1.f;
    rt::vec3 obj_at{0.f, 0.f, 0.f};
    bool gamma = true, quiet = false, png = false;
    std::string data = "raytracing_amd/data", out, obj;
};

[[noreturn]] void usage(const char* why) {
    std::fprintf(stderr, "rt_main: %s\n", why);
    std::fprintf(stderr,
                 "usage: rt_main [--width W] [--spp N] [--frames F] [--aperture A] [--bounces B] [--no-gamma]\n"
                 "               [--seed S] [--device D] [--data DIR] [--out PREFIX] [--chunk N] [--quiet] [--png]\n"
                 "               [--obj FILE [--obj-scale S] [--obj-at X Y Z]] [--spheres N] [--gpus G [--same-device]] [--batch]\n");
    std::exit(2);
}

Options parse(int argc, char** argv) {
    Options o;
    auto need = [&](int& i, int n) {
        if (i + n >= argc) usage((std::string("missing value after ") + argv[i]).c_str());
    };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a == "--width") { need(i, 1); o.width = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--spp") { need(i, 1); o.spp = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--frames") { need(i, 1); o.frames = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--bounces") { need(i, 1); o.bounces = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--seed") { need(i, 1); o.seed = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--spheres") { need(i, 1); o.spheres = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--chunk") { need(i, 1); o.chunk = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--device") { need(i, 1); o.device = std::atoi(argv[++i]); }
        else if (a == "--gpus") { need(i, 1); o.gpus = (uint32_t)std::strtoul(argv[++i], nullptr, 10); }
        else if (a == "--same-device") o.same_device = true;
        else if (a == "--batch") o.batch = true;
        else if (a == "--aperture") { need(i, 1); o.aperture = std::strtof(argv[++i], nullptr); }
        else if (a == "--no-gamma") o.gamma = false;
        else if (a == "--quiet") o.quiet = true;
        else if (a == "--png") o.png = true;
        else if (a == "--data") { need(i, 1); o.data = argv[++i]; }
        else if (a == "--out") { need(i, 1); o.out = argv[++i]; }
        else if (a == "--obj") { need(i, 1); o.obj = argv[++i]; }
        else if (a == "--obj-scale") { need(i, 1); o.obj_scale = std::strtof(argv[++i], nullptr); }
        else if (a == "--obj-at") {
            need(i, 3);
            o.obj_at.x = std::strtof(argv[++i], nullptr);
            o.obj_at.y = std::strtof(argv[++i], nullptr);
            o.obj_at.z = std::strtof(argv[++i], nullptr);
        } else usage(("unknown option " + a).c_str());
    }
    if (o.width < 2 || o.spp == 0) usage("--width must be >= 2 and --spp >= 1");
    if (o.gpus == 0 || o.gpus > 64) usage("--gpus must be in 1..64");
    if (o.batch && (o.gpus != 1 || o.chunk != 0 || o.frames == 0)) usage("--batch renders whole frames on one GPU");
    return o;
}

void check(rt_ctx* ctx, int rc, const char* what) {
    if (rc == RT_OK) return;
    std::fprintf(stderr, "rt_main: %s failed (%d): %s\n", what, rc, rt_last_error(ctx));
    std::exit(1);
}

}  // namespace

int main(int argc, char** argv) {
    const Options o = parse(argc, argv);
    try {
        // IMAGE (src/raytracer.rs:118-120)
        const float aspect_ratio = 16.0f / 9.0f;
        const uint32_t image_width = o.width;
        const uint32_t image_height = rt_image_height(image_width, aspect_ratio);
        if (image_height < 2) usage("image too small");
        std::vector<uint32_t> img_buffer((size_t)image_width * image_height);

        const std::vector<float> disc = rt::load_table(o.data + "/disc_blue_noise_64.txt", 128);
        const std::vector<float> sphere = rt::load_table(o.data + "/spherical_blue_noise_256.txt", 768);
        const std::vector<float> screen = rt::load_table(o.data + "/screenspace_blue_noise_128x128.txt", 16384);

        rt::Scene world;
        // --spheres N: the synthetic benchmark scene (N random spheres on a plane, fixed camera)
        // instead of the reference's scene and dolly
        rt::HittableList scene = o.spheres ? rt::random_spheres(o.spheres) : rt::Scene::get_scene();
        if (!o.obj.empty()) {
            size_t n = rt::load_obj(o.obj, scene, rt::Lambertian({0.2f, 0.25f, 0.6f}), o.obj_scale, o.obj_at);
            if (!o.quiet) std::printf("INFO: %zu triangles from %s\n", n, o.obj.c_str());
        }

        const rt_scene flat = scene.flatten();
        // one context per GPU, set up by a host thread of its own (the reference spawns one worker
        // per core, src/raytracer.rs:146-171)
        const uint32_t G = o.gpus;
        std::vector<rt_ctx*> ctx(G, nullptr);
        std::vector<int> rc(G, RT_OK);
        std::vector<std::string> failed(G);
        auto on_every_gpu = [&](auto&& body) {
            std::vector<std::thread> workers;
            for (uint32_t g = 0; g < G; ++g) workers.emplace_back([&, g] { body(g); });
            for (auto& w : workers) w.join();
            for (uint32_t g = 0; g < G; ++g)
                if (rc[g] != RT_OK) {
                    std::fprintf(stderr, "rt_main: GPU %u: %s\n", g, failed[g].c_str());
                    std::exit(1);
                }
        };
        // the error text is read on the thread that made the call (the text of a failed rt_create is per thread)
        auto step = [&](uint32_t g, int code, const char* what) {
            if (rc[g] == RT_OK && code != RT_OK) {
                rc[g] = code;
                failed[g] = std::string(what) + " failed (" + std::to_string(code) + "): " + rt_last_error(ctx[g]);
            }
            return rc[g] == RT_OK;
        };
        on_every_gpu([&](uint32_t g) {
            const int device = o.same_device ? o.device : o.device + (int)g;
            step(g, rt_create(device, &ctx[g]), "rt_create") &&
                step(g, rt_upload_tables(ctx[g], disc.data(), sphere.data(), screen.data()), "rt_upload_tables") &&
                step(g, rt_upload_scene(ctx[g], &flat), "rt_upload_scene");
        });

        uint64_t total_segments = 0;
        double total_ms = 0.0;
        if (o.batch) {
            // the ten frames of the reference's run (src/raytracer.rs:165-168) as one call: the camera of
            // frame f is where Scene::move_cam has taken it after f frames
            std::vector<rt_camera> cams;
            uint32_t spp = o.spp;
            for (uint32_t frame = 0; frame < o.frames; ++frame) {
                const rt::Camera cam = o.spheres ? rt::random_spheres_camera(o.spp) : world.get_camera(o.aperture, o.spp);
                cams.push_back(cam.c);
                spp = cam.samples_per_px;
                world.move_cam();
            }
            rt_params p;
            std::memset(&p, 0, sizeof p);
            p.width = image_width; p.height = image_height;
            p.spp = spp; p.sample_begin = 0; p.sample_count = spp;
            p.max_bounces = o.bounces; p.gamma = o.gamma ? 1u : 0u; p.t_min = 0.001f;
            p.band_begin = 0; p.band_stride = 1;
            p.sampling_mode = RT_SAMPLING_DET; p.seed = o.seed;
            const auto start_time = std::chrono::steady_clock::now();
            check(ctx[0], rt_begin_frames(ctx[0], image_width, image_height, o.frames), "rt_begin_frames");
            check(ctx[0], rt_render_frames(ctx[0], &p, cams.data(), o.frames), "rt_render_frames");
            check(ctx[0], rt_ray_count(ctx[0], &total_segments), "rt_ray_count");
            total_ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start_time).count();
            for (uint32_t frame = 0; frame < o.frames; ++frame) {
                check(ctx[0], rt_select_frame(ctx[0], frame), "rt_select_frame");
                check(ctx[0], rt_read_packed(ctx[0], img_buffer.data(), img_buffer.size()), "rt_read_packed");
                if (o.out.empty()) continue;
                char name[32];
                std::snprintf(name, sizeof name, "_%04u.%s", frame, o.png ? "png" : "ppm");
                const std::string path = o.out + name;
                if (o.png) check(ctx[0], rt_write_png(path.c_str(), img_buffer.data(), image_width, image_height), "rt_write_png");
                else check(ctx[0], rt_write_ppm(path.c_str(), img_buffer.data(), image_width, image_height), "rt_write_ppm");
            }
            if (!o.quiet)
                std::printf("%u frames in one call: %.3f ms, %.1f Mrays/s\n", o.frames, total_ms, (double)total_segments / total_ms * 1e-3);
        }
        for (uint32_t frame = 0; frame < o.frames && !o.batch; ++frame) {
            const rt::Camera cam = o.spheres ? rt::random_spheres_camera(o.spp) : world.get_camera(o.aperture, o.spp);
            const auto start_time = std::chrono::steady_clock::now();
            std::vector<uint64_t> segments(G, 0);
            on_every_gpu([&](uint32_t g) {
                rt_params p;
                std::memset(&p, 0, sizeof p);
                p.width = image_width; p.height = image_height;
                p.spp = cam.samples_per_px;
                p.max_bounces = o.bounces; p.gamma = o.gamma ? 1u : 0u; p.t_min = 0.001f;
                p.band_begin = g; p.band_stride = G;
                p.sampling_mode = RT_SAMPLING_DET; p.seed = o.seed;
                if (!step(g, rt_set_camera(ctx[g], &cam.c), "rt_set_camera") ||
                    !step(g, rt_begin_frame(ctx[g], image_width, image_height), "rt_begin_frame"))
                    return;
                const uint32_t chunk = o.chunk ? o.chunk : p.spp;
                for (uint32_t s = 0; s < p.spp; s += chunk) {
                    p.sample_begin = s;
                    p.sample_count = (p.spp - s < chunk) ? p.spp - s : chunk;
                    if (!step(g, rt_render(ctx[g], &p), "rt_render")) return;
                }
                step(g, rt_ray_count(ctx[g], &segments[g]), "rt_ray_count");   // waits for this GPU's share
            });
            if (G == 1) check(ctx[0], rt_read_packed(ctx[0], img_buffer.data(), img_buffer.size()), "rt_read_packed");
            else check(ctx[0], rt_gather_packed(ctx.data(), G, img_buffer.data(), img_buffer.size()), "rt_gather_packed");
            const double ms =
                std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - start_time).count();
            uint64_t frame_segments = 0;
            for (uint64_t n : segments) frame_segments += n;
            total_segments += frame_segments;
            total_ms += ms;
            if (!o.out.empty()) {
                char name[32];
                std::snprintf(name, sizeof name, "_%04u.%s", frame, o.png ? "png" : "ppm");
                const std::string path = o.out + name;
                if (o.png) check(ctx[0], rt_write_png(path.c_str(), img_buffer.data(), image_width, image_height), "rt_write_png");
                else check(ctx[0], rt_write_ppm(path.c_str(), img_buffer.data(), image_width, image_height), "rt_write_ppm");
            }
            if (!o.quiet)
                std::printf("frame %u: COUNTER %u, %.3f ms, FPS: %.2f, %.1f Mrays/s\n", frame, world.counter, ms,
                            1000.0 / ms, (double)frame_segments / ms * 1e-3);
            world.move_cam();
        }
        std::printf("{\"frames\": %u, \"width\": %u, \"height\": %u, \"spp\": %u, \"gpus\": %u, \"segments\": %llu, \"ms\": %.3f}\n",
                    o.frames, image_width, image_height, o.spp, G, (unsigned long long)total_segments, total_ms);
        for (rt_ctx* c : ctx) rt_destroy(c);
    } catch (const std::exception& e) {
        std::fprintf(stderr, "rt_main: %s\n", e.what());
        return 1;
    }
    return 0;
}
