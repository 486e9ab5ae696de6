// rt_host.cpp - host-side helpers of librt_hip.so that need no GPU: camera
// construction, the image-height rule, table text parsing and the PPM writer.
// Built with -ffp-contract=off: the camera numbers must be the ones scalar f32
// code without fused multiply-add produces.
#include <cerrno>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <vector>

#include "rt_hip.h"

namespace {

struct F3 {
    float x, y, z;
};
inline F3 f3(const float a[3]) { return F3{a[0], a[1], a[2]}; }
inline F3 sub(F3 a, F3 b) { return F3{a.x - b.x, a.y - b.y, a.z - b.z}; }
inline F3 scale(F3 a, float t) { return F3{a.x * t, a.y * t, a.z * t}; }
inline F3 over(F3 a, float t) { return F3{a.x / t, a.y / t, a.z / t}; }
inline float norm(F3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); }
inline F3 normalised(F3 a) { return over(a, norm(a)); }
inline F3 cross3(F3 a, F3 b) {
    return F3{a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
inline void put(float dst[3], F3 a) { dst[0] = a.x; dst[1] = a.y; dst[2] = a.z; }

}  // namespace

extern "C" {

// Camera::new, src/camera/mod.rs:35-73; deg_to_rad :30-32
int rt_camera_new(float v_fov_deg, float aspect_ratio, float aperture, float focus_dist,
                  const float origin[3], const float look_at[3], const float v_up[3], rt_camera *out) {
    if (!origin || !look_at || !v_up || !out) return RT_E_BADARG;
    const float pi = 3.14159265358979323846264338327950288f;
    const float theta = v_fov_deg * pi / 180.0f;
    const float half_height = std::tan(theta / 2.0f);
    const float viewport_height = 2.0f * half_height;
    const float viewport_width = aspect_ratio * viewport_height;

    const F3 eye = f3(origin);
    const F3 w = normalised(sub(eye, f3(look_at)));
    const F3 u = normalised(cross3(f3(v_up), w));
    const F3 v = cross3(w, u);

    const F3 horizontal = scale(scale(u, viewport_width), focus_dist);
    const F3 vertical = scale(scale(v, viewport_height), focus_dist);
    const F3 llc = sub(sub(sub(eye, over(horizontal, 2.0f)), over(vertical, 2.0f)), scale(w, focus_dist));

    put(out->origin, eye);
    put(out->lower_left_corner, llc);
    put(out->horizontal, horizontal);
    put(out->vertical, vertical);
    put(out->u, u);
    put(out->v, v);
    out->lens_radius = aperture / 2.0f;
    return RT_OK;
}

// Scene::get_camera, src/scene/mod.rs:59-76: the dolly counter (:87) and the
// aperture (:70, 0.0 with 0.1 commented out) become arguments
int rt_scene_camera(uint32_t counter, float aperture, rt_camera *out) {
    if (!out) return RT_E_BADARG;
    const float eye[3] = {1.0f - 0.0f, 1.30f - 0.0f, 4.0f - (float)counter * 0.1f};
    const float target[3] = {0.25f, 0.60f, -0.50f};
    const float up[3] = {0.0f, 1.0f, 0.0f};
    const float focus = norm(sub(f3(eye), f3(target)));
    return rt_camera_new(27.0f, 16.0f / 9.0f, aperture, focus, eye, target, up, out);
}

// `(image_width as f32 / aspect_ratio) as usize`, src/raytracer.rs:118-120
uint32_t rt_image_height(uint32_t width, float aspect_ratio) {
    const float h = (float)width / aspect_ratio;
    if (!(h > 0.0f)) return 0u;   // NaN and negatives saturate to 0
    if (h >= 4294967296.0f) return 0xffffffffu;
    return (uint32_t)h;
}

// Reads decimal tokens (anything strtof accepts) separated by whitespace,
// commas, brackets or parentheses.  Each token is rounded to nearest binary32
// directly from its decimal text, like the literals of src/denoising/mod.rs:55-61.
size_t rt_parse_floats(const char *text, size_t len, float *out, size_t cap) {
    if (!text) return 0;
    std::vector<char> buf(text, text + len);
    buf.push_back('\0');
    size_t count = 0;
    char *p = buf.data();
    char *end = p + len;
    while (p < end) {
        const char ch = *p;
        if (!((ch >= '0' && ch <= '9') || ch == '-' || ch == '+' || ch == '.')) { ++p; continue; }
        char *next = nullptr;
        errno = 0;
        const float v = std::strtof(p, &next);
        if (next == p) { ++p; continue; }
        if (out && count < cap) out[count] = v;
        ++count;
        p = next;
    }
    return count;
}

// PPM::write_file, src/image/mod.rs:61-67.  Pixels come as 0x00RRGGBB words in
// display order (from_u8_rgb, src/raytracer.rs:41-44).
int rt_write_ppm(const char *path, const uint32_t *packed, uint32_t width, uint32_t height) {
    if (!path || !packed || width == 0 || height == 0) return RT_E_BADARG;
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RT_E_IO;
    bool ok = std::fprintf(f, "P6 %u %u 255\n", width, height) > 0;
    std::vector<unsigned char> line((size_t)width * 3);
    for (uint32_t y = 0; ok && y < height; ++y) {
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t c = packed[(size_t)y * width + x];
            line[3 * x] = (unsigned char)((c >> 16) & 0xffu);
            line[3 * x + 1] = (unsigned char)((c >> 8) & 0xffu);
            line[3 * x + 2] = (unsigned char)(c & 0xffu);
        }
        ok = std::fwrite(line.data(), 1, line.size(), f) == line.size();
    }
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_E_IO;
}

// PNG (8-bit RGB, no filtering, stored deflate blocks): the format of the reference's
// output_images/*.png, for side-by-side viewing.  No compression library is needed.
namespace {
uint32_t crc32_update(uint32_t crc, const unsigned char *p, size_t n) {
    static uint32_t table[256];
    static bool ready = false;
    if (!ready) {
        for (uint32_t i = 0; i < 256; ++i) {
            uint32_t c = i;
            for (int k = 0; k < 8; ++k) c = (c & 1u) ? 0xEDB88320u ^ (c >> 1) : c >> 1;
            table[i] = c;
        }
        ready = true;
    }
    for (size_t i = 0; i < n; ++i) crc = table[(crc ^ p[i]) & 0xffu] ^ (crc >> 8);
    return crc;
}
void put_be32(std::vector<unsigned char> &v, uint32_t x) {
    v.push_back((unsigned char)(x >> 24)); v.push_back((unsigned char)(x >> 16));
    v.push_back((unsigned char)(x >> 8)); v.push_back((unsigned char)x);
}
bool write_chunk(std::FILE *f, const char type[4], const std::vector<unsigned char> &data) {
    std::vector<unsigned char> head;
    put_be32(head, (uint32_t)data.size());
    head.insert(head.end(), type, type + 4);
    uint32_t crc = crc32_update(0xffffffffu, head.data() + 4, 4);
    crc = crc32_update(crc, data.data(), data.size()) ^ 0xffffffffu;
    std::vector<unsigned char> tail;
    put_be32(tail, crc);
    return std::fwrite(head.data(), 1, head.size(), f) == head.size() &&
           (data.empty() || std::fwrite(data.data(), 1, data.size(), f) == data.size()) &&
           std::fwrite(tail.data(), 1, 4, f) == 4;
}
}  // namespace

int rt_write_png(const char *path, const uint32_t *packed, uint32_t width, uint32_t height) {
    if (!path || !packed || width == 0 || height == 0) return RT_E_BADARG;
    // raw image: per row a filter byte 0 then RGB triples
    std::vector<unsigned char> raw;
    try { raw.reserve((size_t)height * ((size_t)width * 3 + 1)); } catch (...) { return RT_E_OOM; }
    for (uint32_t y = 0; y < height; ++y) {
        raw.push_back(0);
        for (uint32_t x = 0; x < width; ++x) {
            const uint32_t c = packed[(size_t)y * width + x];
            raw.push_back((unsigned char)((c >> 16) & 0xffu));
            raw.push_back((unsigned char)((c >> 8) & 0xffu));
            raw.push_back((unsigned char)(c & 0xffu));
        }
    }
    // zlib stream of stored blocks
    std::vector<unsigned char> z;
    z.push_back(0x78); z.push_back(0x01);
    uint32_t a = 1, b = 0;   // adler32
    for (size_t pos = 0; pos < raw.size();) {
        const size_t n = std::min<size_t>(65535, raw.size() - pos);
        z.push_back(pos + n == raw.size() ? 1 : 0);
        z.push_back((unsigned char)(n & 0xff)); z.push_back((unsigned char)(n >> 8));
        z.push_back((unsigned char)(~n & 0xff)); z.push_back((unsigned char)((~n >> 8) & 0xff));
        z.insert(z.end(), raw.begin() + pos, raw.begin() + pos + n);
        for (size_t k = pos; k < pos + n; ++k) { a = (a + raw[k]) % 65521u; b = (b + a) % 65521u; }
        pos += n;
    }
    put_be32(z, (b << 16) | a);
    std::FILE *f = std::fopen(path, "wb");
    if (!f) return RT_E_IO;
    static const unsigned char sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    bool ok = std::fwrite(sig, 1, 8, f) == 8;
    std::vector<unsigned char> ihdr;
    put_be32(ihdr, width); put_be32(ihdr, height);
    ihdr.push_back(8); ihdr.push_back(2); ihdr.push_back(0); ihdr.push_back(0); ihdr.push_back(0);
    ok = ok && write_chunk(f, "IHDR", ihdr) && write_chunk(f, "IDAT", z) && write_chunk(f, "IEND", {});
    ok = (std::fclose(f) == 0) && ok;
    return ok ? RT_OK : RT_E_IO;
}

}  // extern "C"
