// hip_ffi.rs - raw binding of include/rt_hip.h for the reference crate.
// `mod hip_ffi;` next to the other modules in src/raytracer.rs.
// Layouts follow the C header field for field (#[repr(C)]).
#![allow(non_camel_case_types, dead_code)]

use std::os::raw::{c_char, c_int, c_void};

#[repr(C)]
pub struct rt_ctx {
    _private: [u8; 0],
}

pub const RT_OK: c_int = 0;
pub const RT_E_BADARG: c_int = -1;
pub const RT_E_HIP: c_int = -2;
pub const RT_E_OOM: c_int = -3;
pub const RT_E_STATE: c_int = -4;
pub const RT_E_IO: c_int = -5;

pub const RT_METAL: i32 = 0;
pub const RT_LAMBERTIAN: i32 = 1;
pub const RT_DIELECTRIC: i32 = 2;

pub const RT_SAMPLING_DET: u32 = 0;

#[repr(C)]
pub struct rt_scene {
    pub n_spheres: u32,
    pub sph_cx: *const f32,
    pub sph_cy: *const f32,
    pub sph_cz: *const f32,
    pub sph_r: *const f32,
    pub sph_mat: *const u32,
    pub sph_id: *const u32,
    pub n_planes: u32,
    pub pl_nx: *const f32,
    pub pl_ny: *const f32,
    pub pl_nz: *const f32,
    pub pl_px: *const f32,
    pub pl_py: *const f32,
    pub pl_pz: *const f32,
    pub pl_mat: *const u32,
    pub pl_id: *const u32,
    pub n_cubes: u32,
    pub cb_cx: *const f32,
    pub cb_cy: *const f32,
    pub cb_cz: *const f32,
    pub cb_w: *const f32,
    pub cb_h: *const f32,
    pub cb_d: *const f32,
    pub cb_mat: *const u32,
    pub cb_id: *const u32,
    pub n_tris: u32,
    pub tri_v0x: *const f32,
    pub tri_v0y: *const f32,
    pub tri_v0z: *const f32,
    pub tri_v1x: *const f32,
    pub tri_v1y: *const f32,
    pub tri_v1z: *const f32,
    pub tri_v2x: *const f32,
    pub tri_v2y: *const f32,
    pub tri_v2z: *const f32,
    pub tri_mat: *const u32,
    pub tri_id: *const u32,
    pub n_materials: u32,
    pub mat_kind: *const i32,
    pub mat_r: *const f32,
    pub mat_g: *const f32,
    pub mat_b: *const f32,
    pub mat_param: *const f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_camera {
    pub origin: [f32; 3],
    pub lower_left_corner: [f32; 3],
    pub horizontal: [f32; 3],
    pub vertical: [f32; 3],
    pub u: [f32; 3],
    pub v: [f32; 3],
    pub lens_radius: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Default)]
pub struct rt_params {
    pub width: u32,
    pub height: u32,
    pub spp: u32,
    pub sample_begin: u32,
    pub sample_count: u32,
    pub max_bounces: u32,
    pub gamma: u32,
    pub t_min: f32,
    pub band_begin: u32,
    pub band_stride: u32,
    pub sampling_mode: u32,
    pub seed: u32,
}

extern "C" {
    pub fn rt_create(device: c_int, out: *mut *mut rt_ctx) -> c_int;
    pub fn rt_destroy(ctx: *mut rt_ctx);
    pub fn rt_last_error(ctx: *const rt_ctx) -> *const c_char;
    pub fn rt_upload_tables(ctx: *mut rt_ctx, disc: *const f32, sphere: *const f32, screen: *const f32) -> c_int;
    pub fn rt_upload_scene(ctx: *mut rt_ctx, scene: *const rt_scene) -> c_int;
    pub fn rt_set_camera(ctx: *mut rt_ctx, cam: *const rt_camera) -> c_int;
    pub fn rt_begin_frame(ctx: *mut rt_ctx, width: u32, height: u32) -> c_int;
    pub fn rt_begin_frames(ctx: *mut rt_ctx, width: u32, height: u32, n_frames: u32) -> c_int;
    pub fn rt_select_frame(ctx: *mut rt_ctx, frame: u32) -> c_int;
    pub fn rt_render(ctx: *mut rt_ctx, params: *const rt_params) -> c_int;
    pub fn rt_render_frames(ctx: *mut rt_ctx, params: *const rt_params, cameras: *const rt_camera,
                            n_frames: u32) -> c_int;
    /// rt_render on caller-owned device buffers, working memory from the library's fixed
    /// per-GPU scratch: makes no allocation (for callers under graph capture)
    pub fn rt_render_static(ctx: *mut rt_ctx, params: *const rt_params, accum_device: *mut c_void,
                            ids_device: *mut i32, segments_device: *mut u64) -> c_int;
    pub fn rt_pack(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_sync(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_read_accum(ctx: *mut rt_ctx, rgb: *mut f32, n_px: usize) -> c_int;
    pub fn rt_read_sample_counts(ctx: *mut rt_ctx, counts: *mut u32, n_px: usize) -> c_int;
    pub fn rt_read_packed(ctx: *mut rt_ctx, out: *mut u32, n_px: usize) -> c_int;
    pub fn rt_read_primary_ids(ctx: *mut rt_ctx, ids: *mut i32, n_px: usize) -> c_int;
    pub fn rt_ray_count(ctx: *mut rt_ctx, segments: *mut u64) -> c_int;
    /// fan-in of a frame rendered by n_ctx contexts (band_begin = g, band_stride = n_ctx)
    pub fn rt_gather_packed(ctxs: *const *mut rt_ctx, n_ctx: u32, out: *mut u32, n_px: usize) -> c_int;
    pub fn rt_device_buffers(ctx: *mut rt_ctx, accum: *mut *mut c_void, packed: *mut *mut c_void,
                             ids: *mut *mut c_void, stream: *mut *mut c_void) -> c_int;
    pub fn rt_export_packed_bands(ctx: *mut rt_ctx, dst_device: *mut u32, capacity_px: usize,
                                  band_begin: u32, band_stride: u32) -> c_int;
    /// self-check: short division / square-root sequences against `/` and sqrt on the GPU
    pub fn rt_check_arith(ctx: *mut rt_ctx, operands: u64, seed: u32, mismatches: *mut u64) -> c_int;
    pub fn rt_kernel_info(ctx: *mut rt_ctx, max_bounces: u32, scan_modes: *mut u32, depth_budget: *mut u32) -> c_int;
    pub fn rt_build_id() -> *const c_char;
    pub fn rt_timer_start(ctx: *mut rt_ctx) -> c_int;
    pub fn rt_timer_stop(ctx: *mut rt_ctx, milliseconds: *mut f32) -> c_int;
    pub fn rt_camera_new(v_fov_deg: f32, aspect_ratio: f32, aperture: f32, focus_dist: f32,
                         origin: *const f32, look_at: *const f32, v_up: *const f32,
                         out: *mut rt_camera) -> c_int;
    pub fn rt_scene_camera(counter: u32, aperture: f32, out: *mut rt_camera) -> c_int;
    pub fn rt_image_height(width: u32, aspect_ratio: f32) -> u32;
    pub fn rt_parse_floats(text: *const c_char, len: usize, out: *mut f32, cap: usize) -> usize;
    pub fn rt_write_ppm(path: *const c_char, packed: *const u32, width: u32, height: u32) -> c_int;
    pub fn rt_write_png(path: *const c_char, packed: *const u32, width: u32, height: u32) -> c_int;
}
