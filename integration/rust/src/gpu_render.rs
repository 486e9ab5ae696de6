// gpu_render.rs - safe wrapper the reference's main() calls instead of spawning
// calculate_some_pxls workers.  Needs two one-line additions to the reference's
// traits so a `Box<dyn Hittable>` can describe itself:
//
//   pub trait Material { ...; fn describe(&self) -> (i32, colour, f32); }
//       Metal       -> (RT_METAL,      self.albedo, self.fuzz)
//       Lambertian  -> (RT_LAMBERTIAN, self.albedo, 0.0)
//       Dielectric  -> (RT_DIELECTRIC, self.albedo, self.index_of_refraction)
//   pub trait Hittable { ...; fn flatten(&self, id: u32, out: &mut FlatScene); }
//       Sphere -> out.push_sphere(id, self.center, self.radius, self.material.describe())
//       Plane  -> out.push_plane (id, self.normal, self.point,  self.material.describe())
//       Cube   -> out.push_cube  (id, self.center, self.w, self.h, self.d, self.material.describe())
//
// `id` is the position in HittableList.list; it decides ties exactly as the
// insertion-ordered scan does.

use crate::hip_ffi::*;
use std::ffi::CStr;
use std::ptr;

#[derive(Default)]
pub struct FlatScene {
    sph: [Vec<f32>; 4],
    sph_mat: Vec<u32>,
    sph_id: Vec<u32>,
    pl: [Vec<f32>; 6],
    pl_mat: Vec<u32>,
    pl_id: Vec<u32>,
    cb: [Vec<f32>; 6],
    cb_mat: Vec<u32>,
    cb_id: Vec<u32>,
    mat_kind: Vec<i32>,
    mat: [Vec<f32>; 4],
}

impl FlatScene {
    fn push_material(&mut self, m: (i32, [f32; 3], f32)) -> u32 {
        self.mat_kind.push(m.0);
        for k in 0..3 {
            self.mat[k].push(m.1[k]);
        }
        self.mat[3].push(m.2);
        (self.mat_kind.len() - 1) as u32
    }
    pub fn push_sphere(&mut self, id: u32, c: [f32; 3], r: f32, m: (i32, [f32; 3], f32)) {
        let mi = self.push_material(m);
        for k in 0..3 {
            self.sph[k].push(c[k]);
        }
        self.sph[3].push(r);
        self.sph_mat.push(mi);
        self.sph_id.push(id);
    }
    pub fn push_plane(&mut self, id: u32, n: [f32; 3], p: [f32; 3], m: (i32, [f32; 3], f32)) {
        let mi = self.push_material(m);
        for k in 0..3 {
            self.pl[k].push(n[k]);
            self.pl[3 + k].push(p[k]);
        }
        self.pl_mat.push(mi);
        self.pl_id.push(id);
    }
    pub fn push_cube(&mut self, id: u32, c: [f32; 3], w: f32, h: f32, d: f32, m: (i32, [f32; 3], f32)) {
        let mi = self.push_material(m);
        for k in 0..3 {
            self.cb[k].push(c[k]);
        }
        self.cb[3].push(w);
        self.cb[4].push(h);
        self.cb[5].push(d);
        self.cb_mat.push(mi);
        self.cb_id.push(id);
    }
    fn as_c(&self) -> rt_scene {
        rt_scene {
            n_spheres: self.sph_id.len() as u32,
            sph_cx: self.sph[0].as_ptr(), sph_cy: self.sph[1].as_ptr(), sph_cz: self.sph[2].as_ptr(),
            sph_r: self.sph[3].as_ptr(), sph_mat: self.sph_mat.as_ptr(), sph_id: self.sph_id.as_ptr(),
            n_planes: self.pl_id.len() as u32,
            pl_nx: self.pl[0].as_ptr(), pl_ny: self.pl[1].as_ptr(), pl_nz: self.pl[2].as_ptr(),
            pl_px: self.pl[3].as_ptr(), pl_py: self.pl[4].as_ptr(), pl_pz: self.pl[5].as_ptr(),
            pl_mat: self.pl_mat.as_ptr(), pl_id: self.pl_id.as_ptr(),
            n_cubes: self.cb_id.len() as u32,
            cb_cx: self.cb[0].as_ptr(), cb_cy: self.cb[1].as_ptr(), cb_cz: self.cb[2].as_ptr(),
            cb_w: self.cb[3].as_ptr(), cb_h: self.cb[4].as_ptr(), cb_d: self.cb[5].as_ptr(),
            cb_mat: self.cb_mat.as_ptr(), cb_id: self.cb_id.as_ptr(),
            n_tris: 0,
            tri_v0x: ptr::null(), tri_v0y: ptr::null(), tri_v0z: ptr::null(),
            tri_v1x: ptr::null(), tri_v1y: ptr::null(), tri_v1z: ptr::null(),
            tri_v2x: ptr::null(), tri_v2y: ptr::null(), tri_v2z: ptr::null(),
            tri_mat: ptr::null(), tri_id: ptr::null(),
            n_materials: self.mat_kind.len() as u32,
            mat_kind: self.mat_kind.as_ptr(),
            mat_r: self.mat[0].as_ptr(), mat_g: self.mat[1].as_ptr(), mat_b: self.mat[2].as_ptr(),
            mat_param: self.mat[3].as_ptr(),
        }
    }
}

pub struct GpuRenderer {
    ctx: *mut rt_ctx,
}

fn check(ctx: *const rt_ctx, rc: i32) -> Result<(), String> {
    if rc == RT_OK {
        return Ok(());
    }
    let msg = unsafe { CStr::from_ptr(rt_last_error(ctx)) }.to_string_lossy().into_owned();
    Err(format!("rt_hip error {}: {}", rc, msg))
}

impl GpuRenderer {
    /// `disc`, `sphere`, `screen`: the three tables of denoising/mod.rs flattened to f32
    /// (64*2, 256*3, 128*128 values).
    pub fn new(device: i32, disc: &[f32], sphere: &[f32], screen: &[f32]) -> Result<Self, String> {
        assert!(disc.len() == 128 && sphere.len() == 768 && screen.len() == 16384);
        let mut ctx = ptr::null_mut();
        check(ptr::null(), unsafe { rt_create(device, &mut ctx) })?;
        let r = GpuRenderer { ctx };
        check(r.ctx, unsafe { rt_upload_tables(r.ctx, disc.as_ptr(), sphere.as_ptr(), screen.as_ptr()) })?;
        Ok(r)
    }
    pub fn set_scene(&mut self, flat: &FlatScene) -> Result<(), String> {
        let c = flat.as_c();
        check(self.ctx, unsafe { rt_upload_scene(self.ctx, &c) })
    }
    pub fn set_camera(&mut self, cam: &rt_camera) -> Result<(), String> {
        check(self.ctx, unsafe { rt_set_camera(self.ctx, cam) })
    }
    /// One whole frame into `img_buffer` (0x00RRGGBB, display order), the job of the
    /// worker threads plus the receive loop of main().
    pub fn render_frame(&mut self, width: usize, height: usize, spp: u32, max_bounces: u32,
                        gamma: bool, img_buffer: &mut [u32]) -> Result<u64, String> {
        assert!(img_buffer.len() == width * height);
        let p = rt_params {
            width: width as u32, height: height as u32, spp, sample_begin: 0, sample_count: spp,
            max_bounces, gamma: gamma as u32, t_min: 0.001, band_begin: 0, band_stride: 1,
            sampling_mode: RT_SAMPLING_DET, seed: 0,
        };
        check(self.ctx, unsafe { rt_begin_frame(self.ctx, p.width, p.height) })?;
        check(self.ctx, unsafe { rt_render(self.ctx, &p) })?;
        check(self.ctx, unsafe { rt_read_packed(self.ctx, img_buffer.as_mut_ptr(), img_buffer.len()) })?;
        let mut segments = 0u64;
        check(self.ctx, unsafe { rt_ray_count(self.ctx, &mut segments) })?;
        Ok(segments)
    }
    /// The frames of one run (NUM_FRAMES_TO_RENDER cameras, one per frame) in a single call:
    /// frame f is rendered through `cameras[f]` and handed to `show(f, pixels)` in order.  For the
    /// reference's own scene the frames share one kernel launch.
    pub fn render_frames<F: FnMut(usize, &[u32])>(&mut self, width: usize, height: usize, spp: u32,
                                                  max_bounces: u32, gamma: bool, cameras: &[rt_camera],
                                                  mut show: F) -> Result<u64, String> {
        let p = rt_params {
            width: width as u32, height: height as u32, spp, sample_begin: 0, sample_count: spp,
            max_bounces, gamma: gamma as u32, t_min: 0.001, band_begin: 0, band_stride: 1,
            sampling_mode: RT_SAMPLING_DET, seed: 0,
        };
        let n = cameras.len() as u32;
        check(self.ctx, unsafe { rt_begin_frames(self.ctx, p.width, p.height, n) })?;
        check(self.ctx, unsafe { rt_render_frames(self.ctx, &p, cameras.as_ptr(), n) })?;
        let mut pixels = vec![0u32; width * height];
        for f in 0..cameras.len() {
            check(self.ctx, unsafe { rt_select_frame(self.ctx, f as u32) })?;
            check(self.ctx, unsafe { rt_read_packed(self.ctx, pixels.as_mut_ptr(), pixels.len()) })?;
            show(f, &pixels);
        }
        let mut segments = 0u64;
        check(self.ctx, unsafe { rt_ray_count(self.ctx, &mut segments) })?;
        Ok(segments)
    }
}

impl Drop for GpuRenderer {
    fn drop(&mut self) {
        unsafe { rt_destroy(self.ctx) }
    }
}
