//! Raw bindings to `librtx.so`, declaration by declaration after `include/rtx.h`
//! (ABI version 3), plus `Context`, a small owner type for the context pointer.
//!
//! Written against the header; it could not be compiled where it was written (no Rust
//! toolchain there).  `tests/test_abi.py` checks that the functions and struct fields named
//! here are the header's.
#![allow(non_camel_case_types)]
use std::ffi::CStr;
use std::os::raw::{c_char, c_int, c_void};

pub const RTX_ABI_VERSION: u32 = 5;

pub const RTX_OK: c_int = 0;
pub const RTX_ERR_INVALID_ARG: c_int = -1;
pub const RTX_ERR_INVALID_SCENE: c_int = -2;
pub const RTX_ERR_DEVICE: c_int = -3;
pub const RTX_ERR_NO_SCENE: c_int = -4;
pub const RTX_ERR_UNSUPPORTED: c_int = -5;
pub const RTX_ERR_NO_MEMORY: c_int = -6;

pub const RTX_PLANE: u32 = 0;
pub const RTX_ELLIPSOID: u32 = 1;
pub const RTX_BOX: u32 = 2;
pub const RTX_DIFFUSE: u32 = 0;
pub const RTX_METALLIC: u32 = 1;
pub const RTX_DIELECTRIC: u32 = 2;
pub const RTX_MODE_MEGAKERNEL: u32 = 0;
pub const RTX_MODE_WAVEFRONT: u32 = 1;
pub const RTX_ACCEL_AUTO: u32 = 0;
pub const RTX_ACCEL_BRUTE_LDS: u32 = 1;
pub const RTX_ACCEL_BRUTE_SCALAR: u32 = 2;
pub const RTX_ACCEL_BVH: u32 = 3;
pub const RTX_RNG_PHILOX: u32 = 0;
pub const RTX_RNG_CONST: u32 = 1;
pub const RTX_REDUCE_P2P: u32 = 0;
pub const RTX_REDUCE_RCCL: u32 = 1;
pub const RTX_MAX_DEVICES: usize = 16;
pub const RTX_STATIC_MAX_TILES: u32 = 262144;

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_prim {
    pub kind: u32,
    pub param: [f32; 3],
    pub position: [f32; 3],
    pub rotation: [f32; 4],
    pub color: [f32; 3],
    pub emission: [f32; 3],
    pub material: u32,
    pub ior: f32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtx_scene {
    pub width: u32,
    pub height: u32,
    pub ray_depth: u32,
    pub n_samples: u32,
    pub background: [f32; 3],
    pub cam_pos: [f32; 3],
    pub cam_axis: [f32; 9],
    pub tg_fov_x: f32,
    pub tg_fov_y: f32,
    pub n_prims: u32,
    pub prims: *const rtx_prim,
}

#[repr(C)]
#[derive(Clone, Copy, Debug)]
pub struct rtx_opts {
    pub seed: u64,
    pub mode: u32,
    pub accel: u32,
    pub rng_mode: u32,
    pub const_words: [u32; 4],
    pub device: i32,
    pub collect_stats: u32,
    pub n_devices: i32,
    pub reduce: u32,
    pub devices: *const i32,
}

impl Default for rtx_opts {
    /// One GPU (the current device), megakernel, automatic acceleration structure, seed 1.
    fn default() -> Self {
        rtx_opts { seed: 1, mode: RTX_MODE_MEGAKERNEL, accel: RTX_ACCEL_AUTO, rng_mode: RTX_RNG_PHILOX,
                   const_words: [0; 4], device: -1, collect_stats: 0, n_devices: 0, reduce: RTX_REDUCE_P2P,
                   devices: std::ptr::null() }
    }
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct rtx_stats {
    pub samples: u64,
    pub rays: u64,
    pub hits: [u64; 3],
    pub cosine_samples: u64,
    pub light_samples: [u64; 3],
    pub pdf_evals: u64,
    pub light_pdf: [u64; 3],
    pub diffuse_bounces: u64,
    pub metallic_bounces: u64,
    pub dielectric_bounces: u64,
    pub prim_tests: [u64; 3],
    pub light_tests: [u64; 3],
    pub bvh_node_tests: u64,
}

#[repr(C)]
pub struct rtx_ctx {
    _private: [u8; 0],
}

extern "C" {
    pub fn rtx_create(out: *mut *mut rtx_ctx, opts: *const rtx_opts) -> c_int;
    pub fn rtx_destroy(ctx: *mut rtx_ctx);
    pub fn rtx_set_scene(ctx: *mut rtx_ctx, scene: *const rtx_scene) -> c_int;
    pub fn rtx_render(ctx: *mut rtx_ctx, scene: *const rtx_scene, rgb: *mut f32) -> c_int;
    pub fn rtx_render_device(ctx: *mut rtx_ctx, n_samples: u32, d_rgb: *mut f32) -> c_int;
    pub fn rtx_render_samples(ctx: *mut rtx_ctx, sample_first: u32, sample_count: u32, d_sum: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rtx_render_samples_static(ctx: *mut rtx_ctx, sample_first: u32, sample_count: u32, d_sum: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rtx_set_share(ctx: *mut rtx_ctx, share_index: u32, share_count: u32) -> c_int;
    pub fn rtx_accumulate_samples(ctx: *mut rtx_ctx, sample_first: u32, sample_count: u32, d_sum: *mut f32, hip_stream: *mut c_void) -> c_int;
    pub fn rtx_divide(ctx: *mut rtx_ctx, d_rgb: *mut f32, n_floats: u64, divisor: f32, hip_stream: *mut c_void) -> c_int;
    pub fn rtx_tonemap_bytes(ctx: *mut rtx_ctx, d_sum: *const f32, n_floats: u64, divisor: f32, d_bytes: *mut u8, hip_stream: *mut c_void) -> c_int;
    pub fn rtx_render_bytes(ctx: *mut rtx_ctx, scene: *const rtx_scene, bytes: *mut u8) -> c_int;
    pub fn rtx_progress_begin(ctx: *mut rtx_ctx, scene: *const rtx_scene, sum: *const f32, samples_done: u32) -> c_int;
    pub fn rtx_progress_step(ctx: *mut rtx_ctx, sample_count: u32) -> c_int;
    pub fn rtx_progress_read(ctx: *mut rtx_ctx, rgb_mean: *mut f32, sum: *mut f32, bytes: *mut u8, samples_done: *mut u32) -> c_int;
    pub fn rtx_primary(ctx: *mut rtx_ctx, scene: *const rtx_scene, hit_id: *mut i32, t: *mut f32, normal: *mut f32, inside: *mut u8) -> c_int;
    pub fn rtx_get_stats(ctx: *mut rtx_ctx, out: *mut rtx_stats) -> c_int;
    pub fn rtx_last_kernel_ms(ctx: *mut rtx_ctx, ms: *mut f32) -> c_int;
    pub fn rtx_get_lights(ctx: *mut rtx_ctx, indices: *mut i32, capacity: i32) -> c_int;
    pub fn rtx_device_count(ctx: *mut rtx_ctx) -> c_int;
    pub fn rtx_alloc_pinned(out: *mut *mut c_void, bytes: u64) -> c_int;
    pub fn rtx_free_pinned(p: *mut c_void);
    pub fn rtx_last_error() -> *const c_char;
    pub fn rtx_abi_version() -> u32;
}

/// The calling thread's last error message.
pub fn last_error() -> String {
    unsafe { CStr::from_ptr(rtx_last_error()).to_string_lossy().into_owned() }
}

/// Owner of an `rtx_ctx`: one host thread at a time; one GPU, or `opts.n_devices` of them.
pub struct Context {
    raw: *mut rtx_ctx,
}

impl Context {
    /// All `n` GPUs 0..n of the node behind one context: `render` deals the frame out to them (same bits as one GPU).
    pub fn with_gpus(n: i32, seed: u64) -> Result<Context, String> {
        let opts = rtx_opts { seed, n_devices: n, ..Default::default() };
        Context::new(&opts)
    }

    pub fn new(opts: &rtx_opts) -> Result<Context, String> {
        unsafe {
            if rtx_abi_version() != RTX_ABI_VERSION {
                return Err(format!("librtx.so has ABI version {}, these bindings {}", rtx_abi_version(), RTX_ABI_VERSION));
            }
            let mut raw = std::ptr::null_mut();
            if rtx_create(&mut raw, opts) != RTX_OK {
                return Err(last_error());
            }
            Ok(Context { raw })
        }
    }

    /// `render(&mut scene)`: mean linear radiance, `width * height * 3` floats, row 0 = top.
    /// `prims` must be the slice `scene.prims` points into.
    pub fn render(&mut self, scene: &rtx_scene, prims: &[rtx_prim], rgb: &mut [f32]) -> Result<(), String> {
        assert_eq!(prims.len(), scene.n_prims as usize);
        assert_eq!(prims.as_ptr(), scene.prims);
        assert_eq!(rgb.len(), scene.width as usize * scene.height as usize * 3);
        match unsafe { rtx_render(self.raw, scene, rgb.as_mut_ptr()) } {
            RTX_OK => Ok(()),
            _ => Err(last_error()),
        }
    }

    pub fn as_ptr(&mut self) -> *mut rtx_ctx {
        self.raw
    }
}

impl Drop for Context {
    fn drop(&mut self) {
        unsafe { rtx_destroy(self.raw) }
    }
}
