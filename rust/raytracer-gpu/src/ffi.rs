//! `include/rt_amd.h`, declaration by declaration.  tests/test_rust_binding_cpu.py of the backend's repository parses
//! both files and fails when a struct field, a function, an argument or a constant differs.
#![allow(non_camel_case_types)]
use std::os::raw::{c_char, c_int, c_void};

pub const RT_OK: c_int = 0;
pub const RT_ERR_INVALID: c_int = -1;
pub const RT_ERR_STATE: c_int = -2;
pub const RT_ERR_DEVICE: c_int = -3;
pub const RT_ERR_LIMIT: c_int = -4;
pub const RT_NO_HIT: u32 = 0xFFFFFFFF;
pub const RT_MAX_LEVELS: u32 = 32;
pub const RT_NOALLOC_MAX_RAYS: u32 = 65536;

pub const RT_MAT_LAMBERTIAN: u32 = 0;
pub const RT_MAT_METAL: u32 = 1;
pub const RT_MAT_DIELECTRIC: u32 = 2;
pub const RT_MAT_DIFFUSE_LIGHT: u32 = 3;
pub const RT_MAT_ISOTROPIC: u32 = 4;

pub const RT_TEX_SOLID: u32 = 0;
pub const RT_TEX_CHECKER: u32 = 1;
pub const RT_TEX_IMAGE: u32 = 2;
pub const RT_TEX_PERLIN: u32 = 3;

pub const RT_XF_TRANSLATE: u32 = 0;
pub const RT_XF_ROTATE_Y: u32 = 1;

pub const RT_NODE_BVH: u8 = 0;
pub const RT_NODE_MEDIUM: u8 = 1;
pub const RT_NODE_SPHERES: u8 = 2;
pub const RT_NODE_QUADS: u8 = 3;
pub const RT_NODE_TRIANGLES: u8 = 4;
pub const RT_NODE_QUADRICS: u8 = 5;
pub const RT_NODE_CULL: u8 = 6;

pub const RT_FLAG_COUNTERS: u32 = 1;
pub const RT_FLAG_NO_CULL: u32 = 2;
pub const RT_FLAG_WHOLE_PATHS: u32 = 4;
pub const RT_FLAG_QUEUES: u32 = 8;
pub const RT_FLAG_SPLIT_BOUNCE: u32 = 16;
pub const RT_FLAG_FAST_BVH: u32 = 32;

/// Opaque `rt_scene`.
#[repr(C)]
pub struct RtScene {
    _private: [u8; 0],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtMaterial {
    pub kind: u32,
    pub texture: u32,
    pub albedo: [f64; 3],
    pub param: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtTexture {
    pub kind: u32,
    pub a: u32,
    pub b: u32,
    pub reserved: u32,
    pub color: [f64; 3],
    pub scale: f64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default, PartialEq)]
pub struct RtXformOp {
    pub kind: u32,
    pub reserved: u32,
    pub v: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtChain {
    pub first_op: u32,
    pub n_ops: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtMedium {
    pub neg_inv_density: f64,
    pub material: u32,
    pub prim_id: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtNode {
    pub kind: u8,
    pub level: u8,
    pub chain: u16,
    pub skip: u32,
    pub first: u32,
    pub count: u32,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtCamera {
    pub width: u32,
    pub height: u32,
    pub pixel00: [f64; 3],
    pub delta_u: [f64; 3],
    pub delta_v: [f64; 3],
    pub center: [f64; 3],
    pub defocus_angle: f64,
    pub disk_u: [f64; 3],
    pub disk_v: [f64; 3],
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtRenderOpts {
    pub spp_begin: u32,
    pub spp_count: u32,
    pub max_depth: u32,
    pub flags: u32,
    pub seed: u64,
    pub background: [f64; 3],
    pub batch_paths: u64,
}

#[repr(C)]
#[derive(Clone, Copy, Debug, Default)]
pub struct RtStats {
    pub paths: u64,
    pub rays: u64,
    pub node_tests: u64,
    pub prim_tests: u64,
    pub cull_tests: u64,
    pub prim_bytes: u64,
    pub prim_flops: u64,
    pub ms_total: f64,
    pub ms_traverse: f64,
    pub ms_shade: f64,
    pub ms_paths: f64,
    pub ms_bounce: f64,
    pub ms_finish: f64,
    pub traverse_launches: u32,
    pub path_launches: u32,
    pub bounce_launches: u32,
    pub bounces: u32,
    pub lockstep: u32,
    pub finish_launches: u32,
    pub list_rays: u64,
    pub wide_tests: u64,
    pub shade_quads: u64,
}

#[link(name = "rt_amd")]
extern "C" {
    pub fn rt_scene_create() -> *mut RtScene;
    pub fn rt_scene_destroy(s: *mut RtScene);
    pub fn rt_last_error() -> *const c_char;
    pub fn rt_scene_set_materials(s: *mut RtScene, m: *const RtMaterial, n: u32) -> c_int;
    pub fn rt_scene_set_textures(s: *mut RtScene, t: *const RtTexture, n: u32) -> c_int;
    pub fn rt_scene_set_image(s: *mut RtScene, slot: u32, rgb8: *const u8, w: u32, h: u32) -> c_int;
    pub fn rt_scene_set_perlin(s: *mut RtScene, slot: u32, vec256x3: *const f64, perm3x256: *const u32) -> c_int;
    pub fn rt_scene_set_spheres(s: *mut RtScene, center_radius: *const f64, material: *const u32, prim_id: *const u32, n: u32) -> c_int;
    pub fn rt_scene_set_quads(s: *mut RtScene, rec16: *const f64, material: *const u32, prim_id: *const u32, n: u32) -> c_int;
    pub fn rt_scene_set_triangles(s: *mut RtScene, rec16: *const f64, material: *const u32, prim_id: *const u32, n: u32) -> c_int;
    pub fn rt_scene_set_quadrics(s: *mut RtScene, rec10: *const f64, material: *const u32, prim_id: *const u32, n: u32) -> c_int;
    pub fn rt_scene_set_media(s: *mut RtScene, m: *const RtMedium, n: u32) -> c_int;
    pub fn rt_scene_set_xforms(s: *mut RtScene, ops: *const RtXformOp, n_ops: u32, chains: *const RtChain, n_chains: u32) -> c_int;
    pub fn rt_scene_set_nodes(s: *mut RtScene, nodes: *const RtNode, n_nodes: u32, boxes6: *const f64, n_boxes: u32) -> c_int;
    pub fn rt_scene_set_cull_boxes(s: *mut RtScene, tight6: *const f64, n_boxes: u32) -> c_int;
    pub fn rt_scene_commit(s: *mut RtScene, device: c_int) -> c_int;
    pub fn rt_scene_set_fast_bvh(s: *mut RtScene, on: c_int) -> c_int;
    pub fn rt_intersect_batch(s: *const RtScene, seed: u64, o3: *const f64, d3: *const f64, tmin: *const f64, tmax: *const f64, rng3: *const u32, n: u64, hit: *mut u8, prim: *mut u32, t: *mut f64, p3: *mut f64, n3: *mut f64, uv2: *mut f64, front: *mut u8, ctr_out: *mut u32) -> c_int;
    pub fn rt_intersect_batch_noalloc(s: *const RtScene, seed: u64, od6_dev: *const f64, tmin_dev: *const f64, tmax_dev: *const f64, rng3_dev: *mut u32, n: u32, hit_dev: *mut u8, prim_dev: *mut u32, t_dev: *mut f64, p3_dev: *mut f64, n3_dev: *mut f64, uv2_dev: *mut f64, front_dev: *mut u8) -> c_int;
    pub fn rt_render(s: *const RtScene, cam: *const RtCamera, opts: *const RtRenderOpts, film_rgb: *mut f64, stats: *mut RtStats) -> c_int;
    pub fn rt_render_device(s: *const RtScene, cam: *const RtCamera, opts: *const RtRenderOpts, film_rgb_device: *mut c_void, stats: *mut RtStats) -> c_int;
    pub fn rt_render_multi(scenes: *const *const RtScene, n: c_int, cam: *const RtCamera, opts: *const RtRenderOpts, film_rgb: *mut f64, stats: *mut RtStats) -> c_int;
    pub fn rt_multi_rccl_reductions() -> u64;
    pub fn rt_debug_device_allocs() -> u64;
    pub fn rt_multi_shutdown();
    pub fn rt_trace_samples(s: *const RtScene, cam: *const RtCamera, opts: *const RtRenderOpts, n: u64, pixel: *const u32, sample: *const u32, radiance3: *mut f64, nrays: *mut u32) -> c_int;
    pub fn rt_tonemap_rgb8(s: *const RtScene, film_rgb: *const f64, width: u32, height: u32, spp: u32, rgb8: *mut u8) -> c_int;
    pub fn rt_debug_arith(device: c_int, a: *const f64, b: *const f64, n: u64, quot: *mut f64, root: *mut f64, recip: *mut f64, logv: *mut f64) -> c_int;
}
