//! `SceneBuilder` collects what the reference's object tree writes into it (see `flatten`) as the flat arrays of
//! `include/rt_amd.h`; `commit` hands them to `librt_amd.so` and returns a `GpuScene` that can render.
//!
//! Layout rules (the same as the C++ host layer of the backend, `rt_host.cpp`):
//! * the tree is laid out in the order the reference's recursive `hit()` visits it (`hittable_list.rs:88-102`,
//!   `bvh.rs:50-57`); a `HittableSlice` is its children in sequence, `Translate` / `RotateY` change the current
//!   chain, `BVHNode` and `ConstantMedium` open an entry whose `skip` is the first entry after their subtree;
//! * consecutive primitives of one type at one level and chain merge into one run;
//! * primitive ids are handed out in depth-first order, first visit wins (an `Arc` shared by two parents keeps one id);
//! * materials and textures are interned by `Arc` pointer.
use crate::ffi::*;
use std::collections::HashMap;
use std::ffi::CStr;
use std::os::raw::c_int;

#[derive(Debug)]
pub struct GpuError {
    pub code: c_int,
    pub message: String,
}
impl std::fmt::Display for GpuError {
    fn fmt(&self, f: &mut std::fmt::Formatter<'_>) -> std::fmt::Result {
        write!(f, "rt_amd error {}: {}", self.code, self.message)
    }
}
impl std::error::Error for GpuError {}

fn check(code: c_int) -> Result<(), GpuError> {
    if code == RT_OK {
        return Ok(());
    }
    let message = unsafe { CStr::from_ptr(rt_last_error()) }.to_string_lossy().into_owned();
    Err(GpuError { code, message })
}

#[derive(Default)]
pub struct SceneBuilder {
    pub materials: Vec<RtMaterial>,
    pub textures: Vec<RtTexture>,
    pub images: Vec<(Vec<u8>, u32, u32)>,
    pub perlins: Vec<(Vec<f64>, Vec<u32>)>,
    pub spheres: Vec<f64>,
    pub quads: Vec<f64>,
    pub triangles: Vec<f64>,
    pub quadrics: Vec<f64>,
    pub sphere_mat: Vec<u32>,
    pub sphere_id: Vec<u32>,
    pub quad_mat: Vec<u32>,
    pub quad_id: Vec<u32>,
    pub tri_mat: Vec<u32>,
    pub tri_id: Vec<u32>,
    pub quadric_mat: Vec<u32>,
    pub quadric_id: Vec<u32>,
    pub media: Vec<RtMedium>,
    pub ops: Vec<RtXformOp>,
    pub chains: Vec<RtChain>,
    pub nodes: Vec<RtNode>,
    pub boxes: Vec<f64>,
    pub cull_boxes: Vec<f64>,
    level: u32,
    in_medium: bool,
    slice_depth: u32,
    merge_barrier: usize,
    chain_stack: Vec<RtXformOp>,
    chain_ids: Vec<(Vec<RtXformOp>, u32)>,
    prim_ids: HashMap<usize, u32>,
    material_ids: HashMap<usize, u32>,
    texture_ids: HashMap<usize, u32>,
    n_prim_ids: u32,
}

/// Address of the object behind a (possibly fat) reference: the key objects are interned by.
pub fn address_of<T: ?Sized>(r: &T) -> usize {
    r as *const T as *const u8 as usize
}

impl SceneBuilder {
    pub fn new() -> SceneBuilder {
        let mut b = SceneBuilder::default();
        b.chains.push(RtChain { first_op: 0, n_ops: 0 }); // chain 0: the world
        b.chain_ids.push((Vec::new(), 0));
        b
    }

    // ---- interning --------------------------------------------------------------------------------------
    /// The id of the primitive at `address`: a new one on the first visit.
    pub fn prim_id(&mut self, address: usize) -> u32 {
        if let Some(&id) = self.prim_ids.get(&address) {
            return id;
        }
        let id = self.n_prim_ids;
        self.n_prim_ids += 1;
        self.prim_ids.insert(address, id);
        id
    }
    pub fn num_prim_ids(&self) -> u32 {
        self.n_prim_ids
    }
    /// Row of the material at `address`, made by `make` (which may intern textures) on the first visit.
    pub fn material(&mut self, address: usize, make: impl FnOnce(&mut SceneBuilder) -> RtMaterial) -> u32 {
        if let Some(&id) = self.material_ids.get(&address) {
            return id;
        }
        let row = make(self);
        let id = self.materials.len() as u32;
        self.materials.push(row);
        self.material_ids.insert(address, id);
        id
    }
    /// The same for textures; a checker's children are interned by `make` before the checker itself is pushed, so
    /// that children precede their parent (which `rt_scene_commit` requires).
    pub fn texture(&mut self, address: usize, make: impl FnOnce(&mut SceneBuilder) -> RtTexture) -> u32 {
        if let Some(&id) = self.texture_ids.get(&address) {
            return id;
        }
        let row = make(self);
        let id = self.textures.len() as u32;
        self.textures.push(row);
        self.texture_ids.insert(address, id);
        id
    }
    pub fn add_image(&mut self, rgb8: Vec<u8>, width: u32, height: u32) -> u32 {
        self.images.push((rgb8, width, height));
        self.images.len() as u32 - 1
    }
    pub fn add_perlin(&mut self, vectors_256x3: Vec<f64>, perm_3x256: Vec<u32>) -> u32 {
        self.perlins.push((vectors_256x3, perm_3x256));
        self.perlins.len() as u32 - 1
    }

    // ---- instance chains ----------------------------------------------------------------------------------
    fn chain(&mut self) -> u16 {
        if let Some((_, id)) = self.chain_ids.iter().find(|(ops, _)| *ops == self.chain_stack) {
            return *id as u16;
        }
        let id = self.chains.len() as u32;
        assert!(id < 65535, "too many distinct instance chains");
        self.chains.push(RtChain { first_op: self.ops.len() as u32, n_ops: self.chain_stack.len() as u32 });
        self.ops.extend_from_slice(&self.chain_stack);
        self.chain_ids.push((self.chain_stack.clone(), id));
        id as u16
    }
    /// `Translate::hit` / `RotateY::hit` around whatever `inner` flattens (hittable/mod.rs:78-91,118-135).
    pub fn with_translate(&mut self, offset: [f64; 3], inner: impl FnOnce(&mut SceneBuilder)) {
        self.chain_stack.push(RtXformOp { kind: RT_XF_TRANSLATE, reserved: 0, v: offset });
        inner(self);
        self.chain_stack.pop();
    }
    pub fn with_rotate_y(&mut self, sin_theta: f64, cos_theta: f64, inner: impl FnOnce(&mut SceneBuilder)) {
        self.chain_stack.push(RtXformOp { kind: RT_XF_ROTATE_Y, reserved: 0, v: [sin_theta, cos_theta, 0.0] });
        inner(self);
        self.chain_stack.pop();
    }

    // ---- tree entries -------------------------------------------------------------------------------------
    fn emit_run(&mut self, kind: u8, index: u32) {
        let chain = self.chain();
        if self.nodes.len() > self.merge_barrier {
            let level = self.level as u8;
            let last = self.nodes.last_mut().unwrap();
            if last.kind == kind && last.level == level && last.chain == chain && last.first + last.count == index {
                last.count += 1;
                return;
            }
        }
        let at = self.nodes.len() as u32;
        self.nodes.push(RtNode { kind, level: self.level as u8, chain, skip: at + 1, first: index, count: 1 });
    }
    pub fn sphere(&mut self, address: usize, center: [f64; 3], radius: f64, material: u32) {
        let k = self.sphere_mat.len() as u32;
        self.spheres.extend_from_slice(&[center[0], center[1], center[2], radius]);
        self.sphere_mat.push(material);
        let id = self.prim_id(address);
        self.sphere_id.push(id);
        self.emit_run(RT_NODE_SPHERES, k);
    }
    /// `rec16` = normal, d, q, u, v, w exactly as the constructor derived them (parallelogram.rs:40-58).
    pub fn quad(&mut self, address: usize, rec16: [f64; 16], material: u32) {
        let k = self.quad_mat.len() as u32;
        self.quads.extend_from_slice(&rec16);
        self.quad_mat.push(material);
        let id = self.prim_id(address);
        self.quad_id.push(id);
        self.emit_run(RT_NODE_QUADS, k);
    }
    pub fn triangle(&mut self, address: usize, rec16: [f64; 16], material: u32) {
        let k = self.tri_mat.len() as u32;
        self.triangles.extend_from_slice(&rec16);
        self.tri_mat.push(material);
        let id = self.prim_id(address);
        self.tri_id.push(id);
        self.emit_run(RT_NODE_TRIANGLES, k);
    }
    /// `rec10` = p1, p2, p3, j (quadric.rs:14-21).
    pub fn quadric(&mut self, address: usize, rec10: [f64; 10], material: u32) {
        let k = self.quadric_mat.len() as u32;
        self.quadrics.extend_from_slice(&rec10);
        self.quadric_mat.push(material);
        let id = self.prim_id(address);
        self.quadric_id.push(id);
        self.emit_run(RT_NODE_QUADRICS, k);
    }
    fn open(&mut self, kind: u8, first: u32) -> usize {
        assert!(self.level + 2 <= RT_MAX_LEVELS, "BVH/medium nesting deeper than RT_MAX_LEVELS");
        let chain = self.chain();
        self.nodes.push(RtNode { kind, level: self.level as u8, chain, skip: 0, first, count: 0 });
        self.level += 1;
        self.nodes.len() - 1
    }
    fn close(&mut self, at: usize) {
        self.level -= 1;
        self.nodes[at].skip = self.nodes.len() as u32;
    }
    /// One `BVHNode` (bvh.rs:13-26): `bounding_box` is the node's own box, kept for its narrowing; `cull_box`, if
    /// any, bounds the geometry below (see `FlattenHittable::extent`).  `children` flattens `Leaf.objects` in list
    /// order or `left` then `right` (bvh.rs:50-57).
    pub fn bvh_node(&mut self, bounding_box: [f64; 6], cull_box: Option<[f64; 6]>, children: impl FnOnce(&mut SceneBuilder)) {
        let index = (self.boxes.len() / 6) as u32;
        self.boxes.extend_from_slice(&bounding_box);
        let inf = f64::INFINITY;
        self.cull_boxes.extend_from_slice(&cull_box.unwrap_or([-inf, inf, -inf, inf, -inf, inf]));
        let at = self.open(RT_NODE_BVH, index);
        children(self);
        self.close(at);
    }
    /// One `ConstantMedium` (constant_medium.rs:14-43); `boundary` flattens its boundary object.
    pub fn constant_medium(&mut self, address: usize, neg_inv_density: f64, phase_function: u32, boundary: impl FnOnce(&mut SceneBuilder)) {
        assert!(!self.in_medium, "a ConstantMedium inside a medium's boundary is not supported");
        let prim_id = self.prim_id(address);
        let k = self.media.len() as u32;
        self.media.push(RtMedium { neg_inv_density, material: phase_function, prim_id });
        let at = self.open(RT_NODE_MEDIUM, k);
        self.in_medium = true;
        boundary(self);
        self.in_medium = false;
        self.close(at);
    }
    /// A `HittableSlice` (hittable_list.rs:72-106): `children` flattens its objects in list order.  A slice nested in
    /// something else with three or more children and a bounded extent gets a CULL entry in front (no counterpart in
    /// the reference: it only lets rays that cannot touch the geometry pass the list over).
    pub fn slice(&mut self, n_children: usize, cull_box: Option<[f64; 6]>, children: impl FnOnce(&mut SceneBuilder)) {
        let bracket = self.slice_depth > 0 && n_children >= 3 && cull_box.is_some();
        let mut at = 0usize;
        if bracket {
            let index = (self.boxes.len() / 6) as u32;
            let cb = cull_box.unwrap();
            self.boxes.extend_from_slice(&cb);
            self.cull_boxes.extend_from_slice(&cb);
            let chain = self.chain();
            self.nodes.push(RtNode { kind: RT_NODE_CULL, level: self.level as u8, chain, skip: 0, first: index, count: 0 });
            at = self.nodes.len() - 1;
        }
        self.slice_depth += 1;
        children(self);
        self.slice_depth -= 1;
        if bracket {
            self.nodes[at].skip = self.nodes.len() as u32;
            self.merge_barrier = self.nodes.len();
        }
    }
    /// The children of a BVH leaf's slice: in list order, without a CULL entry (the leaf's own cull box bounds them).
    pub fn leaf_items(&mut self, children: impl FnOnce(&mut SceneBuilder)) {
        self.slice_depth += 1;
        children(self);
        self.slice_depth -= 1;
    }

    // ---- hand-over ----------------------------------------------------------------------------------------
    /// `rt_scene_set_*` for everything collected, then `rt_scene_commit` on HIP device `device`.
    pub fn commit(&self, device: i32) -> Result<GpuScene, GpuError> {
        let ptr = unsafe { rt_scene_create() };
        if ptr.is_null() {
            return Err(GpuError { code: RT_ERR_DEVICE, message: "rt_scene_create failed".into() });
        }
        let scene = GpuScene { ptr };
        unsafe {
            check(rt_scene_set_materials(ptr, self.materials.as_ptr(), self.materials.len() as u32))?;
            check(rt_scene_set_textures(ptr, self.textures.as_ptr(), self.textures.len() as u32))?;
            for (slot, (rgb, w, h)) in self.images.iter().enumerate() {
                check(rt_scene_set_image(ptr, slot as u32, rgb.as_ptr(), *w, *h))?;
            }
            for (slot, (vec, perm)) in self.perlins.iter().enumerate() {
                check(rt_scene_set_perlin(ptr, slot as u32, vec.as_ptr(), perm.as_ptr()))?;
            }
            check(rt_scene_set_spheres(ptr, self.spheres.as_ptr(), self.sphere_mat.as_ptr(), self.sphere_id.as_ptr(), self.sphere_mat.len() as u32))?;
            check(rt_scene_set_quads(ptr, self.quads.as_ptr(), self.quad_mat.as_ptr(), self.quad_id.as_ptr(), self.quad_mat.len() as u32))?;
            check(rt_scene_set_triangles(ptr, self.triangles.as_ptr(), self.tri_mat.as_ptr(), self.tri_id.as_ptr(), self.tri_mat.len() as u32))?;
            check(rt_scene_set_quadrics(ptr, self.quadrics.as_ptr(), self.quadric_mat.as_ptr(), self.quadric_id.as_ptr(), self.quadric_mat.len() as u32))?;
            check(rt_scene_set_media(ptr, self.media.as_ptr(), self.media.len() as u32))?;
            check(rt_scene_set_xforms(ptr, self.ops.as_ptr(), self.ops.len() as u32, self.chains.as_ptr(), self.chains.len() as u32))?;
            check(rt_scene_set_nodes(ptr, self.nodes.as_ptr(), self.nodes.len() as u32, self.boxes.as_ptr(), (self.boxes.len() / 6) as u32))?;
            check(rt_scene_set_cull_boxes(ptr, self.cull_boxes.as_ptr(), (self.cull_boxes.len() / 6) as u32))?;
            check(rt_scene_commit(ptr, device))?;
        }
        Ok(scene)
    }
}

/// A scene committed to one GPU.  Not re-entrant; distinct scenes may be used from distinct threads.
pub struct GpuScene {
    ptr: *mut RtScene,
}
unsafe impl Send for GpuScene {}
impl Drop for GpuScene {
    fn drop(&mut self) {
        unsafe { rt_scene_destroy(self.ptr) }
    }
}
impl GpuScene {
    pub fn as_ptr(&self) -> *const RtScene {
        self.ptr
    }
    /// `RT_FLAG_FAST_BVH` for every later call: nearest-first traversal, which is not the reference's result where two
    /// primitives are hit at exactly the same parameter.
    pub fn set_fast_bvh(&self, on: bool) -> Result<(), GpuError> {
        check(unsafe { rt_scene_set_fast_bvh(self.ptr, on as c_int) })
    }
    /// `Camera::render`: the film of radiance SUMS over samples `[spp_begin, spp_begin + spp_count)`, 3 f64 per pixel.
    pub fn render(&self, cam: &RtCamera, opts: &RtRenderOpts) -> Result<(Vec<f64>, RtStats), GpuError> {
        let mut film = vec![0f64; 3 * cam.width as usize * cam.height as usize];
        let mut stats = RtStats::default();
        check(unsafe { rt_render(self.ptr, cam, opts, film.as_mut_ptr(), &mut stats) })?;
        Ok((film, stats))
    }
    /// `pixel_color / spp` then `proccess_color` (camera.rs:265-267, point3/color.rs:49-68).
    pub fn tonemap(&self, film: &[f64], width: u32, height: u32, spp: u32) -> Result<Vec<u8>, GpuError> {
        assert_eq!(film.len(), 3 * width as usize * height as usize);
        let mut rgb = vec![0u8; film.len()];
        check(unsafe { rt_tonemap_rgb8(self.ptr, film.as_ptr(), width, height, spp, rgb.as_mut_ptr()) })?;
        Ok(rgb)
    }
}
