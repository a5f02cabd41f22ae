//! To be added to the reference crate (villarjorge/raytracer) as `src/gpu_impls.rs`, with `mod gpu_impls;` in
//! `main.rs`, `raytracer-gpu = { path = "..." }` in `Cargo.toml` and the supertraits named in
//! `raytracer_gpu::flatten` on `Hittable`, `Material` and `Texture`.  It reads private fields of the reference's
//! types, which therefore become `pub(crate)`: `Sphere {center, radius, material}`, `Parallelogram` and `Triangle`
//! `{normal, d, q, u, v, w, material}`, `Quadric {p1, p2, p3, j, material}`, `Translate {object, offset}`,
//! `RotateY {object, sin_theta, cos_theta}`, `HittableSlice {objects}`, `AABB {x, y, z}`, `SolidColor {albedo}`,
//! `ImageTexture {image}`, `DiffuseLight {texture}`, and the `Camera` fields `render_gpu` reads.  Nothing else in
//! the reference changes; a scene function ends with `cam.render_gpu(&world)` where it had `cam.thrender(&world)`.
use std::sync::Arc;

use raytracer_gpu::builder::{address_of, SceneBuilder};
use raytracer_gpu::ffi::*;
use raytracer_gpu::flatten::{cull_box, join, rotated_y, translated};
use raytracer_gpu::{FlattenHittable, FlattenMaterial, FlattenTexture};

use crate::aabb::AABB;
use crate::bvh::BVHNode;
use crate::camera::Camera;
use crate::hittable::constant_medium::ConstantMedium;
use crate::hittable::hittable_list::HittableSlice;
use crate::hittable::parallelogram::Parallelogram;
use crate::hittable::quadric::Quadric;
use crate::hittable::sphere::Sphere;
use crate::hittable::triangle::Triangle;
use crate::hittable::{Hittable, RotateY, Translate};
use crate::material::{Dielectric, DiffuseLight, Isotropic, Lambertian, Material, Metal};
use crate::point3::Point3;
use crate::texture::{CheckerTexture, ImageTexture, PerlinNoiseTexture, SolidColor, Texture};

fn v3(p: &Point3) -> [f64; 3] {
    [p.x, p.y, p.z]
}
fn box6(b: &AABB) -> [f64; 6] {
    [b.x.start, b.x.end, b.y.start, b.y.end, b.z.start, b.z.end]
}
fn corners(points: &[Point3]) -> Option<[f64; 6]> {
    let mut e = [f64::INFINITY, f64::NEG_INFINITY, f64::INFINITY, f64::NEG_INFINITY, f64::INFINITY, f64::NEG_INFINITY];
    for p in points {
        for (c, v) in [p.x, p.y, p.z].iter().enumerate() {
            e[2 * c] = e[2 * c].min(*v);
            e[2 * c + 1] = e[2 * c + 1].max(*v);
        }
    }
    Some(e)
}
fn material_id(b: &mut SceneBuilder, m: &Arc<dyn Material>) -> u32 {
    b.material(address_of(m.as_ref()), |b| m.flatten(b))
}
fn texture_id(b: &mut SceneBuilder, t: &Arc<dyn Texture>) -> u32 {
    b.texture(address_of(t.as_ref()), |b| t.flatten(b))
}

// ---- hittables -------------------------------------------------------------------------------------------------
impl FlattenHittable for Sphere {
    fn flatten(&self, b: &mut SceneBuilder) {
        let m = material_id(b, &self.material);
        b.sphere(address_of(self), v3(&self.center), self.radius, m);
    }
    fn extent(&self) -> Option<[f64; 6]> {
        let (c, r) = (&self.center, self.radius);
        Some([c.x - r, c.x + r, c.y - r, c.y + r, c.z - r, c.z + r])
    }
}
fn planar(n: &Point3, d: f64, q: &Point3, u: &Point3, v: &Point3, w: &Point3) -> [f64; 16] {
    [n.x, n.y, n.z, d, q.x, q.y, q.z, u.x, u.y, u.z, v.x, v.y, v.z, w.x, w.y, w.z]
}
impl FlattenHittable for Parallelogram {
    fn flatten(&self, b: &mut SceneBuilder) {
        let m = material_id(b, &self.material);
        b.quad(address_of(self), planar(&self.normal, self.d, &self.q, &self.u, &self.v, &self.w), m);
    }
    fn extent(&self) -> Option<[f64; 6]> {
        corners(&[self.q, self.q + self.u, self.q + self.v, self.q + self.u + self.v])
    }
}
impl FlattenHittable for Triangle {
    fn flatten(&self, b: &mut SceneBuilder) {
        let m = material_id(b, &self.material);
        b.triangle(address_of(self), planar(&self.normal, self.d, &self.q, &self.u, &self.v, &self.w), m);
    }
    fn extent(&self) -> Option<[f64; 6]> {
        corners(&[self.q, self.q + self.u, self.q + self.v])
    }
}
impl FlattenHittable for Quadric {
    fn flatten(&self, b: &mut SceneBuilder) {
        let m = material_id(b, &self.material);
        let (p1, p2, p3) = (&self.p1, &self.p2, &self.p3);
        b.quadric(address_of(self), [p1.x, p1.y, p1.z, p2.x, p2.y, p2.z, p3.x, p3.y, p3.z, self.j], m);
    }
    fn extent(&self) -> Option<[f64; 6]> {
        None // the surfaces are unbounded (an infinite cylinder, a double cone)
    }
}
impl FlattenHittable for HittableSlice {
    fn flatten(&self, b: &mut SceneBuilder) {
        let cb = cull_box(self.extent());
        b.slice(self.objects.len(), cb, |b| {
            for o in self.objects.iter() {
                o.flatten(b); // list order = visit order (hittable_list.rs:93-99)
            }
        });
    }
    fn extent(&self) -> Option<[f64; 6]> {
        let mut it = self.objects.iter();
        let first = it.next()?.extent();
        it.fold(first, |e, o| join(e, o.extent()))
    }
}
impl FlattenHittable for BVHNode {
    fn flatten(&self, b: &mut SceneBuilder) {
        let cb = cull_box(self.extent());
        match self {
            BVHNode::Leaf { objects, bounding_box } => b.bvh_node(box6(bounding_box), cb, |b| {
                b.leaf_items(|b| {
                    for o in objects.objects.iter() {
                        o.flatten(b);
                    }
                })
            }),
            BVHNode::Internal { left, right, bounding_box } => b.bvh_node(box6(bounding_box), cb, |b| {
                left.flatten(b); // bvh.rs:50-57: left first, then right
                right.flatten(b);
            }),
        }
    }
    fn extent(&self) -> Option<[f64; 6]> {
        match self {
            BVHNode::Leaf { objects, .. } => objects.extent(),
            BVHNode::Internal { left, right, .. } => join(left.extent(), right.extent()),
        }
    }
}
impl FlattenHittable for Translate {
    fn flatten(&self, b: &mut SceneBuilder) {
        b.with_translate(v3(&self.offset), |b| self.object.flatten(b));
    }
    fn extent(&self) -> Option<[f64; 6]> {
        translated(self.object.extent(), v3(&self.offset))
    }
}
impl FlattenHittable for RotateY {
    fn flatten(&self, b: &mut SceneBuilder) {
        b.with_rotate_y(self.sin_theta, self.cos_theta, |b| self.object.flatten(b));
    }
    fn extent(&self) -> Option<[f64; 6]> {
        rotated_y(self.object.extent(), self.sin_theta, self.cos_theta)
    }
}
impl FlattenHittable for ConstantMedium {
    fn flatten(&self, b: &mut SceneBuilder) {
        let phase = material_id(b, &self.phase_function);
        b.constant_medium(address_of(self), self.neg_inv_density, phase, |b| self.boundary.flatten(b));
    }
    fn extent(&self) -> Option<[f64; 6]> {
        None
    }
}

// ---- materials (material.rs) -----------------------------------------------------------------------------------
impl FlattenMaterial for Lambertian {
    fn flatten(&self, b: &mut SceneBuilder) -> RtMaterial {
        RtMaterial { kind: RT_MAT_LAMBERTIAN, texture: texture_id(b, &self.texture), ..Default::default() }
    }
}
impl FlattenMaterial for Metal {
    fn flatten(&self, _b: &mut SceneBuilder) -> RtMaterial {
        RtMaterial { kind: RT_MAT_METAL, texture: 0, albedo: v3(&self.albedo), param: self.fuzz }
    }
}
impl FlattenMaterial for Dielectric {
    fn flatten(&self, _b: &mut SceneBuilder) -> RtMaterial {
        RtMaterial { kind: RT_MAT_DIELECTRIC, param: self.refraction_index, ..Default::default() }
    }
}
impl FlattenMaterial for DiffuseLight {
    fn flatten(&self, b: &mut SceneBuilder) -> RtMaterial {
        RtMaterial { kind: RT_MAT_DIFFUSE_LIGHT, texture: texture_id(b, &self.texture), ..Default::default() }
    }
}
impl FlattenMaterial for Isotropic {
    fn flatten(&self, b: &mut SceneBuilder) -> RtMaterial {
        RtMaterial { kind: RT_MAT_ISOTROPIC, texture: texture_id(b, &self.texture), ..Default::default() }
    }
}

// ---- textures (texture.rs) -------------------------------------------------------------------------------------
impl FlattenTexture for SolidColor {
    fn flatten(&self, _b: &mut SceneBuilder) -> RtTexture {
        RtTexture { kind: RT_TEX_SOLID, color: v3(&self.albedo), ..Default::default() }
    }
}
impl FlattenTexture for CheckerTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> RtTexture {
        let (even, odd) = (texture_id(b, &self.even), texture_id(b, &self.odd)); // children first
        RtTexture { kind: RT_TEX_CHECKER, a: even, b: odd, scale: self.inverse_scale, ..Default::default() }
    }
}
impl FlattenTexture for ImageTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> RtTexture {
        let slot = b.add_image(self.image.as_raw().clone(), self.image.width(), self.image.height());
        RtTexture { kind: RT_TEX_IMAGE, a: slot, ..Default::default() }
    }
}
impl FlattenTexture for PerlinNoiseTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> RtTexture {
        let n = &self.perlin_noise;
        let vectors: Vec<f64> = n.random_vectors.iter().flat_map(|v| [v.x, v.y, v.z]).collect();
        let perm: Vec<u32> = n.x_perm.iter().chain(n.y_perm.iter()).chain(n.z_perm.iter()).copied().collect();
        let slot = b.add_perlin(vectors, perm);
        RtTexture { kind: RT_TEX_PERLIN, a: slot, scale: self.scale, ..Default::default() }
    }
}

// ---- Camera::render_gpu (camera.rs:217-271 on the GPU) ---------------------------------------------------------
impl Camera {
    pub fn render_gpu(&self, world: &dyn Hittable) {
        let mut b = SceneBuilder::new();
        world.flatten(&mut b);
        let scene = b.commit(0).expect("scene commit");
        let cam = RtCamera {
            width: self.image_width,
            height: self.image_height,
            pixel00: v3(&self.pixel00_loc),
            delta_u: v3(&self.pixel_delta_u),
            delta_v: v3(&self.pixel_delta_v),
            center: v3(&self.camera_center),
            defocus_angle: self.defocus_angle,
            disk_u: v3(&self.defocus_disk_u),
            disk_v: v3(&self.defocus_disk_v),
        };
        let opts = RtRenderOpts {
            spp_begin: 0,
            spp_count: self.samples_per_pixel,
            max_depth: self.max_depth,
            flags: 0,
            seed: 42, // the reference's ThreadRng is OS-seeded; the backend's generator is counter-based and keyed
            background: v3(&self.background_color),
            batch_paths: 0,
        };
        let (film, stats) = scene.render(&cam, &opts).expect("render");
        println!("Traced {} paths, {} rays in {:.1} ms on the GPU", stats.paths, stats.rays, stats.ms_total);
        let rgb = scene.tonemap(&film, cam.width, cam.height, opts.spp_count).expect("tonemap");
        image::RgbImage::from_raw(cam.width, cam.height, rgb).unwrap().save("images/image.png").unwrap();
    }
}
