//! The one hook the reference's traits gain.  `Hittable`, `Material` and `Texture` objects sit behind
//! `Arc<dyn ...>` with private fields (`hittable/mod.rs:65-68`, `material.rs:18-31`, `texture.rs:11-13`), so each
//! trait gets these as supertraits (or their methods are added to the trait itself):
//!
//! ```ignore
//! pub trait Hittable: Send + Sync + raytracer_gpu::FlattenHittable { ... }
//! pub trait Material: Sync + Send + raytracer_gpu::FlattenMaterial { ... }
//! pub trait Texture: Sync + Send + raytracer_gpu::FlattenTexture { ... }
//! ```
//!
//! `reference/gpu_impls.rs` implements them for every type of the reference.
use crate::builder::SceneBuilder;
use crate::ffi::{RtMaterial, RtTexture};

pub trait FlattenHittable {
    /// Appends this object (and everything below it) to `b`, in the order `hit()` visits it.
    fn flatten(&self, b: &mut SceneBuilder);
    /// Geometric bounds (x0 x1 y0 y1 z0 z1, in the frame the object's parent sees) of every point a hit on this
    /// object can be reported at, or `None` when unbounded (quadrics) or not to be skipped on an interval test
    /// (`ConstantMedium`, which decides from the whole line whether it draws a random number,
    /// constant_medium.rs:55-63).  Not part of the reference: it only feeds the cull boxes, which skip work without
    /// changing results.  The reference's own boxes all reach the origin (aabb.rs:29-33) and are poor at that.
    fn extent(&self) -> Option<[f64; 6]>;
}
pub trait FlattenMaterial {
    /// The material's table row; textures it holds are interned through `b`.
    fn flatten(&self, b: &mut SceneBuilder) -> RtMaterial;
}
pub trait FlattenTexture {
    fn flatten(&self, b: &mut SceneBuilder) -> RtTexture;
}

/// Union of two extents.
pub fn join(a: Option<[f64; 6]>, b: Option<[f64; 6]>) -> Option<[f64; 6]> {
    match (a, b) {
        (Some(a), Some(b)) => Some([a[0].min(b[0]), a[1].max(b[1]), a[2].min(b[2]), a[3].max(b[3]), a[4].min(b[4]), a[5].max(b[5])]),
        _ => None,
    }
}
/// An extent widened by 1e-5 (1 + largest |coordinate|): a margin that dwarfs the rounding of the intersection
/// routines and of the box test itself.
pub fn cull_box(extent: Option<[f64; 6]>) -> Option<[f64; 6]> {
    let e = extent?;
    if e.iter().any(|v| !v.is_finite()) || e[0] > e[1] || e[2] > e[3] || e[4] > e[5] {
        return None;
    }
    let big = e.iter().fold(0.0f64, |m, v| m.max(v.abs()));
    let margin = 1e-5 * (1.0 + big);
    Some([e[0] - margin, e[1] + margin, e[2] - margin, e[3] + margin, e[4] - margin, e[5] + margin])
}
/// The extent of an object seen through `Translate` (hittable/mod.rs:88).
pub fn translated(e: Option<[f64; 6]>, offset: [f64; 3]) -> Option<[f64; 6]> {
    let e = e?;
    Some([e[0] + offset[0], e[1] + offset[0], e[2] + offset[1], e[3] + offset[1], e[4] + offset[2], e[5] + offset[2]])
}
/// The extent of an object seen through `RotateY`: object space -> parent space is the rotation hit points get
/// (hittable/mod.rs:131-132), applied to all eight corners.
pub fn rotated_y(e: Option<[f64; 6]>, sin_theta: f64, cos_theta: f64) -> Option<[f64; 6]> {
    let e = e?;
    let (mut lo, mut hi) = ([f64::INFINITY; 2], [f64::NEG_INFINITY; 2]);
    for &x in &[e[0], e[1]] {
        for &z in &[e[4], e[5]] {
            let p = [cos_theta * x + sin_theta * z, -sin_theta * x + cos_theta * z];
            for c in 0..2 {
                lo[c] = lo[c].min(p[c]);
                hi[c] = hi[c].max(p[c]);
            }
        }
    }
    Some([lo[0], hi[0], e[2], e[3], lo[1], hi[1]])
}
