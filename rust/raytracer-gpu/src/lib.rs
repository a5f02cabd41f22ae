//! GPU backend for villarjorge/raytracer: the C ABI of `librt_amd.so` (`ffi`), a safe collector of the flat scene
//! arrays it takes (`builder`), and the traits through which the reference's object tree writes itself into that
//! collector (`flatten`).  The implementations of those traits for the reference's own types need its private
//! fields and therefore live in the reference crate: `reference/gpu_impls.rs` is that file.
pub mod builder;
pub mod ffi;
pub mod flatten;

pub use builder::{GpuError, GpuScene, SceneBuilder};
pub use flatten::{FlattenHittable, FlattenMaterial, FlattenTexture};
