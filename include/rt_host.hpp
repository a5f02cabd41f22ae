// rt_host.hpp -- C++ host layer: the reference's scene-building API (Hittable / Material / Texture / Camera,
// .obj loading) restated in C++ with the reference's names, plus the one step the reference lacks: flattening the
// object tree into the arrays librt_amd.so takes (include/rt_amd.h).  Nothing here intersects a ray; rendering is
// Camera::render -> flatten -> rt_scene_* -> rt_render on the GPU.
//
// The reference is a Rust crate; each class cites the file:line whose constructor arithmetic it reproduces,
// because bounding boxes, BVH topology and the derived plane constants decide which hits the renderer accepts.
#pragma once

#include <cstdint>
#include <cstdio>
#include <memory>
#include <string>
#include <vector>

#include "rt_amd.h"

namespace rtamd {

// ---- src/point3/mod.rs ------------------------------------------------------------------------------------
struct Point3 {
    double x = 0.0, y = 0.0, z = 0.0;
    Point3() {}
    Point3(double x_, double y_, double z_) : x(x_), y(y_), z(z_) {}
};
typedef Point3 Vector3;
typedef Point3 Color;

// ---- src/aabb.rs --------------------------------------------------------------------------------------------
struct Range {
    double start, end;
};
struct AABB {
    Range x, y, z;
    AABB();  // AABB::default(), aabb.rs:29-33
    static AABB from_points(Point3 a, Point3 b);
    const Range& axis_interval(int n) const { return n == 0 ? x : (n == 1 ? y : z); }
    int longest_axis() const;
    AABB shifted(Point3 by) const;  // impl Add<Point3> for AABB, aabb.rs:53-68
};
AABB join_aabbs(const AABB& a, const AABB& b);

// ---- src/texture.rs, src/perlin.rs ------------------------------------------------------------------------
class Texture {
public:
    virtual ~Texture() {}
    virtual rt_texture_kind kind() const = 0;
};
typedef std::shared_ptr<const Texture> TexturePtr;

class SolidColor : public Texture {
public:
    Color albedo;
    static std::shared_ptr<SolidColor> create(Color c);  // SolidColor::new
    rt_texture_kind kind() const override { return RT_TEX_SOLID; }
};
class CheckerTexture : public Texture {
public:
    double inverse_scale;
    TexturePtr even, odd;
    static std::shared_ptr<CheckerTexture> from_pointers(double scale, TexturePtr even, TexturePtr odd);
    static std::shared_ptr<CheckerTexture> from_colors(double scale, Color even, Color odd);
    rt_texture_kind kind() const override { return RT_TEX_CHECKER; }
};
class ImageTexture : public Texture {
public:
    std::vector<uint8_t> rgb;  // row-major RGB8, top row first
    uint32_t width = 0, height = 0;
    // Reads a binary PPM (P6, maxval 255) or a PNG.  On failure returns the reference's magenta checker
    // (ImageTexture::new_or_fallback, texture.rs:89-113).  Binary PPM and PNG are decoded; JPEG is not.
    static TexturePtr new_or_fallback(const std::string& path);
    static std::shared_ptr<ImageTexture> from_rgb8(const uint8_t* rgb, uint32_t w, uint32_t h);
    rt_texture_kind kind() const override { return RT_TEX_IMAGE; }
};
struct PerlinNoise {  // perlin.rs:11-16
    double random_vectors[256][3];
    uint32_t x_perm[256], y_perm[256], z_perm[256];
};
// create_perlin_noise (perlin.rs:18-31): the caller supplies the 768 uniform [-1,1) draws in reference order;
// the permutations are the identity because perlin.rs:37 iterates an empty range.
PerlinNoise create_perlin_noise(const double* draws768);
class PerlinNoiseTexture : public Texture {
public:
    PerlinNoise perlin_noise;
    double scale;
    static std::shared_ptr<PerlinNoiseTexture> create(const PerlinNoise& n, double scale);
    rt_texture_kind kind() const override { return RT_TEX_PERLIN; }
};

// ---- src/material.rs --------------------------------------------------------------------------------------
class Material {
public:
    virtual ~Material() {}
    virtual rt_material_kind kind() const = 0;
};
typedef std::shared_ptr<const Material> MaterialPtr;
class Lambertian : public Material {
public:
    TexturePtr texture;
    static std::shared_ptr<Lambertian> from_color(Color c);
    static std::shared_ptr<Lambertian> from_texture(TexturePtr t);
    rt_material_kind kind() const override { return RT_MAT_LAMBERTIAN; }
};
class Metal : public Material {
public:
    Color albedo;
    double fuzz;
    rt_material_kind kind() const override { return RT_MAT_METAL; }
};
std::shared_ptr<Metal> metal(Color albedo, double fuzz);
class Dielectric : public Material {
public:
    double refraction_index;
    rt_material_kind kind() const override { return RT_MAT_DIELECTRIC; }
};
std::shared_ptr<Dielectric> dielectric(double refraction_index);
class DiffuseLight : public Material {
public:
    TexturePtr texture;
    static std::shared_ptr<DiffuseLight> from_color(Color c);
    static std::shared_ptr<DiffuseLight> from_texture(TexturePtr t);
    rt_material_kind kind() const override { return RT_MAT_DIFFUSE_LIGHT; }
};
class Isotropic : public Material {
public:
    TexturePtr texture;
    rt_material_kind kind() const override { return RT_MAT_ISOTROPIC; }
};

// ---- flattened scene ---------------------------------------------------------------------------------------
struct FlatScene {
    std::vector<rt_material> materials;
    std::vector<rt_texture> textures;
    struct Image { std::vector<uint8_t> rgb; uint32_t w, h; };
    std::vector<Image> images;
    struct Perlin { std::vector<double> vec; std::vector<uint32_t> perm; };
    std::vector<Perlin> perlins;
    std::vector<double> spheres, quads, triangles, quadrics, boxes;
    std::vector<double> cull_boxes;  // one per entry of `boxes`: conservative bounds of the subtree's hit points
    std::vector<uint32_t> sphere_mat, sphere_id, quad_mat, quad_id, tri_mat, tri_id, quadric_mat, quadric_id;
    std::vector<rt_medium> media;
    std::vector<rt_xform_op> ops;
    std::vector<rt_chain> chains;
    std::vector<rt_node> nodes;
    uint32_t n_prim_ids = 0;
    // Creates an rt_scene, hands every array over and commits it to `device`.  Returns nullptr on failure
    // (rt_last_error() has the reason).
    rt_scene* commit(int device) const;
};
class Flattener;  // internal

// ---- src/hittable/* ----------------------------------------------------------------------------------------
class Hittable {
public:
    virtual ~Hittable() {}
    const AABB& bounding_box() const { return bbox_; }
    virtual void flatten(Flattener& f) const = 0;
    // Geometric bounds, in the frame the object's parent sees, of every point a hit on this object can be reported
    // at: grows [lo, hi] (which start out empty).  Unlike bounding_box() this is not part of the reference; it only
    // feeds the cull boxes, which skip work without changing results.  Unbounded objects set +-infinity.
    virtual void extent(double lo[3], double hi[3]) const = 0;
protected:
    AABB bbox_;
};
typedef std::shared_ptr<const Hittable> HittablePtr;

class Sphere : public Hittable {  // sphere.rs:12-37
public:
    Point3 center;
    double radius;
    MaterialPtr material;
    Sphere(Point3 center, double radius, MaterialPtr material);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class Parallelogram : public Hittable {  // parallelogram.rs:12-58
public:
    double d;
    Vector3 normal, q, u, v, w;
    MaterialPtr material;
    Parallelogram(Point3 q, Vector3 u, Vector3 v, MaterialPtr material);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class Triangle : public Hittable {  // triangle.rs:11-87
public:
    double d;
    Vector3 normal, q, u, v, w;
    MaterialPtr material;
    Triangle(Point3 q, Vector3 u, Vector3 v, MaterialPtr material);
    static std::shared_ptr<Triangle> from_vertex_locations(Point3 v1, Point3 v2, Point3 v3, MaterialPtr material);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class Quadric : public Hittable {  // quadric.rs:14-21
public:
    Point3 p1, p2, p3;
    double j;
    MaterialPtr material;
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
    friend std::shared_ptr<Quadric> make_quadric(Point3, Point3, Point3, double, MaterialPtr, AABB);
};
std::shared_ptr<Quadric> y_cylinder(Point3 center, double radius, MaterialPtr material);      // quadric.rs:110-135
std::shared_ptr<Quadric> quadric_sphere(Point3 center, double radius, MaterialPtr material);  // quadric.rs:138-158
std::shared_ptr<Quadric> y_cone(Point3 center, Point3 offset, MaterialPtr material);          // quadric.rs:161-177

class HittableSlice;
class BVHNode;
class HittableList {  // hittable_list.rs:13-47
public:
    std::vector<HittablePtr> objects;
    AABB bounding_box;
    template <class T>
    void add(std::shared_ptr<T> o) { add_pointer(o); }
    void add_pointer(HittablePtr o);
    std::shared_ptr<HittableSlice> to_hittable_slice() const;
    std::shared_ptr<BVHNode> to_bvh_node() const;
};
HittableList create_box(Point3 a, Point3 b, MaterialPtr material);  // parallelogram.rs:146-195

class HittableSlice : public Hittable {  // hittable_list.rs:72-106
public:
    std::vector<HittablePtr> objects;
    static std::shared_ptr<HittableSlice> from_hittable_list(const HittableList& l);
    void flatten_items(Flattener& f) const;  // the children in list order, nothing else
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class BVHNode : public Hittable {  // bvh.rs:13-139
public:
    bool leaf = false;
    std::shared_ptr<HittableSlice> objects;  // Leaf
    HittablePtr left, right;                 // Internal
    explicit BVHNode(std::vector<HittablePtr> objects);  // BVHNode::new
    static std::shared_ptr<BVHNode> from_hittable_list(const HittableList& l);
    static std::shared_ptr<BVHNode> from_vec(std::vector<HittablePtr> objects);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class Translate : public Hittable {  // hittable/mod.rs:71-107
public:
    HittablePtr object;
    Vector3 offset;
    Translate(HittablePtr object, Vector3 offset);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class RotateY : public Hittable {  // hittable/mod.rs:110-189
public:
    HittablePtr object;
    double sin_theta, cos_theta;
    RotateY(HittablePtr object, double angle_in_degrees);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};
class ConstantMedium : public Hittable {  // constant_medium.rs:14-43
public:
    HittablePtr boundary;
    double neg_inv_density;
    MaterialPtr phase_function;
    ConstantMedium(HittablePtr boundary, double density, TexturePtr texture);
    static std::shared_ptr<ConstantMedium> from_color(HittablePtr boundary, double density, Color color);
    void flatten(Flattener& f) const override;
    void extent(double lo[3], double hi[3]) const override;
};

// load_model (load_obj.rs:15-115).  Throws std::runtime_error where the reference panics.
std::shared_ptr<BVHNode> load_model(const std::string& model_path, double scale, MaterialPtr material);

// Depth-first flatten of the tree under `world`.  Throws std::runtime_error on scenes the device format cannot
// express (a medium inside a medium's boundary, nesting deeper than RT_MAX_LEVELS).
FlatScene flatten(const Hittable& world);

// ---- src/camera.rs -----------------------------------------------------------------------------------------
struct CameraPosition { Point3 look_from, look_at; Vector3 view_up; };
struct ThinLens { double defocus_angle, focus_distance; };
struct ImageQuality {
    uint32_t samples_per_pixel, max_depth;
    static ImageQuality low() { return ImageQuality{20, 4}; }        // camera.rs:64-69
    static ImageQuality medium() { return ImageQuality{200, 50}; }   // camera.rs:71-76
};
class Camera {
public:
    rt_camera cam;  // the derived fields of camera.rs:23-37
    uint32_t samples_per_pixel, max_depth;
    Color background_color;
    uint64_t seed = 42;  // key of the counter-based generator (the reference is OS-seeded)
    int device = 0;
    std::string output_path = "images/image.png";  // camera.rs:238,270
    FILE* metrics = nullptr;  // where render() prints one JSON line of what it traced (the reference prints banners and
                              // a progress bar, camera.rs:252,256,269); null: nothing
    static Camera create(double aspect_ratio, uint32_t image_width, ImageQuality q, double vfov, ThinLens lens,
                         CameraPosition pos, Color background);  // Camera::new, camera.rs:80-143
    // Camera::render / Camera::thrender (camera.rs:217-271) on the GPU.  Returns 0, or an RT_ERR_* code.
    int render(const Hittable& world, rt_stats* stats = nullptr) const;
    int thrender(const Hittable& world, rt_stats* stats = nullptr) const { return render(world, stats); }
};

// Reads a binary PPM (P6) or a PNG into 8-bit RGB, top row first.  False when the file is neither (JPEG: convert it
// beforehand; decoders differ in the last bit anyway).
bool read_image_rgb8(const std::string& path, std::vector<uint8_t>& rgb, uint32_t& w, uint32_t& h);
// Writes an 8-bit RGB PNG (zlib deflate).  Returns false on I/O failure.
bool write_png(const std::string& path, const uint8_t* rgb, uint32_t w, uint32_t h);

}  // namespace rtamd
