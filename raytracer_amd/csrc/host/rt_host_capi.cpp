// rt_host_capi.cpp -- C wrapper over the C++ host layer (include/rt_host.h).
#include "rt_host.h"

#include <cmath>

#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

#include "rt_host.hpp"

using namespace rtamd;

struct rth_builder {
    std::vector<TexturePtr> textures;
    std::vector<MaterialPtr> materials;
    std::vector<HittablePtr> objects;
    std::vector<HittableList> lists;
    FlatScene flat;
    std::vector<uint8_t> texels;      // images back to back, as the device holds them
    std::vector<uint32_t> image_desc; // per image: first pixel, width, height
    std::vector<double> perlin_vec;
    std::vector<uint32_t> perlin_perm;
    bool have_flat = false;
    std::string err;
};

static Point3 p3(const double* p) { return Point3(p[0], p[1], p[2]); }
static int bad(rth_builder* b, const char* what) {
    b->err = what;
    return -1;
}
#define NEEDB if (!b) return -1
#define TEX(i) if ((i) < 0 || (size_t)(i) >= b->textures.size()) return bad(b, "texture handle out of range")
#define MAT(i) if ((i) < 0 || (size_t)(i) >= b->materials.size()) return bad(b, "material handle out of range")
#define OBJ(i) if ((i) < 0 || (size_t)(i) >= b->objects.size()) return bad(b, "object handle out of range")
#define LST(i) if ((i) < 0 || (size_t)(i) >= b->lists.size()) return bad(b, "list handle out of range")
#define GUARD(body)                          \
    try {                                    \
        body                                 \
    } catch (const std::exception& e) {      \
        b->err = e.what();                   \
        return -1;                           \
    } catch (...) {                          \
        b->err = "unexpected failure";       \
        return -1;                           \
    }
static int push_tex(rth_builder* b, TexturePtr t) { b->textures.push_back(t); return (int)b->textures.size() - 1; }
static int push_mat(rth_builder* b, MaterialPtr m) { b->materials.push_back(m); return (int)b->materials.size() - 1; }
static int push_obj(rth_builder* b, HittablePtr o) { b->objects.push_back(o); return (int)b->objects.size() - 1; }

extern "C" {

rth_builder* rth_new(void) { return new (std::nothrow) rth_builder(); }
void rth_free(rth_builder* b) { delete b; }
const char* rth_error(rth_builder* b) { return b ? b->err.c_str() : "null builder"; }

int rth_tex_solid(rth_builder* b, double r, double g, double bl) { NEEDB; GUARD(return push_tex(b, SolidColor::create(Color(r, g, bl)));) }
int rth_tex_checker(rth_builder* b, double scale, int even, int odd) {
    NEEDB; TEX(even); TEX(odd);
    GUARD(return push_tex(b, CheckerTexture::from_pointers(scale, b->textures[even], b->textures[odd]));)
}
int rth_tex_image(rth_builder* b, const uint8_t* rgb8, uint32_t w, uint32_t h) {
    NEEDB;
    if (!rgb8 || !w || !h) return bad(b, "empty image");
    GUARD(return push_tex(b, ImageTexture::from_rgb8(rgb8, w, h));)
}
int rth_tex_image_file(rth_builder* b, const char* path) {
    NEEDB;
    if (!path) return bad(b, "null path");
    GUARD(return push_tex(b, ImageTexture::new_or_fallback(path));)
}
int rth_tex_perlin(rth_builder* b, double scale, const double* draws768) {
    NEEDB;
    if (!draws768) return bad(b, "null perlin draws");
    GUARD(return push_tex(b, PerlinNoiseTexture::create(create_perlin_noise(draws768), scale));)
}
int rth_mat_lambertian(rth_builder* b, int tex) { NEEDB; TEX(tex); GUARD(return push_mat(b, Lambertian::from_texture(b->textures[tex]));) }
int rth_mat_metal(rth_builder* b, double r, double g, double bl, double fuzz) { NEEDB; GUARD(return push_mat(b, metal(Color(r, g, bl), fuzz));) }
int rth_mat_dielectric(rth_builder* b, double ri) { NEEDB; GUARD(return push_mat(b, dielectric(ri));) }
int rth_mat_diffuse_light(rth_builder* b, int tex) { NEEDB; TEX(tex); GUARD(return push_mat(b, DiffuseLight::from_texture(b->textures[tex]));) }
int rth_mat_isotropic(rth_builder* b, int tex) {
    NEEDB; TEX(tex);
    GUARD(auto m = std::make_shared<Isotropic>();
          m->texture = b->textures[tex];
          return push_mat(b, m);)
}
int rth_sphere(rth_builder* b, const double* c, double r, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, std::make_shared<Sphere>(p3(c), r, b->materials[mat]));)
}
int rth_quad(rth_builder* b, const double* q, const double* u, const double* v, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, std::make_shared<Parallelogram>(p3(q), p3(u), p3(v), b->materials[mat]));)
}
int rth_triangle(rth_builder* b, const double* q, const double* u, const double* v, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, std::make_shared<Triangle>(p3(q), p3(u), p3(v), b->materials[mat]));)
}
int rth_y_cylinder(rth_builder* b, const double* c, double r, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, y_cylinder(p3(c), r, b->materials[mat]));)
}
int rth_quadric_sphere(rth_builder* b, const double* c, double r, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, quadric_sphere(p3(c), r, b->materials[mat]));)
}
int rth_y_cone(rth_builder* b, const double* c, const double* off, int mat) {
    NEEDB; MAT(mat);
    GUARD(return push_obj(b, y_cone(p3(c), p3(off), b->materials[mat]));)
}
int rth_list_new(rth_builder* b) { NEEDB; GUARD(b->lists.emplace_back(); return (int)b->lists.size() - 1;) }
int rth_list_add(rth_builder* b, int list, int obj) {
    NEEDB; LST(list); OBJ(obj);
    GUARD(b->lists[list].add_pointer(b->objects[obj]); return 0;)
}
int rth_box(rth_builder* b, const double* a, const double* c, int mat) {
    NEEDB; MAT(mat);
    GUARD(b->lists.push_back(create_box(p3(a), p3(c), b->materials[mat])); return (int)b->lists.size() - 1;)
}
int rth_slice(rth_builder* b, int list) { NEEDB; LST(list); GUARD(return push_obj(b, b->lists[list].to_hittable_slice());) }
int rth_bvh(rth_builder* b, int list) {
    NEEDB; LST(list);
    GUARD(return push_obj(b, b->lists[list].to_bvh_node());)
}
int rth_translate(rth_builder* b, int obj, const double* off) {
    NEEDB; OBJ(obj);
    if (!off) return bad(b, "null offset");
    GUARD(return push_obj(b, std::make_shared<Translate>(b->objects[obj], p3(off)));)
}
int rth_rotate_y(rth_builder* b, int obj, double deg) {
    NEEDB; OBJ(obj);
    GUARD(return push_obj(b, std::make_shared<RotateY>(b->objects[obj], deg));)
}
int rth_constant_medium(rth_builder* b, int boundary, double density, int tex) {
    NEEDB; OBJ(boundary); TEX(tex);
    GUARD(return push_obj(b, std::make_shared<ConstantMedium>(b->objects[boundary], density, b->textures[tex]));)
}
int rth_load_model(rth_builder* b, const char* path, double scale, int mat) {
    NEEDB; MAT(mat);
    if (!path) return bad(b, "null path");
    GUARD(return push_obj(b, load_model(path, scale, b->materials[mat]));)
}
int rth_bbox(rth_builder* b, int obj, double* o) {
    NEEDB; OBJ(obj);
    if (!o) return bad(b, "null output");
    const AABB& bb = b->objects[obj]->bounding_box();
    o[0] = bb.x.start; o[1] = bb.x.end; o[2] = bb.y.start; o[3] = bb.y.end; o[4] = bb.z.start; o[5] = bb.z.end;
    return 0;
}
int rth_extent(rth_builder* b, int obj, double* o) {
    NEEDB; OBJ(obj);
    if (!o) return bad(b, "null output");
    double lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
    b->objects[obj]->extent(lo, hi);
    for (int c = 0; c < 3; ++c) { o[2 * c] = lo[c]; o[2 * c + 1] = hi[c]; }
    return 0;
}
int rth_set_world(rth_builder* b, int obj) {
    NEEDB; OBJ(obj);
    GUARD(
        b->flat = flatten(*b->objects[obj]);
        b->texels.clear(); b->image_desc.clear(); b->perlin_vec.clear(); b->perlin_perm.clear();
        for (auto& im : b->flat.images) {
            b->image_desc.push_back((uint32_t)(b->texels.size() / 3));
            b->image_desc.push_back(im.w);
            b->image_desc.push_back(im.h);
            b->texels.insert(b->texels.end(), im.rgb.begin(), im.rgb.end());
        }
        for (auto& pn : b->flat.perlins) {
            b->perlin_vec.insert(b->perlin_vec.end(), pn.vec.begin(), pn.vec.end());
            b->perlin_perm.insert(b->perlin_perm.end(), pn.perm.begin(), pn.perm.end());
        }
        b->have_flat = true;
        return 0;)
}
uint32_t rth_num_prim_ids(rth_builder* b) { return (b && b->have_flat) ? b->flat.n_prim_ids : 0; }
const void* rth_flat_array(rth_builder* b, int what, uint64_t* count) {
    if (count) *count = 0;
    if (!b || !count || !b->have_flat) return nullptr;
    const FlatScene& f = b->flat;
#define RET(vec, per) { *count = (vec).size() / (per); return (vec).data(); }
    switch (what) {
        case 0: RET(f.nodes, 1)
        case 1: RET(f.boxes, 6)
        case 2: RET(f.spheres, 4)
        case 3: RET(f.quads, 16)
        case 4: RET(f.triangles, 16)
        case 5: RET(f.quadrics, 10)
        case 6: RET(f.media, 1)
        case 7: RET(f.ops, 1)
        case 8: RET(f.chains, 1)
        case 9: RET(f.materials, 1)
        case 10: RET(f.textures, 1)
        case 11: RET(f.sphere_id, 1)
        case 12: RET(f.quad_id, 1)
        case 13: RET(f.tri_id, 1)
        case 14: RET(f.quadric_id, 1)
        case 15: RET(f.sphere_mat, 1)
        case 16: RET(f.quad_mat, 1)
        case 17: RET(f.tri_mat, 1)
        case 18: RET(f.quadric_mat, 1)
        case 19: RET(b->texels, 3)
        case 20: RET(b->image_desc, 3)
        case 21: RET(b->perlin_vec, 768)
        case 22: RET(b->perlin_perm, 768)
        case 23: RET(f.cull_boxes, 6)
        default: return nullptr;
    }
#undef RET
}
rt_scene* rth_commit(rth_builder* b, int device) {
    if (!b) return nullptr;
    if (!b->have_flat) { b->err = "rth_set_world has not been called"; return nullptr; }
    try {
        rt_scene* s = b->flat.commit(device);
        if (!s) b->err = rt_last_error();
        return s;
    } catch (const std::exception& e) {
        b->err = e.what();
    } catch (...) {
        b->err = "unexpected failure";
    }
    return nullptr;
}
void rth_camera_new(double aspect_ratio, uint32_t image_width, double vfov, double defocus_angle, double focus_distance,
                    const double* look_from, const double* look_at, const double* view_up, rt_camera* out) {
    if (!look_from || !look_at || !view_up || !out) return;
    try {
        Camera c = Camera::create(aspect_ratio, image_width, ImageQuality{1, 1}, vfov, ThinLens{defocus_angle, focus_distance},
                                  CameraPosition{p3(look_from), p3(look_at), p3(view_up)}, Color(0, 0, 0));
        *out = c.cam;
    } catch (...) {
        std::memset(out, 0, sizeof *out);  // (an image of width 0 is refused by every render call)
    }
}
int rth_thrender(rth_builder* b, int world, const rt_camera* cam, uint32_t spp, uint32_t max_depth, const double* background,
                 uint64_t seed, int device, const char* output_path, int print_metrics, rt_stats* stats) {
    NEEDB; OBJ(world);
    if (!cam || !background || !output_path) return bad(b, "null argument");
    GUARD(
        Camera c = Camera::create(1.0, 1, ImageQuality{spp, max_depth}, 40.0, ThinLens{0.0, 1.0},
                                  CameraPosition{Point3(0, 0, 1), Point3(0, 0, 0), Point3(0, 1, 0)}, p3(background));
        c.cam = *cam;
        c.seed = seed;
        c.device = device;
        c.output_path = output_path;
        c.metrics = print_metrics ? stdout : nullptr;
        const int rc = c.thrender(*b->objects[world], stats);
        if (rc != 0) b->err = rt_last_error();
        return rc;)
}
int rth_read_image(const char* path, uint8_t* rgb8, uint64_t capacity, uint32_t* w, uint32_t* h) {
    if (!path || !w || !h) return -1;
    try {
        std::vector<uint8_t> rgb;
        if (!read_image_rgb8(path, rgb, *w, *h)) return -1;
        if (rgb8) {
            if (capacity < rgb.size()) return -1;
            std::memcpy(rgb8, rgb.data(), rgb.size());
        }
        return 0;
    } catch (...) {
        return -1;
    }
}
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h) {
    if (!path || !rgb8 || !w || !h) return -1;
    try {
        return write_png(path, rgb8, w, h) ? 0 : -1;
    } catch (...) {
        return -1;
    }
}

}  // extern "C"
