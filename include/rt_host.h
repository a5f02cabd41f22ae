/* rt_host.h -- C view of the C++ host layer (include/rt_host.hpp) for callers that cannot include C++: a handle
 * based builder with one call per constructor of the reference's scene-building API (the files under src/hittable,
 * src/material.rs, src/texture.rs, src/bvh.rs, src/camera.rs), the flatten step, and PNG output.  The Python package
 * drives it through ctypes.  Handles are small non-negative integers; every call returns < 0 on error and
 * rth_error() then holds the reason. */
#ifndef RT_HOST_H
#define RT_HOST_H

#include <stdint.h>

#include "rt_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct rth_builder rth_builder;

rth_builder* rth_new(void);
void rth_free(rth_builder* b);
const char* rth_error(rth_builder* b);

/* textures (texture.rs:19,37-57,88-113,138-158) */
int rth_tex_solid(rth_builder* b, double r, double g, double bl);
int rth_tex_checker(rth_builder* b, double scale, int even, int odd);
int rth_tex_image(rth_builder* b, const uint8_t* rgb8, uint32_t w, uint32_t h);
int rth_tex_image_file(rth_builder* b, const char* path); /* ImageTexture::new_or_fallback */
int rth_tex_perlin(rth_builder* b, double scale, const double* draws768);
/* materials (material.rs:70-80,106,168,177-183,192-194) */
int rth_mat_lambertian(rth_builder* b, int tex);
int rth_mat_metal(rth_builder* b, double r, double g, double bl, double fuzz);
int rth_mat_dielectric(rth_builder* b, double refraction_index);
int rth_mat_diffuse_light(rth_builder* b, int tex);
int rth_mat_isotropic(rth_builder* b, int tex);
/* primitives */
int rth_sphere(rth_builder* b, const double* center, double radius, int mat);                     /* sphere.rs:23 */
int rth_quad(rth_builder* b, const double* q, const double* u, const double* v, int mat);         /* parallelogram.rs:40 */
int rth_triangle(rth_builder* b, const double* q, const double* u, const double* v, int mat);     /* triangle.rs:39 */
int rth_y_cylinder(rth_builder* b, const double* center, double radius, int mat);                 /* quadric.rs:110 */
int rth_quadric_sphere(rth_builder* b, const double* center, double radius, int mat);             /* quadric.rs:138 */
int rth_y_cone(rth_builder* b, const double* center, const double* offset, int mat);              /* quadric.rs:161 */
/* collections: list handles are separate from object handles */
int rth_list_new(rth_builder* b);                                   /* HittableList::default */
int rth_list_add(rth_builder* b, int list, int obj);                /* HittableList::add / add_pointer */
int rth_box(rth_builder* b, const double* a, const double* c, int mat); /* create_box -> list handle */
int rth_slice(rth_builder* b, int list);                            /* to_hittable_slice -> object */
int rth_bvh(rth_builder* b, int list);                              /* to_bvh_node -> object */
int rth_translate(rth_builder* b, int obj, const double* offset);   /* Translate::new */
int rth_rotate_y(rth_builder* b, int obj, double degrees);          /* RotateY::new */
int rth_constant_medium(rth_builder* b, int boundary, double density, int tex); /* ConstantMedium::new */
int rth_load_model(rth_builder* b, const char* path, double scale, int mat);    /* load_model -> object */
int rth_bbox(rth_builder* b, int obj, double* out6);
/* Geometric bounds of the object's possible hit points (x0 x1 y0 y1 z0 z1; +-inf when unbounded, inf/-inf when
 * empty): what the cull boxes are made from.  Not part of the reference. */
int rth_extent(rth_builder* b, int obj, double* out6);
int rth_set_world(rth_builder* b, int obj); /* flattens the tree under obj */

/* flattened arrays of the current world (valid until the next rth_set_world / rth_free) */
uint32_t rth_num_prim_ids(rth_builder* b);
/* what: 0 nodes(rt_node) 1 boxes(6 f64) 2 spheres(4 f64) 3 quads(16 f64) 4 triangles(16 f64) 5 quadrics(10 f64)
 *       6 media(rt_medium) 7 ops(rt_xform_op) 8 chains(rt_chain) 9 materials(rt_material) 10 textures(rt_texture)
 *       11 sphere_id 12 quad_id 13 tri_id 14 quadric_id (u32) 15 sphere_mat 16 quad_mat 17 tri_mat 18 quadric_mat
 *       19 image texels (RGB8 pixels, images back to back) 20 image descriptors (first pixel, w, h: 3 u32)
 *       21 perlin vectors (768 f64 per table) 22 perlin permutations (768 u32 per table)
 *       23 cull boxes (6 f64, one per entry of 1) */
const void* rth_flat_array(rth_builder* b, int what, uint64_t* count);
/* FlatScene::commit: returns a committed rt_scene (caller destroys it with rt_scene_destroy) or NULL */
rt_scene* rth_commit(rth_builder* b, int device);

/* Camera::new (camera.rs:80-143) */
void rth_camera_new(double aspect_ratio, uint32_t image_width, double vfov, double defocus_angle,
                    double focus_distance, const double* look_from, const double* look_at, const double* view_up,
                    rt_camera* out);
/* Camera::thrender (camera.rs:248-271) with a camera already derived: flattens the tree under `world`, renders `spp`
 * samples per pixel on GPU `device` and writes the tone-mapped image as PNG to output_path (the reference's is the
 * fixed "images/image.png", whose directory must exist).  print_metrics != 0 prints one JSON line of what was traced
 * to stdout, where the reference prints banners and a progress bar.  stats may be NULL. */
int rth_thrender(rth_builder* b, int world, const rt_camera* cam, uint32_t spp, uint32_t max_depth,
                 const double* background, uint64_t seed, int device, const char* output_path, int print_metrics,
                 rt_stats* stats);
/* The decoder behind ImageTexture::new_or_fallback (texture.rs:89-93): binary PPM or PNG -> RGB8, top row first.
 * rgb8 may be NULL to ask for the size only; -1 when the file is neither or rgb8 is too small. */
int rth_read_image(const char* path, uint8_t* rgb8, uint64_t capacity, uint32_t* w, uint32_t* h);
int rth_write_png(const char* path, const uint8_t* rgb8, uint32_t w, uint32_t h);

#ifdef __cplusplus
}
#endif
#endif
