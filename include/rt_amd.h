/* rt_amd.h -- C ABI of the MI355X path-tracing backend (librt_amd.so).
 *
 * The reference (villarjorge/raytracer, a pure-Rust crate) has no FFI of its own; the operator this library
 * sits behind is Camera::render / Camera::thrender (src/camera.rs:217,248), whose inputs are an immutable Camera
 * (src/camera.rs:23-37) and an immutable tree of `dyn Hittable` (src/hittable/mod.rs:65-68), and whose per-ray
 * contract is Hittable::hit(&Ray, &Range<f64>, &mut HitRecord) -> bool.  A host that keeps the reference's
 * scene-building API flattens that tree once (see raytracer_amd/csrc/host/) and hands the flat arrays over
 * through the rt_scene_set_* calls below; every comment names the reference item an entry point stands in for.
 *
 * Conventions: plain C, opaque handle, int status (0 = ok, < 0 = error; rt_last_error() gives the text for the
 * calling thread).  The caller owns every host buffer it passes; the library copies what it needs and owns all
 * device memory until rt_scene_destroy.  A handle is not re-entrant; distinct handles may be used from distinct
 * threads.  No exception crosses this boundary.  All arithmetic types are f64 as in the reference.
 */
#ifndef RT_AMD_H
#define RT_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RT_OK 0
#define RT_ERR_INVALID (-1) /* bad argument or inconsistent scene description */
#define RT_ERR_STATE (-2)   /* call order violated (e.g. render before commit) */
#define RT_ERR_DEVICE (-3)  /* HIP runtime error */
#define RT_ERR_LIMIT (-4)   /* scene exceeds a compiled-in limit */

#define RT_NO_HIT 0xFFFFFFFFu

typedef struct rt_scene rt_scene;

/* ---- materials and textures (src/material.rs, src/texture.rs) ---------------------------------------- */
enum rt_material_kind {
    RT_MAT_LAMBERTIAN = 0,    /* material.rs:41-68   texture */
    RT_MAT_METAL = 1,         /* material.rs:83-104  albedo, param = fuzz */
    RT_MAT_DIELECTRIC = 2,    /* material.rs:114-159 param = refraction index */
    RT_MAT_DIFFUSE_LIGHT = 3, /* material.rs:173-189 texture (emits, never scatters) */
    RT_MAT_ISOTROPIC = 4      /* material.rs:192-208 texture */
};
typedef struct rt_material {
    uint32_t kind;
    uint32_t texture; /* index into the texture table */
    double albedo[3];
    double param;
} rt_material;

enum rt_texture_kind {
    RT_TEX_SOLID = 0,   /* texture.rs:15-29   color */
    RT_TEX_CHECKER = 1, /* texture.rs:31-80   scale = inverse_scale, a = even texture, b = odd texture */
    RT_TEX_IMAGE = 2,   /* texture.rs:84-136  a = image slot */
    RT_TEX_PERLIN = 3   /* texture.rs:138-158 a = perlin slot, scale */
};
typedef struct rt_texture {
    uint32_t kind;
    uint32_t a;
    uint32_t b;
    uint32_t reserved;
    double color[3];
    double scale;
} rt_texture;

/* ---- instances (src/hittable/mod.rs:71-189) ----------------------------------------------------------- */
enum rt_xform_kind {
    RT_XF_TRANSLATE = 0, /* v = offset */
    RT_XF_ROTATE_Y = 1   /* v[0] = sin(theta), v[1] = cos(theta), evaluated by the host at construction */
};
typedef struct rt_xform_op {
    uint32_t kind;
    uint32_t reserved;
    double v[3];
} rt_xform_op;
/* A chain lists the instance transforms between the world and a node, outermost first.  Chain 0 is empty. */
typedef struct rt_chain {
    uint32_t first_op;
    uint32_t n_ops;
} rt_chain;

/* ---- participating media (src/hittable/constant_medium.rs:14-43) ------------------------------------- */
typedef struct rt_medium {
    double neg_inv_density;
    uint32_t material; /* the Isotropic phase function */
    uint32_t prim_id;
} rt_medium;

/* ---- flattened scene tree -------------------------------------------------------------------------------
 * The tree of BVHNode / HittableSlice / Translate / RotateY / ConstantMedium / primitives is laid out in the
 * order the reference's recursive hit() visits it (HittableSlice in list order, hittable_list.rs:88-102;
 * BVHNode::Internal left then right, bvh.rs:50-57).  Only three kinds of entry remain:
 *   BVH     one BVHNode (leaf or internal): `first` indexes its box; its descendants follow it directly and
 *           end at `skip`.
 *   MEDIUM  one ConstantMedium: `first` indexes the medium table; its boundary subtree follows and ends at
 *           `skip`.  A boundary subtree must not contain another medium.
 *   run     `count` consecutive primitives of one type starting at `first` in that type's array, in list order.
 *   CULL    no counterpart in the reference: `first` indexes a box that conservatively bounds every point the
 *           entries up to `skip` can report a hit at.  A ray whose interval misses it skips them; nothing is
 *           narrowed, so results are unchanged.  It does not count as a level.
 * Slices and instances leave no entry: a slice is just its children in sequence, an instance is a change of
 * `chain`.  `level` is the number of enclosing BVH/MEDIUM entries: an entry reads the ray interval of level
 * `level` and a BVH/MEDIUM entry writes the narrowed interval of level `level + 1` (bvh.rs:31-37).
 */
enum rt_node_kind {
    RT_NODE_BVH = 0,
    RT_NODE_MEDIUM = 1,
    RT_NODE_SPHERES = 2,
    RT_NODE_QUADS = 3,
    RT_NODE_TRIANGLES = 4,
    RT_NODE_QUADRICS = 5,
    RT_NODE_CULL = 6
};
typedef struct rt_node {
    uint8_t kind;
    uint8_t level;
    uint16_t chain;
    uint32_t skip; /* index of the first entry after this entry's subtree (BVH, MEDIUM); self + 1 for runs */
    uint32_t first;
    uint32_t count;
} rt_node;

#define RT_MAX_LEVELS 32 /* deepest `level` + 2 a scene may use */

/* ---- camera (src/camera.rs:23-37, the fields Camera::new derives at :80-143) ------------------------- */
typedef struct rt_camera {
    uint32_t width, height;
    double pixel00[3], delta_u[3], delta_v[3], center[3];
    double defocus_angle;
    double disk_u[3], disk_v[3];
} rt_camera;

typedef struct rt_render_opts {
    uint32_t spp_begin;   /* first global sample index of this call (spp sharding across GPUs) */
    uint32_t spp_count;   /* samples per pixel rendered by this call */
    uint32_t max_depth;   /* camera.rs:28 */
    uint32_t flags;       /* RT_FLAG_* */
    uint64_t seed;        /* key of the counter-based generator */
    double background[3]; /* camera.rs:36 */
    uint64_t batch_paths; /* paths in flight per wavefront batch; 0 = library default: 128 M (scenes without BVH entries)
                             or 256 M (with), about 200 to 410 bytes of workspace each (kept by the scene handle until
                             rt_scene_destroy), fewer where that workspace would not fit two fifths of the device memory
                             free when it has to grow */
} rt_render_opts;
#define RT_FLAG_COUNTERS 1u /* also count BVH box tests and primitive tests (slower) */
#define RT_FLAG_NO_CULL 2u  /* ignore the conservative cull boxes: do exactly the tests the reference does */
#define RT_FLAG_WHOLE_PATHS 4u /* scenes without BVH entries only (ignored otherwise): trace whole paths in one
                                  kernel, path state in registers, instead of the queue-based schedule (one traverse
                                  and one shade launch per bounce over all paths in flight).  Without either flag the
                                  library picks: whole paths where that kernel is lean (no BVH entries, no media,
                                  only Lambertian / DiffuseLight / Isotropic over solid colours), queues otherwise.
                                  The film is the same either way */
#define RT_FLAG_QUEUES 8u      /* always the queue-based schedule */
#define RT_FLAG_SPLIT_BOUNCE 16u /* queue-based schedule of a scene without BVH entries: one traverse and one shade
                                    launch per bounce, as for scenes with BVH entries, instead of the single bounce
                                    kernel that does both (the default there).  For a scene whose top level is a run of
                                    quads followed by nothing but eight-wide trees (a room around meshes): the quads
                                    get a traverse launch of their own ahead of the trees', in the list's own order,
                                    instead of being tested by the shade launches after them (the default there).  The
                                    film is the same either way.  Also: one shade launch per bounce for every
                                    material, where the default is two (plain materials first, the rest from a list) */

#define RT_FLAG_FAST_BVH 32u /* NOT the reference's result in every case.  BVH entries whose leaves the library walks
                                through its own eight-wide tree visit that tree's children nearest first instead of in the
                                reference's fixed order (upper half first, bvh.rs:50-57,127-130): fewer boxes and
                                primitives are looked at before the closest hit is known.  The closest hit is the same
                                except where two primitives are hit at exactly the same parameter: the reference's
                                order decides such ties (hittable_list.rs:94-96), this mode lets the first one found
                                win.  Never the default; bench.py reports the rate of differing primitive ids */

typedef struct rt_stats {
    uint64_t paths;      /* camera samples traced */
    uint64_t rays;       /* world.hit calls: one per bounce including the camera ray (camera.rs:363,429) */
    uint64_t node_tests; /* AABB::hit calls, only with RT_FLAG_COUNTERS (with RT_FLAG_NO_CULL as well: exactly the
                            reference's; without it: those the kernels still make, a leaf's box below an eight-wide tree) */
    uint64_t prim_tests; /* primitive / medium hit calls, only with RT_FLAG_COUNTERS */
    uint64_t cull_tests; /* conservative cull-box tests (not in the reference), only with RT_FLAG_COUNTERS */
    uint64_t prim_bytes; /* bytes of the primitive records those calls read (sphere 40, quad/triangle 136, quadric 88,
                            medium 16), only with RT_FLAG_COUNTERS */
    uint64_t prim_flops; /* algorithmic flops of those calls (sphere 35, quad/triangle 60, quadric 70, medium 30 each),
                            only with RT_FLAG_COUNTERS */
    double ms_total;     /* device time of the whole call */
    double ms_traverse;  /* summed duration of the traverse+intersect launches (HIP events) */
    double ms_shade;     /* summed duration of the shade+compact launches (with shade_quads != 0 they also test the top
                            level's quads; two launches per bounce where a scene mixes plain and other materials) */
    double ms_paths;     /* summed duration of the whole-path launches (scenes without BVH entries) */
    double ms_bounce;    /* summed duration of the bounce launches: traverse, shade and compaction in one kernel
                            (queue-based schedule of scenes without BVH entries) */
    double ms_finish;    /* duration of the launch that follows the last paths to their ends (at most one per call) */
    uint32_t traverse_launches;
    uint32_t path_launches;
    uint32_t bounce_launches;
    uint32_t bounces; /* most world.hit calls any one path made */
    uint32_t lockstep; /* 1 when the scene has no BVH entries: its kernels are k_bounce_lockstep (+ k_finish) /
                          k_traverse_lockstep / k_paths_lockstep, otherwise k_traverse_persistent */
    uint32_t finish_launches;
    uint64_t list_rays;  /* rays handed out through work lists to the launches that walk eight-wide trees, summed over
                            those launches (a ray counts once per tree whose bounds it can reach); 0 where no lists are used */
    uint64_t wide_tests; /* child boxes of the library's eight-wide trees tested (f32; none in the reference), only with
                            RT_FLAG_COUNTERS and the cull boxes on */
    uint64_t shade_quads; /* quads of the top level that the shade launches test per ray, ahead of shading: a run of quads
                            that precedes nothing but eight-wide trees (a room around meshes) is left to them, and the
                            traverse launches then walk the trees only; 0: the traverse launches test everything */
} rt_stats;

/* ---- lifetime ------------------------------------------------------------------------------------------ */
rt_scene* rt_scene_create(void);
void rt_scene_destroy(rt_scene* s);
const char* rt_last_error(void);

/* ---- scene description: what the Arc<dyn Hittable/Material/Texture> graph holds ---------------------- */
/* Arc<dyn Material> / Arc<dyn Texture> objects reachable from the scene (material.rs, texture.rs). */
int rt_scene_set_materials(rt_scene* s, const rt_material* m, uint32_t n);
int rt_scene_set_textures(rt_scene* s, const rt_texture* t, uint32_t n);
/* ImageTexture's RGB8 buffer (texture.rs:84-93), row-major, top row first. */
int rt_scene_set_image(rt_scene* s, uint32_t slot, const uint8_t* rgb8, uint32_t w, uint32_t h);
/* PerlinNoise tables (perlin.rs:11-16): 256 vectors (x,y,z) and the x/y/z permutations, 256 entries each. */
int rt_scene_set_perlin(rt_scene* s, uint32_t slot, const double* vec256x3, const uint32_t* perm3x256);
/* Sphere (sphere.rs:12-37): 4 f64 per sphere: center xyz, radius (already max(r, 0)). */
int rt_scene_set_spheres(rt_scene* s, const double* center_radius, const uint32_t* material,
                         const uint32_t* prim_id, uint32_t n);
/* Parallelogram (parallelogram.rs:12-58) / Triangle (triangle.rs:11-57): 16 f64 per primitive as the
 * constructors derive them: normal xyz, d, q xyz, u xyz, v xyz, w xyz. */
int rt_scene_set_quads(rt_scene* s, const double* rec16, const uint32_t* material, const uint32_t* prim_id,
                       uint32_t n);
int rt_scene_set_triangles(rt_scene* s, const double* rec16, const uint32_t* material, const uint32_t* prim_id,
                           uint32_t n);
/* Quadric (quadric.rs:14-21): 10 f64 per quadric: p1 xyz, p2 xyz, p3 xyz, j. */
int rt_scene_set_quadrics(rt_scene* s, const double* rec10, const uint32_t* material, const uint32_t* prim_id,
                          uint32_t n);
/* ConstantMedium table (constant_medium.rs:14-43). */
int rt_scene_set_media(rt_scene* s, const rt_medium* m, uint32_t n);
/* Translate / RotateY chains (hittable/mod.rs:71-189). */
int rt_scene_set_xforms(rt_scene* s, const rt_xform_op* ops, uint32_t n_ops, const rt_chain* chains,
                        uint32_t n_chains);
/* The flattened BVHNode / HittableSlice tree (bvh.rs:13-26, hittable_list.rs:72-75) and the BVHNode boxes
 * (aabb.rs:8-12), 6 f64 per box: x.start, x.end, y.start, y.end, z.start, z.end. */
int rt_scene_set_nodes(rt_scene* s, const rt_node* nodes, uint32_t n_nodes, const double* boxes6, uint32_t n_boxes);
/* Optional, no counterpart in the reference: for box k of rt_scene_set_nodes that belongs to a BVH entry, a box
 * (same 6-f64 layout, +-inf allowed) that conservatively bounds every point the entry's subtree can report a hit
 * at, in the entry's own ray space.  The reference's own boxes are inflated towards the origin (aabb.rs:29-33 seeds
 * every union with the point 1e-4) and must be kept for their narrowing; these let a ray skip a subtree it cannot
 * hit before that narrowing is done.  Call after rt_scene_set_nodes; n_boxes must match. */
int rt_scene_set_cull_boxes(rt_scene* s, const double* tight6, uint32_t n_boxes);
/* Validates the description and uploads it to HIP device `device`. */
int rt_scene_commit(rt_scene* s, int device);
/* RT_FLAG_FAST_BVH for every later call on this scene, rt_intersect_batch included (which takes no flags); on = 0
 * switches it off again.  No counterpart in the reference. */
int rt_scene_set_fast_bvh(rt_scene* s, int on);

/* ---- Hittable::hit on a batch of rays (hittable/mod.rs:65-68) ----------------------------------------
 * Ray i is (o3[3i..], d3[3i..]) with interval tmin[i]..tmax[i] (camera.rs:363 uses 0.001..inf).  rng3, if not
 * NULL, gives per ray the (pixel, sample, first draw index) of the random stream a ConstantMedium on the path
 * draws from; when NULL ray i uses (i, 0, 0).  Output arrays may be NULL.  prim is RT_NO_HIT on a miss.
 * All pointers are host pointers. */
int rt_intersect_batch(const rt_scene* s, uint64_t seed, const double* o3, const double* d3, const double* tmin,
                       const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim,
                       double* t, double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out);

/* The same for callers that cannot allocate device memory (tight serving loops, harnesses that forbid hipMalloc).  The
 * call is synchronous: it launches on the scene's own stream and waits for it, so it cannot itself be captured into a
 * caller's HIP graph.  No counterpart in the reference.  Every pointer is DEVICE memory on the scene's
 * device, owned by the caller; the call allocates nothing, on the device or through any library: what its kernels
 * need beyond these arrays (the closest hits travelling between launches, a work counter) lives in a static device
 * array of the library, 1 MiB + 256 bytes per GPU, held until the process ends.  Hence the limit: n <= 65536 rays per
 * call (RT_ERR_LIMIT beyond it); and calls that overlap in time on one GPU -- from any thread, on any scene handle --
 * share that array and race: issue them one after the other.
 *   od6_dev    6 * n f64: all origin x, then origin y, origin z, direction x, y, z (structure of arrays)
 *   tmin_dev, tmax_dev   n f64 each
 *   rng3_dev   3 * n u32: all pixels, then all samples, then all first draw indices; the draw indices are updated
 *              in place to the index after the last draw a ConstantMedium made (rt_intersect_batch's ctr_out)
 *   outputs    as rt_intersect_batch's, in device memory; any of them may be NULL
 * Results are those of rt_intersect_batch bit for bit.  The scene's stream is synchronised before the call returns. */
#define RT_NOALLOC_MAX_RAYS 65536u
int rt_intersect_batch_noalloc(const rt_scene* s, uint64_t seed, const double* od6_dev, const double* tmin_dev,
                               const double* tmax_dev, uint32_t* rng3_dev, uint32_t n, uint8_t* hit_dev,
                               uint32_t* prim_dev, double* t_dev, double* p3_dev, double* n3_dev, double* uv2_dev,
                               uint8_t* front_dev);

/* ---- Camera::render / thrender (camera.rs:217-271) -----------------------------------------------------
 * Renders samples [spp_begin, spp_begin + spp_count) of every pixel and writes film_rgb[3*(j*width+i)..] = the
 * SUM of their radiances (not the average), in sample order, so that shards rendered on different GPUs add up
 * to the whole.  rt_render takes a host film buffer; rt_render_device takes a device buffer on the scene's
 * device (for a collective that runs on device memory).  stats may be NULL. */
int rt_render(const rt_scene* s, const rt_camera* cam, const rt_render_opts* opts, double* film_rgb, rt_stats* stats);
int rt_render_device(const rt_scene* s, const rt_camera* cam, const rt_render_opts* opts, void* film_rgb_device,
                     rt_stats* stats);

/* The same on several GPUs from ONE process: scenes[r] is the same scene committed to device r (rt_scene_commit
 * with different devices; the same device may appear more than once).  The samples [spp_begin, spp_begin +
 * spp_count) are split into n contiguous shards (the remainder goes to the low ranks, unlike camera.rs:324, which
 * drops it), every shard is rendered on its device by a host thread of its own, and the shard films are summed on
 * scenes[0]'s device: with one RCCL reduce when every scene has a device of its own (librccl is loaded at the first
 * such call; the sum's association is then RCCL's), else -- the same device more than once, or no RCCL to load, or
 * RT_MULTI_RCCL=0 in the environment -- by peer copies, added in rank order.  film_rgb is a host buffer.  stats
 * (nullable) holds sums over the shards, times being the slowest shard's.  The multi-process equivalent -- one
 * process per GPU and an all-reduce of rt_render_device's film -- needs no call of its own. */
int rt_render_multi(const rt_scene* const* scenes, int n, const rt_camera* cam, const rt_render_opts* opts,
                    double* film_rgb, rt_stats* stats);
/* How many of this process's rt_render_multi calls had their films summed by RCCL (diagnostic). */
uint64_t rt_multi_rccl_reductions(void);
/* Destroys the RCCL communicators rt_render_multi keeps between calls (one set per list of devices).  Call it before
 * the process winds down its HIP runtime -- at library unload that runtime may already be gone, so nothing is destroyed
 * there; a later rt_render_multi creates its communicators anew.  No counterpart in the reference. */
void rt_multi_shutdown(void);

/* Per-sample radiance of chosen (pixel, sample) pairs: get_ray + ray_color for each (camera.rs:258-264).
 * radiance3 has 3 f64 per pair; nrays (nullable) the number of world.hit calls of each path. */
int rt_trace_samples(const rt_scene* s, const rt_camera* cam, const rt_render_opts* opts, uint64_t n,
                     const uint32_t* pixel, const uint32_t* sample, double* radiance3, uint32_t* nrays);

/* pixel_color / spp, then proccess_color (camera.rs:265-267, point3/color.rs:49-68): host film in, RGB8 out. */
int rt_tonemap_rgb8(const rt_scene* s, const double* film_rgb, uint32_t width, uint32_t height, uint32_t spp,
                    uint8_t* rgb8);

/* How many device allocations (hipMalloc) this process's calls into the library have made so far (diagnostic: the
 * test of rt_intersect_batch_noalloc reads it before and after a call).  No counterpart in the reference. */
uint64_t rt_debug_device_allocs(void);

/* Device self-check used by the tests: evaluates a/b, sqrt(a), 1/a, log(a) elementwise on the GPU so that the
 * host can compare the first three bit for bit (the intersection code relies on them being correctly rounded). */
int rt_debug_arith(int device, const double* a, const double* b, uint64_t n, double* quot, double* root,
                   double* recip, double* logv);

#ifdef __cplusplus
}
#endif
#endif /* RT_AMD_H */
