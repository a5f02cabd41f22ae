// rt_amd.hip -- kernels and C ABI of librt_amd.so (see include/rt_amd.h).
//
// Wavefront path tracer for gfx950: one path per lane, path state in SoA queues in HBM, per bounce one
// traverse+intersect launch and one shade+compact launch.  Build with -ffp-contract=off.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "rt_core.h"

using namespace rt;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                    \
    do {                                                                                                 \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess) return fail(RT_ERR_DEVICE, "%s: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                                          __FILE__, __LINE__);                                           \
    } while (0)

// ------------------------------------------------------------------------------------------------ kernels
#define TRAV_BLOCK 64    // one wavefront per workgroup: no barriers, LDS interval slots per lane
#define SHADE_BLOCK 256

// Path state queue, SoA: consecutive lanes read consecutive elements of every array.
struct Queue {
    double *ox, *oy, *oz, *dx, *dy, *dz;  // ray
    double *br, *bg, *bb;                 // throughput (product of attenuations so far)
    uint32_t* path;                       // index of the path inside the batch = s_local * npix + pixel
    uint32_t* ctr;                        // next draw index of the path's random stream
};
struct HitBuf {
    double* t;
    uint32_t* ref;
    uint32_t* chain;
};
struct BatchInfo {
    uint32_t npix, width;
    uint32_t sample0;  // global sample index of s_local = 0
    uint64_t seed;
};

extern __shared__ double2 g_levels[];

__global__ void __launch_bounds__(TRAV_BLOCK) k_raygen(rt_camera cam, BatchInfo bi, uint32_t n, Queue q) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= n) return;
    uint32_t pixel = idx % bi.npix, s_local = idx / bi.npix;
    Rng rng = rng_make(bi.seed, pixel, bi.sample0 + s_local, 0u);
    V3 o, d;
    camera_ray(cam, pixel % bi.width, pixel / bi.width, rng, o, d);
    q.ox[idx] = o.x; q.oy[idx] = o.y; q.oz[idx] = o.z;
    q.dx[idx] = d.x; q.dy[idx] = d.y; q.dz[idx] = d.z;
    q.br[idx] = 1.0; q.bg[idx] = 1.0; q.bb[idx] = 1.0;
    q.path[idx] = idx;
    q.ctr[idx] = rng.ctr;
}

// world.hit(ray, 0.001..inf) for every queued path (camera.rs:363).
template <bool COUNT>
__global__ void __launch_bounds__(TRAV_BLOCK) k_traverse(DScene S, BatchInfo bi, uint32_t n, Queue q, HitBuf hb,
                                                         unsigned long long* tallies) {
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally; tally.node_tests = 0; tally.prim_tests = 0; tally.prim_bytes = 0;
    if (idx < n) {
        uint32_t p = q.path[idx];
        Rng rng = rng_make(bi.seed, p % bi.npix, bi.sample0 + p / bi.npix, q.ctr[idx]);
        Ray w = ray_world(mk(q.ox[idx], q.oy[idx], q.oz[idx]), mk(q.dx[idx], q.dy[idx], q.dz[idx]));
        Best b = closest_hit<COUNT>(S, w, 0.001, RT_INF, L, rng, tally);
        hb.t[idx] = b.t; hb.ref[idx] = b.ref; hb.chain[idx] = b.chain;
        q.ctr[idx] = rng.ctr;
    }
    if (COUNT) {
        unsigned long long a = tally.node_tests, b = tally.prim_tests, c = tally.prim_bytes;
        for (int off = 32; off > 0; off >>= 1) { a += __shfl_down(a, off); b += __shfl_down(b, off); c += __shfl_down(c, off); }
        if ((threadIdx.x & 63) == 0) { atomicAdd(&tallies[0], a); atomicAdd(&tallies[1], b); atomicAdd(&tallies[2], c); }
    }
}

// Material::emitted / scatter on the closest hit, throughput update, termination, and compaction of the
// surviving paths into the next queue (camera.rs:349-408 unrolled into a loop).  A path that ends writes its
// radiance once into lum[] at its batch index.
__global__ void __launch_bounds__(SHADE_BLOCK) k_shade(DScene S, BatchInfo bi, uint32_t n, Queue q, HitBuf hb,
                                                       Queue out, uint32_t* out_count, double* lum_r, double* lum_g,
                                                       double* lum_b, double bgr, double bgg, double bgb,
                                                       int last_bounce) {
    __shared__ uint32_t wave_total[SHADE_BLOCK / 64];
    __shared__ uint32_t block_base;
    uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    bool alive = false;
    V3 no = mk(0, 0, 0), nd = mk(0, 0, 0), beta = mk(0, 0, 0);
    uint32_t p = 0, ctr = 0;
    if (idx < n) {
        p = q.path[idx];
        beta = mk(q.br[idx], q.bg[idx], q.bb[idx]);
        uint32_t ref = hb.ref[idx];
        if (ref == RT_NO_HIT) {
            lum_r[p] = beta.x * bgr; lum_g[p] = beta.y * bgg; lum_b[p] = beta.z * bgb;
        } else {
            Ray w = ray_world(mk(q.ox[idx], q.oy[idx], q.oz[idx]), mk(q.dx[idx], q.dy[idx], q.dz[idx]));
            Best b; b.t = hb.t[idx]; b.ref = ref; b.chain = hb.chain[idx];
            Rec rec = finalize<true>(S, w, b);
            Rng rng = rng_make(bi.seed, p % bi.npix, bi.sample0 + p / bi.npix, q.ctr[idx]);
            V3 att, emit;
            const rt_material m = S.mats[rec.mat];
            if (!scatter(S, m, w.d, rec, rng, nd, att, emit)) {
                lum_r[p] = beta.x * emit.x; lum_g[p] = beta.y * emit.y; lum_b[p] = beta.z * emit.z;
            } else if (last_bounce) {
                // the scattered ray would be traced with depth 0: ray_color returns black (camera.rs:350-352)
                lum_r[p] = 0.0; lum_g[p] = 0.0; lum_b[p] = 0.0;
            } else {
                beta = beta * att;
                no = rec.p;
                ctr = rng.ctr;
                alive = true;
            }
        }
    }
    // compaction: ballot + prefix count per wave, one atomic per workgroup
    unsigned long long mask = __ballot(alive);
    uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t prefix = __popcll(mask & ((1ull << lane) - 1ull));
    if (lane == 0) wave_total[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tot = 0;
        for (int k = 0; k < SHADE_BLOCK / 64; ++k) { uint32_t c = wave_total[k]; wave_total[k] = tot; tot += c; }
        block_base = tot ? atomicAdd(out_count, tot) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t dst = block_base + wave_total[wave] + prefix;
        out.ox[dst] = no.x; out.oy[dst] = no.y; out.oz[dst] = no.z;
        out.dx[dst] = nd.x; out.dy[dst] = nd.y; out.dz[dst] = nd.z;
        out.br[dst] = beta.x; out.bg[dst] = beta.y; out.bb[dst] = beta.z;
        out.path[dst] = p;
        out.ctr[dst] = ctr;
    }
}

// film[pix] += lum[s][pix] for s = 0.. in sample order: the running sum of camera.rs:258-264 (Point3::sum folds
// from -0.0, point3/mod.rs:96-109; the host initialises the film with -0.0).
__global__ void k_film_accumulate(const double* lum_r, const double* lum_g, const double* lum_b, uint32_t npix,
                                  uint32_t nsamples, double* film) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    double r = film[3u * pix], g = film[3u * pix + 1u], b = film[3u * pix + 2u];
    for (uint32_t s = 0; s < nsamples; ++s) {
        size_t k = (size_t)s * npix + pix;
        r = r + lum_r[k]; g = g + lum_g[k]; b = b + lum_b[k];
    }
    film[3u * pix] = r; film[3u * pix + 1u] = g; film[3u * pix + 2u] = b;
}
__global__ void k_fill(double* p, size_t n, double v) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
// pixel_color / spp then proccess_color (camera.rs:265-267, color.rs:49-68)
__global__ void k_tonemap(const double* film, uint32_t npix, double inv_spp, uint8_t* rgb) {
    uint32_t pix = blockIdx.x * blockDim.x + threadIdx.x;
    if (pix >= npix) return;
    for (int c = 0; c < 3; ++c) rgb[3u * pix + c] = to_byte(film[3u * pix + c] * inv_spp);
}

// Hittable::hit for a batch of caller-supplied rays, full HitRecord out.
__global__ void __launch_bounds__(TRAV_BLOCK) k_intersect_batch(DScene S, uint64_t seed, const double* o3,
                                                                const double* d3, const double* tmin,
                                                                const double* tmax, const uint32_t* rng3, uint64_t n,
                                                                uint8_t* hit, uint32_t* prim, double* t, double* p3,
                                                                double* n3, double* uv2, uint8_t* front,
                                                                uint32_t* ctr_out) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally; tally.node_tests = 0; tally.prim_tests = 0; tally.prim_bytes = 0;
    Rng rng = rng_make(seed, rng3 ? rng3[3 * i] : (uint32_t)i, rng3 ? rng3[3 * i + 1] : 0u, rng3 ? rng3[3 * i + 2] : 0u);
    Ray w = ray_world(ld3(o3 + 3 * i), ld3(d3 + 3 * i));
    Best b = closest_hit<false>(S, w, tmin[i], tmax[i], L, rng, tally);
    bool h = b.ref != RT_NO_HIT;
    Rec rec;
    rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
    if (h) rec = finalize<true>(S, w, b);
    if (hit) hit[i] = h ? 1 : 0;
    if (prim) prim[i] = rec.prim_id;
    if (t) t[i] = h ? b.t : 0.0;
    if (p3) { p3[3 * i] = rec.p.x; p3[3 * i + 1] = rec.p.y; p3[3 * i + 2] = rec.p.z; }
    if (n3) { n3[3 * i] = rec.n.x; n3[3 * i + 1] = rec.n.y; n3[3 * i + 2] = rec.n.z; }
    if (uv2) { uv2[2 * i] = rec.u; uv2[2 * i + 1] = rec.v; }
    if (front) front[i] = rec.front ? 1 : 0;
    if (ctr_out) ctr_out[i] = rng.ctr;
}

// get_ray + ray_color for chosen (pixel, sample) pairs, one whole path per lane (camera.rs:349-408 as a loop).
__global__ void __launch_bounds__(TRAV_BLOCK) k_trace_samples(DScene S, rt_camera cam, uint64_t seed,
                                                              uint32_t max_depth, double bgr, double bgg, double bgb,
                                                              uint64_t n, const uint32_t* pixel, const uint32_t* sample,
                                                              double* radiance3, uint32_t* nrays) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally; tally.node_tests = 0; tally.prim_tests = 0; tally.prim_bytes = 0;
    Rng rng = rng_make(seed, pixel[i], sample[i], 0u);
    V3 o, d;
    camera_ray(cam, pixel[i] % cam.width, pixel[i] / cam.width, rng, o, d);
    V3 beta = mk(1.0, 1.0, 1.0), lum = mk(0, 0, 0);
    uint32_t rays = 0;
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        Ray w = ray_world(o, d);
        rays++;
        Best b = closest_hit<false>(S, w, 0.001, RT_INF, L, rng, tally);
        if (b.ref == RT_NO_HIT) { lum = beta * mk(bgr, bgg, bgb); break; }
        Rec rec = finalize<true>(S, w, b);
        V3 nd, att, emit;
        const rt_material m = S.mats[rec.mat];
        if (!scatter(S, m, w.d, rec, rng, nd, att, emit)) { lum = beta * emit; break; }
        beta = beta * att;
        o = rec.p;
        d = nd;
    }
    radiance3[3 * i] = lum.x; radiance3[3 * i + 1] = lum.y; radiance3[3 * i + 2] = lum.z;
    if (nrays) nrays[i] = rays;
}

__global__ void k_debug_arith(const double* a, const double* b, uint64_t n, double* quot, double* root, double* recip,
                              double* logv) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    quot[i] = a[i] / b[i];
    root[i] = sqrt(a[i]);
    recip[i] = 1.0 / a[i];
    logv[i] = log(a[i]);
}

// ------------------------------------------------------------------------------------------------ scene object
struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
};
struct rt_scene {
    std::vector<rt_material> mats;
    std::vector<rt_texture> texs;
    struct Image { std::vector<uint8_t> rgb; uint32_t w = 0, h = 0; bool set = false; };
    std::vector<Image> images;
    struct Perlin { std::vector<double> vec; std::vector<uint32_t> perm; bool set = false; };
    std::vector<Perlin> perlins;
    std::vector<double> spheres, quads, tris, quadrics, boxes;
    std::vector<uint32_t> sphere_mat, sphere_id, quad_mat, quad_id, tri_mat, tri_id, quadric_mat, quadric_id;
    std::vector<rt_medium> media;
    std::vector<rt_xform_op> ops;
    std::vector<rt_chain> chains;
    std::vector<rt_node> nodes;

    bool committed = false;
    int device = -1;
    uint32_t levels = 0;  // interval slots per lane the walk needs
    std::vector<void*> allocs;
    DScene d;
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    // render workspace, grown on demand
    size_t ws_paths = 0;
    std::vector<void*> ws_allocs;
    Queue qa, qb;
    HitBuf hb;
    double *lum_r = nullptr, *lum_g = nullptr, *lum_b = nullptr;
    uint32_t* d_count = nullptr;
    unsigned long long* d_tallies = nullptr;
    uint32_t* h_count = nullptr;  // pinned
};

static void free_list(std::vector<void*>& v) {
    for (void* p : v) (void)hipFree(p);
    v.clear();
}

extern "C" {

const char* rt_last_error(void) { return g_err.c_str(); }

rt_scene* rt_scene_create(void) {
    rt_scene* s = new (std::nothrow) rt_scene();
    if (!s) { g_err = "out of memory"; return nullptr; }
    std::memset(&s->d, 0, sizeof s->d);
    return s;
}

void rt_scene_destroy(rt_scene* s) {
    if (!s) return;
    if (s->device >= 0) {
        (void)hipSetDevice(s->device);
        free_list(s->allocs);
        free_list(s->ws_allocs);
        if (s->h_count) (void)hipHostFree(s->h_count);
        for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
        if (s->stream) (void)hipStreamDestroy(s->stream);
    }
    delete s;
}

#define NEED_UNCOMMITTED(s)                                                        \
    if (!(s)) return fail(RT_ERR_INVALID, "null scene");                           \
    if ((s)->committed) return fail(RT_ERR_STATE, "scene already committed")

int rt_scene_set_materials(rt_scene* s, const rt_material* m, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !m) return fail(RT_ERR_INVALID, "null materials");
    s->mats.assign(m, m + n);
    return RT_OK;
}
int rt_scene_set_textures(rt_scene* s, const rt_texture* t, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !t) return fail(RT_ERR_INVALID, "null textures");
    s->texs.assign(t, t + n);
    return RT_OK;
}
int rt_scene_set_image(rt_scene* s, uint32_t slot, const uint8_t* rgb8, uint32_t w, uint32_t h) {
    NEED_UNCOMMITTED(s);
    if (!rgb8 || !w || !h) return fail(RT_ERR_INVALID, "empty image");
    if (slot >= 4096) return fail(RT_ERR_LIMIT, "image slot %u too large", slot);
    if (s->images.size() <= slot) s->images.resize(slot + 1);
    s->images[slot].rgb.assign(rgb8, rgb8 + (size_t)3 * w * h);
    s->images[slot].w = w;
    s->images[slot].h = h;
    s->images[slot].set = true;
    return RT_OK;
}
int rt_scene_set_perlin(rt_scene* s, uint32_t slot, const double* vec256x3, const uint32_t* perm3x256) {
    NEED_UNCOMMITTED(s);
    if (!vec256x3 || !perm3x256) return fail(RT_ERR_INVALID, "null perlin tables");
    if (slot >= 4096) return fail(RT_ERR_LIMIT, "perlin slot %u too large", slot);
    for (int i = 0; i < 768; ++i)
        if (perm3x256[i] > 255u) return fail(RT_ERR_INVALID, "perlin permutation entry %d out of range", i);
    if (s->perlins.size() <= slot) s->perlins.resize(slot + 1);
    s->perlins[slot].vec.assign(vec256x3, vec256x3 + 768);
    s->perlins[slot].perm.assign(perm3x256, perm3x256 + 768);
    s->perlins[slot].set = true;
    return RT_OK;
}
static int set_prims(rt_scene* s, std::vector<double>& dst, std::vector<uint32_t>& dmat, std::vector<uint32_t>& did,
                     const double* rec, uint32_t stride, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && (!rec || !mat || !id)) return fail(RT_ERR_INVALID, "null primitive arrays");
    if (n > RT_REF_INDEX_MASK) return fail(RT_ERR_LIMIT, "too many primitives of one type");
    dst.assign(rec, rec + (size_t)stride * n);
    dmat.assign(mat, mat + n);
    did.assign(id, id + n);
    return RT_OK;
}
int rt_scene_set_spheres(rt_scene* s, const double* cr, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->spheres, s->sphere_mat, s->sphere_id, cr, 4, mat, id, n);
}
int rt_scene_set_quads(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->quads, s->quad_mat, s->quad_id, rec16, 16, mat, id, n);
}
int rt_scene_set_triangles(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->tris, s->tri_mat, s->tri_id, rec16, 16, mat, id, n);
}
int rt_scene_set_quadrics(rt_scene* s, const double* rec10, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->quadrics, s->quadric_mat, s->quadric_id, rec10, 10, mat, id, n);
}
int rt_scene_set_media(rt_scene* s, const rt_medium* m, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !m) return fail(RT_ERR_INVALID, "null media");
    s->media.assign(m, m + n);
    return RT_OK;
}
int rt_scene_set_xforms(rt_scene* s, const rt_xform_op* ops, uint32_t n_ops, const rt_chain* chains,
                        uint32_t n_chains) {
    NEED_UNCOMMITTED(s);
    if ((n_ops && !ops) || (n_chains && !chains)) return fail(RT_ERR_INVALID, "null transform arrays");
    if (n_chains > 65535u) return fail(RT_ERR_LIMIT, "too many transform chains");
    s->ops.assign(ops, ops + n_ops);
    s->chains.assign(chains, chains + n_chains);
    return RT_OK;
}
int rt_scene_set_nodes(rt_scene* s, const rt_node* nodes, uint32_t n_nodes, const double* boxes6, uint32_t n_boxes) {
    NEED_UNCOMMITTED(s);
    if ((n_nodes && !nodes) || (n_boxes && !boxes6)) return fail(RT_ERR_INVALID, "null node arrays");
    s->nodes.assign(nodes, nodes + n_nodes);
    s->boxes.assign(boxes6, boxes6 + (size_t)6 * n_boxes);
    return RT_OK;
}

// Everything the kernels index without checking is checked here.
static int validate(rt_scene* s) {
    const uint32_t nmat = (uint32_t)s->mats.size(), ntex = (uint32_t)s->texs.size();
    for (uint32_t i = 0; i < ntex; ++i) {
        const rt_texture& t = s->texs[i];
        switch (t.kind) {
            case RT_TEX_SOLID: break;
            case RT_TEX_CHECKER:
                // children must precede their parent, which rules out cycles
                if (t.a >= i || t.b >= i) return fail(RT_ERR_INVALID, "texture %u: checker child must have a lower index", i);
                break;
            case RT_TEX_IMAGE:
                if (t.a >= s->images.size() || !s->images[t.a].set) return fail(RT_ERR_INVALID, "texture %u: image slot %u not set", i, t.a);
                break;
            case RT_TEX_PERLIN:
                if (t.a >= s->perlins.size() || !s->perlins[t.a].set) return fail(RT_ERR_INVALID, "texture %u: perlin slot %u not set", i, t.a);
                break;
            default: return fail(RT_ERR_INVALID, "texture %u: unknown kind %u", i, t.kind);
        }
    }
    for (uint32_t i = 0; i < nmat; ++i) {
        const rt_material& m = s->mats[i];
        if (m.kind > RT_MAT_ISOTROPIC) return fail(RT_ERR_INVALID, "material %u: unknown kind %u", i, m.kind);
        bool uses_tex = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        if (uses_tex && m.texture >= ntex) return fail(RT_ERR_INVALID, "material %u: texture %u out of range", i, m.texture);
    }
    auto check_mats = [&](const std::vector<uint32_t>& v, const char* what) -> int {
        for (size_t i = 0; i < v.size(); ++i)
            if (v[i] >= nmat) return fail(RT_ERR_INVALID, "%s %zu: material %u out of range", what, i, v[i]);
        return RT_OK;
    };
    int rc;
    if ((rc = check_mats(s->sphere_mat, "sphere"))) return rc;
    if ((rc = check_mats(s->quad_mat, "quad"))) return rc;
    if ((rc = check_mats(s->tri_mat, "triangle"))) return rc;
    if ((rc = check_mats(s->quadric_mat, "quadric"))) return rc;
    for (size_t i = 0; i < s->media.size(); ++i)
        if (s->media[i].material >= nmat) return fail(RT_ERR_INVALID, "medium %zu: material out of range", i);
    if (s->chains.empty()) { rt_chain c0 = {0, 0}; s->chains.push_back(c0); }
    if (s->chains[0].n_ops != 0) return fail(RT_ERR_INVALID, "chain 0 must be empty");
    for (size_t i = 0; i < s->chains.size(); ++i) {
        const rt_chain& c = s->chains[i];
        if ((uint64_t)c.first_op + c.n_ops > s->ops.size()) return fail(RT_ERR_INVALID, "chain %zu: ops out of range", i);
    }
    for (size_t i = 0; i < s->ops.size(); ++i)
        if (s->ops[i].kind > RT_XF_ROTATE_Y) return fail(RT_ERR_INVALID, "transform op %zu: unknown kind", i);

    const uint32_t nn = (uint32_t)s->nodes.size(), nbox = (uint32_t)(s->boxes.size() / 6);
    struct Open { uint32_t skip; bool medium; };
    std::vector<Open> open;
    uint32_t max_level = 0;
    for (uint32_t i = 0; i < nn; ++i) {
        while (!open.empty() && open.back().skip == i) open.pop_back();
        const rt_node& n = s->nodes[i];
        if (n.level != open.size()) return fail(RT_ERR_INVALID, "node %u: level %u but %zu enclosing entries", i, n.level, open.size());
        if (n.chain >= s->chains.size()) return fail(RT_ERR_INVALID, "node %u: chain out of range", i);
        max_level = std::max<uint32_t>(max_level, n.level);
        uint32_t limit = open.empty() ? nn : open.back().skip;
        bool in_medium = false;
        for (auto& o : open) in_medium |= o.medium;
        if (n.kind == RT_NODE_BVH || n.kind == RT_NODE_MEDIUM) {
            if (n.skip <= i || n.skip > limit) return fail(RT_ERR_INVALID, "node %u: skip %u outside (%u, %u]", i, n.skip, i, limit);
            if (n.kind == RT_NODE_BVH) {
                if (n.first >= nbox) return fail(RT_ERR_INVALID, "node %u: box out of range", i);
            } else {
                if (n.first >= s->media.size()) return fail(RT_ERR_INVALID, "node %u: medium out of range", i);
                if (in_medium) return fail(RT_ERR_LIMIT, "node %u: a medium inside a medium's boundary is not supported", i);
            }
            if (n.level + 2u > RT_MAX_LEVELS) return fail(RT_ERR_LIMIT, "node %u: nesting deeper than %d levels", i, RT_MAX_LEVELS);
            Open o = {n.skip, n.kind == RT_NODE_MEDIUM};
            open.push_back(o);
            max_level = std::max<uint32_t>(max_level, n.level + 1u);
        } else {
            size_t cap;
            switch (n.kind) {
                case RT_NODE_SPHERES: cap = s->sphere_mat.size(); break;
                case RT_NODE_QUADS: cap = s->quad_mat.size(); break;
                case RT_NODE_TRIANGLES: cap = s->tri_mat.size(); break;
                case RT_NODE_QUADRICS: cap = s->quadric_mat.size(); break;
                default: return fail(RT_ERR_INVALID, "node %u: unknown kind %u", i, n.kind);
            }
            if ((uint64_t)n.first + n.count > cap) return fail(RT_ERR_INVALID, "node %u: primitive run out of range", i);
        }
    }
    while (!open.empty() && open.back().skip == nn) open.pop_back();
    if (!open.empty()) return fail(RT_ERR_INVALID, "unterminated BVH/medium entry");
    s->levels = max_level + 1u;
    if (s->levels > RT_MAX_LEVELS) return fail(RT_ERR_LIMIT, "scene needs %u levels, limit %d", s->levels, RT_MAX_LEVELS);
    return RT_OK;
}

extern "C++" {
template <class T>
static int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);  // never hand a kernel a null table
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, bytes));
    s->allocs.push_back(p);
    HIP_TRY(hipMemset(p, 0, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RT_OK;
}
}  // extern "C++"

int rt_scene_commit(rt_scene* s, int device) {
    NEED_UNCOMMITTED(s);
    int rc = validate(s);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RT_ERR_DEVICE, "device %d not present (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    s->device = device;
    DScene& d = s->d;
    d.n_nodes = (uint32_t)s->nodes.size();
#define UP(vec, field) if ((rc = upload(s, vec, &d.field))) return rc
    UP(s->nodes, nodes); UP(s->boxes, boxes); UP(s->spheres, spheres); UP(s->quads, quads); UP(s->tris, tris);
    UP(s->quadrics, quadrics); UP(s->sphere_mat, sphere_mat); UP(s->sphere_id, sphere_id); UP(s->quad_mat, quad_mat);
    UP(s->quad_id, quad_id); UP(s->tri_mat, tri_mat); UP(s->tri_id, tri_id); UP(s->quadric_mat, quadric_mat);
    UP(s->quadric_id, quadric_id); UP(s->media, media); UP(s->ops, ops); UP(s->chains, chains); UP(s->mats, mats);
    UP(s->texs, texs);
    std::vector<uint8_t> texels;
    std::vector<uint32_t> desc;
    for (auto& im : s->images) {
        desc.push_back((uint32_t)(texels.size() / 3));
        desc.push_back(im.w);
        desc.push_back(im.h);
        texels.insert(texels.end(), im.rgb.begin(), im.rgb.end());
    }
    std::vector<double> pv;
    std::vector<uint32_t> pp;
    for (auto& pn : s->perlins) {
        if (pn.set) { pv.insert(pv.end(), pn.vec.begin(), pn.vec.end()); pp.insert(pp.end(), pn.perm.begin(), pn.perm.end()); }
        else { pv.insert(pv.end(), 768, 0.0); pp.insert(pp.end(), 768, 0u); }
    }
    UP(texels, texels); UP(desc, image_desc); UP(pv, perlin_vec); UP(pp, perlin_perm);
#undef UP
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc((void**)&s->h_count, 64, hipHostMallocDefault));
    HIP_TRY(hipMalloc((void**)&s->d_count, 64));
    s->allocs.push_back(s->d_count);
    HIP_TRY(hipMalloc((void**)&s->d_tallies, 64));
    s->allocs.push_back(s->d_tallies);
    s->committed = true;
    return RT_OK;
}

#define NEED_COMMITTED(s)                                                   \
    if (!(s)) return fail(RT_ERR_INVALID, "null scene");                    \
    if (!(s)->committed) return fail(RT_ERR_STATE, "scene not committed")

static size_t levels_lds(const rt_scene* s, uint32_t block) { return (size_t)s->levels * block * sizeof(double2); }

extern "C++" {
template <class T>
static int dev_copy_in(std::vector<void*>& keep, const T* host, size_t n, T** out) {
    *out = nullptr;
    if (!host) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)));
    keep.push_back(p);
    HIP_TRY(hipMemcpy(p, host, n * sizeof(T), hipMemcpyHostToDevice));
    *out = (T*)p;
    return RT_OK;
}
template <class T>
static int dev_alloc(std::vector<void*>& keep, bool want, size_t n, T** out) {
    *out = nullptr;
    if (!want) return RT_OK;
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(n * sizeof(T), 16)));
    keep.push_back(p);
    *out = (T*)p;
    return RT_OK;
}
}  // extern "C++"

int rt_intersect_batch(const rt_scene* s_, uint64_t seed, const double* o3, const double* d3, const double* tmin,
                       const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim, double* t,
                       double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    if (n == 0) return RT_OK;
    if (!o3 || !d3 || !tmin || !tmax) return fail(RT_ERR_INVALID, "null ray arrays");
    if (n > 0x7FFFFFFFull) return fail(RT_ERR_LIMIT, "batch too large");
    HIP_TRY(hipSetDevice(s->device));
    std::vector<void*> tmp;
    int rc = RT_OK;
    double *d_o, *d_d, *d_tmin, *d_tmax, *d_t, *d_p, *d_n, *d_uv;
    uint32_t *d_rng, *d_prim, *d_ctr;
    uint8_t *d_hit, *d_front;
    do {
        if ((rc = dev_copy_in(tmp, o3, 3 * n, &d_o))) break;
        if ((rc = dev_copy_in(tmp, d3, 3 * n, &d_d))) break;
        if ((rc = dev_copy_in(tmp, tmin, n, &d_tmin))) break;
        if ((rc = dev_copy_in(tmp, tmax, n, &d_tmax))) break;
        if ((rc = dev_copy_in(tmp, rng3, 3 * n, &d_rng))) break;
        if ((rc = dev_alloc(tmp, hit != nullptr, n, &d_hit))) break;
        if ((rc = dev_alloc(tmp, prim != nullptr, n, &d_prim))) break;
        if ((rc = dev_alloc(tmp, t != nullptr, n, &d_t))) break;
        if ((rc = dev_alloc(tmp, p3 != nullptr, 3 * n, &d_p))) break;
        if ((rc = dev_alloc(tmp, n3 != nullptr, 3 * n, &d_n))) break;
        if ((rc = dev_alloc(tmp, uv2 != nullptr, 2 * n, &d_uv))) break;
        if ((rc = dev_alloc(tmp, front != nullptr, n, &d_front))) break;
        if ((rc = dev_alloc(tmp, ctr_out != nullptr, n, &d_ctr))) break;
        uint32_t grid = (uint32_t)((n + TRAV_BLOCK - 1) / TRAV_BLOCK);
        hipLaunchKernelGGL(k_intersect_batch, dim3(grid), dim3(TRAV_BLOCK), levels_lds(s, TRAV_BLOCK), s->stream, s->d,
                           seed, d_o, d_d, d_tmin, d_tmax, d_rng, n, d_hit, d_prim, d_t, d_p, d_n, d_uv, d_front, d_ctr);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "intersect kernel: %s", hipGetErrorString(e)); break; }
#define BACK(hostp, devp, cnt) if (hostp && hipMemcpy(hostp, devp, (cnt) * sizeof(*hostp), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
        BACK(hit, d_hit, n); BACK(prim, d_prim, n); BACK(t, d_t, n); BACK(p3, d_p, 3 * n); BACK(n3, d_n, 3 * n);
        BACK(uv2, d_uv, 2 * n); BACK(front, d_front, n); BACK(ctr_out, d_ctr, n);
#undef BACK
    } while (0);
    free_list(tmp);
    return rc;
}

static int ws_queue(rt_scene* s, Queue& q, size_t n) {
    double** f[9] = {&q.ox, &q.oy, &q.oz, &q.dx, &q.dy, &q.dz, &q.br, &q.bg, &q.bb};
    for (auto pp : f) {
        HIP_TRY(hipMalloc((void**)pp, n * sizeof(double)));
        s->ws_allocs.push_back(*pp);
    }
    HIP_TRY(hipMalloc((void**)&q.path, n * sizeof(uint32_t))); s->ws_allocs.push_back(q.path);
    HIP_TRY(hipMalloc((void**)&q.ctr, n * sizeof(uint32_t))); s->ws_allocs.push_back(q.ctr);
    return RT_OK;
}
static int ensure_workspace(rt_scene* s, size_t paths) {
    if (s->ws_paths >= paths) return RT_OK;
    free_list(s->ws_allocs);
    s->ws_paths = 0;
    int rc;
    if ((rc = ws_queue(s, s->qa, paths))) return rc;
    if ((rc = ws_queue(s, s->qb, paths))) return rc;
    HIP_TRY(hipMalloc((void**)&s->hb.t, paths * sizeof(double))); s->ws_allocs.push_back(s->hb.t);
    HIP_TRY(hipMalloc((void**)&s->hb.ref, paths * sizeof(uint32_t))); s->ws_allocs.push_back(s->hb.ref);
    HIP_TRY(hipMalloc((void**)&s->hb.chain, paths * sizeof(uint32_t))); s->ws_allocs.push_back(s->hb.chain);
    HIP_TRY(hipMalloc((void**)&s->lum_r, paths * sizeof(double))); s->ws_allocs.push_back(s->lum_r);
    HIP_TRY(hipMalloc((void**)&s->lum_g, paths * sizeof(double))); s->ws_allocs.push_back(s->lum_g);
    HIP_TRY(hipMalloc((void**)&s->lum_b, paths * sizeof(double))); s->ws_allocs.push_back(s->lum_b);
    s->ws_paths = paths;
    return RT_OK;
}

static int check_render_args(const rt_scene* s, const rt_camera* cam, const rt_render_opts* o) {
    if (!cam || !o) return fail(RT_ERR_INVALID, "null camera or options");
    if (!cam->width || !cam->height) return fail(RT_ERR_INVALID, "empty image");
    if ((uint64_t)cam->width * cam->height > (1ull << 28)) return fail(RT_ERR_LIMIT, "image too large");
    if ((uint64_t)o->spp_begin + o->spp_count > 0xFFFFFFFFull) return fail(RT_ERR_INVALID, "sample range overflows");
    (void)s;
    return RT_OK;
}

int rt_render_device(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, void* film_dev,
                     rt_stats* stats) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    int rc = check_render_args(s, cam, o);
    if (rc) return rc;
    if (!film_dev) return fail(RT_ERR_INVALID, "null film");
    HIP_TRY(hipSetDevice(s->device));
    const uint32_t npix = cam->width * cam->height;
    double* film = (double*)film_dev;
    rt_stats st;
    std::memset(&st, 0, sizeof st);
    const bool count = (o->flags & RT_FLAG_COUNTERS) != 0;

    uint64_t want = o->batch_paths ? o->batch_paths : (16ull << 20);
    uint32_t spb = (uint32_t)std::max<uint64_t>(1, want / npix);  // samples per batch
    spb = std::min<uint32_t>(spb, std::max<uint32_t>(o->spp_count, 1u));
    if ((uint64_t)spb * npix > 0x7FFFFFFFull) spb = (uint32_t)(0x7FFFFFFFull / npix);
    if ((rc = ensure_workspace(s, (size_t)spb * npix))) return rc;

    hipStream_t sm = s->stream;
    HIP_TRY(hipEventRecord(s->ev[0], sm));
    hipLaunchKernelGGL(k_fill, dim3((3u * npix + 255u) / 256u), dim3(256), 0, sm, film, (size_t)3 * npix, -0.0);
    if (count) HIP_TRY(hipMemsetAsync(s->d_tallies, 0, 24, sm));
    const size_t lds = levels_lds(s, TRAV_BLOCK);

    for (uint32_t s0 = 0; s0 < o->spp_count; s0 += spb) {
        uint32_t ns = std::min(spb, o->spp_count - s0);
        uint32_t n = ns * npix;
        BatchInfo bi;
        bi.npix = npix; bi.width = cam->width; bi.sample0 = o->spp_begin + s0; bi.seed = o->seed;
        st.paths += n;
        Queue qin = s->qa, qout = s->qb;
        if (o->max_depth == 0) {
            HIP_TRY(hipMemsetAsync(s->lum_r, 0, (size_t)n * 8, sm));
            HIP_TRY(hipMemsetAsync(s->lum_g, 0, (size_t)n * 8, sm));
            HIP_TRY(hipMemsetAsync(s->lum_b, 0, (size_t)n * 8, sm));
        } else {
            hipLaunchKernelGGL(k_raygen, dim3((n + TRAV_BLOCK - 1) / TRAV_BLOCK), dim3(TRAV_BLOCK), 0, sm, *cam, bi, n, qin);
        }
        for (uint32_t bounce = 0; bounce < o->max_depth && n > 0; ++bounce) {
            st.rays += n;
            st.bounces = std::max(st.bounces, bounce + 1);
            HIP_TRY(hipMemsetAsync(s->d_count, 0, 4, sm));
            HIP_TRY(hipEventRecord(s->ev[1], sm));
            uint32_t tg = (n + TRAV_BLOCK - 1) / TRAV_BLOCK;
            if (count) hipLaunchKernelGGL(k_traverse<true>, dim3(tg), dim3(TRAV_BLOCK), lds, sm, s->d, bi, n, qin, s->hb, s->d_tallies);
            else hipLaunchKernelGGL(k_traverse<false>, dim3(tg), dim3(TRAV_BLOCK), lds, sm, s->d, bi, n, qin, s->hb, s->d_tallies);
            HIP_TRY(hipEventRecord(s->ev[2], sm));
            hipLaunchKernelGGL(k_shade, dim3((n + SHADE_BLOCK - 1) / SHADE_BLOCK), dim3(SHADE_BLOCK), 0, sm, s->d, bi, n, qin,
                               s->hb, qout, s->d_count, s->lum_r, s->lum_g, s->lum_b, o->background[0], o->background[1],
                               o->background[2], (bounce + 1 == o->max_depth) ? 1 : 0);
            HIP_TRY(hipEventRecord(s->ev[3], sm));
            HIP_TRY(hipMemcpyAsync(s->h_count, s->d_count, 4, hipMemcpyDeviceToHost, sm));
            HIP_TRY(hipStreamSynchronize(sm));
            HIP_TRY(hipGetLastError());
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
            st.ms_traverse += ms;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev[2], s->ev[3]));
            st.ms_shade += ms;
            st.traverse_launches++;
            n = *s->h_count;
            std::swap(qin, qout);
        }
        hipLaunchKernelGGL(k_film_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, sm, s->lum_r, s->lum_g, s->lum_b, npix, ns, film);
    }
    HIP_TRY(hipEventRecord(s->ev[3], sm));
    HIP_TRY(hipStreamSynchronize(sm));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[3]));
    st.ms_total = ms;
    if (count) {
        unsigned long long h[3];
        HIP_TRY(hipMemcpy(h, s->d_tallies, 24, hipMemcpyDeviceToHost));
        st.node_tests = h[0];
        st.prim_tests = h[1];
        st.prim_bytes = h[2];
    }
    if (stats) *stats = st;
    return RT_OK;
}

int rt_render(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, double* film_rgb, rt_stats* stats) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    int rc = check_render_args(s, cam, o);
    if (rc) return rc;
    if (!film_rgb) return fail(RT_ERR_INVALID, "null film");
    HIP_TRY(hipSetDevice(s->device));
    size_t bytes = (size_t)3 * cam->width * cam->height * sizeof(double);
    void* d_film = nullptr;
    HIP_TRY(hipMalloc(&d_film, bytes));
    rc = rt_render_device(s, cam, o, d_film, stats);
    if (rc == RT_OK && hipMemcpy(film_rgb, d_film, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "film copy back failed");
    (void)hipFree(d_film);
    return rc;
}

int rt_trace_samples(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, uint64_t n,
                     const uint32_t* pixel, const uint32_t* sample, double* radiance3, uint32_t* nrays) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    int rc = check_render_args(s, cam, o);
    if (rc) return rc;
    if (n == 0) return RT_OK;
    if (!pixel || !sample || !radiance3) return fail(RT_ERR_INVALID, "null sample arrays");
    if (n > 0x7FFFFFFFull) return fail(RT_ERR_LIMIT, "batch too large");
    const uint32_t npix = cam->width * cam->height;
    for (uint64_t i = 0; i < n; ++i)
        if (pixel[i] >= npix) return fail(RT_ERR_INVALID, "pixel %u out of range", pixel[i]);
    HIP_TRY(hipSetDevice(s->device));
    std::vector<void*> tmp;
    uint32_t *d_pix, *d_smp, *d_nr;
    double* d_rad;
    do {
        if ((rc = dev_copy_in(tmp, pixel, n, &d_pix))) break;
        if ((rc = dev_copy_in(tmp, sample, n, &d_smp))) break;
        if ((rc = dev_alloc(tmp, true, 3 * n, &d_rad))) break;
        if ((rc = dev_alloc(tmp, nrays != nullptr, n, &d_nr))) break;
        uint32_t grid = (uint32_t)((n + TRAV_BLOCK - 1) / TRAV_BLOCK);
        hipLaunchKernelGGL(k_trace_samples, dim3(grid), dim3(TRAV_BLOCK), levels_lds(s, TRAV_BLOCK), s->stream, s->d, *cam,
                           o->seed, o->max_depth, o->background[0], o->background[1], o->background[2], n, d_pix, d_smp,
                           d_rad, d_nr);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "trace kernel: %s", hipGetErrorString(e)); break; }
        if (hipMemcpy(radiance3, d_rad, 3 * n * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
        if (nrays && hipMemcpy(nrays, d_nr, n * sizeof(uint32_t), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
    } while (0);
    free_list(tmp);
    return rc;
}

int rt_tonemap_rgb8(const rt_scene* s_, const double* film_rgb, uint32_t width, uint32_t height, uint32_t spp,
                    uint8_t* rgb8) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    if (!film_rgb || !rgb8 || !width || !height || !spp) return fail(RT_ERR_INVALID, "bad tonemap arguments");
    HIP_TRY(hipSetDevice(s->device));
    size_t npix = (size_t)width * height;
    if (npix > (1ull << 28)) return fail(RT_ERR_LIMIT, "image too large");
    std::vector<void*> tmp;
    int rc;
    double* d_film;
    uint8_t* d_rgb;
    do {
        if ((rc = dev_copy_in(tmp, film_rgb, 3 * npix, &d_film))) break;
        if ((rc = dev_alloc(tmp, true, 3 * npix, &d_rgb))) break;
        hipLaunchKernelGGL(k_tonemap, dim3((uint32_t)((npix + 255) / 256)), dim3(256), 0, s->stream, d_film, (uint32_t)npix,
                           1.0 / (double)spp, d_rgb);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "tonemap kernel: %s", hipGetErrorString(e)); break; }
        if (hipMemcpy(rgb8, d_rgb, 3 * npix, hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
    } while (0);
    free_list(tmp);
    return rc;
}

int rt_debug_arith(int device, const double* a, const double* b, uint64_t n, double* quot, double* root, double* recip,
                   double* logv) {
    if (!a || !b || !quot || !root || !recip || !logv) return fail(RT_ERR_INVALID, "null arrays");
    if (n == 0) return RT_OK;
    if (n > 0x7FFFFFFFull) return fail(RT_ERR_LIMIT, "batch too large");
    HIP_TRY(hipSetDevice(device));
    std::vector<void*> tmp;
    int rc;
    double *d_a, *d_b, *d_q, *d_r, *d_c, *d_l;
    do {
        if ((rc = dev_copy_in(tmp, a, n, &d_a))) break;
        if ((rc = dev_copy_in(tmp, b, n, &d_b))) break;
        if ((rc = dev_alloc(tmp, true, n, &d_q))) break;
        if ((rc = dev_alloc(tmp, true, n, &d_r))) break;
        if ((rc = dev_alloc(tmp, true, n, &d_c))) break;
        if ((rc = dev_alloc(tmp, true, n, &d_l))) break;
        hipLaunchKernelGGL(k_debug_arith, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, 0, d_a, d_b, n, d_q, d_r, d_c, d_l);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipDeviceSynchronize();
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "arith kernel: %s", hipGetErrorString(e)); break; }
        hipError_t e2 = hipMemcpy(quot, d_q, n * 8, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess) e2 = hipMemcpy(root, d_r, n * 8, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess) e2 = hipMemcpy(recip, d_c, n * 8, hipMemcpyDeviceToHost);
        if (e2 == hipSuccess) e2 = hipMemcpy(logv, d_l, n * 8, hipMemcpyDeviceToHost);
        if (e2 != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
    } while (0);
    free_list(tmp);
    return rc;
}

}  // extern "C"
