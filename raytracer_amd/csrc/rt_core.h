// rt_core.h -- per-lane device functions of the path tracer: vector math, the counter-based generator, the
// flattened-tree walk with the reference's interval semantics, hit-record reconstruction, textures and materials.
//
// Everything is f64 and keeps the reference's operation order (citations are reference file:line); the file
// must be compiled with -ffp-contract=off so that no multiply-add pair is fused.  Functions are
// __host__ __device__ only so that tests can single-step the same code on the CPU; the library itself never
// runs them on the host.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/rt_amd.h"

#define RT_HD __host__ __device__ __forceinline__

namespace rt {

#define RT_INF (__builtin_inf())
#define RT_PI 3.14159265358979323846264338327950288

// hit reference: kind in the top 3 bits, index in the type's array below
#define RT_REF_KIND_SHIFT 29
#define RT_REF_INDEX_MASK 0x1FFFFFFFu
RT_HD uint32_t make_ref(uint32_t kind, uint32_t index) { return (kind << RT_REF_KIND_SHIFT) | index; }

struct V3 {
    double x, y, z;
};
// src/point3/mod.rs:70-170
RT_HD V3 mk(double x, double y, double z) { V3 r; r.x = x; r.y = y; r.z = z; return r; }
RT_HD V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
RT_HD V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
RT_HD V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
RT_HD V3 operator*(V3 a, double s) { return mk(a.x * s, a.y * s, a.z * s); }
RT_HD V3 operator*(double s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
RT_HD V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
RT_HD V3 operator/(V3 a, double s) { return a * (1.0 / s); }  // :137-144: division is a multiply by 1/s
RT_HD double len2(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // :175-177
RT_HD double dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }  // :183-191
RT_HD V3 cross(V3 u, V3 v) {  // :194-200
    return mk(u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x);
}
RT_HD V3 rotate_y(V3 p, double c, double s) { return mk(c * p.x - s * p.z, p.y, s * p.x + c * p.z); }  // :203-209
RT_HD V3 unit(V3 u) { return u / sqrt(len2(u)); }  // :221-223
RT_HD bool near_zero(V3 a) { return fabs(a.x) < 1e-8 && fabs(a.y) < 1e-8 && fabs(a.z) < 1e-8; }  // :214-217
RT_HD V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }  // :279-281
RT_HD V3 refract(V3 uv, V3 n, double eta) {  // :284-293
    double cos_theta = fmin(dot(-uv, n), 1.0);
    V3 perp = (uv + n * cos_theta) * eta;
    V3 par = (-n) * sqrt(fabs(1.0 - len2(perp)));
    return perp + par;
}
RT_HD V3 ld3(const double* p) { return mk(p[0], p[1], p[2]); }

// ------------------------------------------------------------------------------------------------------------
// Counter-based generator: Philox4x32-10 keyed by the render seed, counter = (pixel, sample, draw block, stream).
// Draw d uses block d >> 1: words {0,1} when d is even, {2,3} when odd; 52 bits become a double in [0,1) the way
// rand 0.9's UniformFloat<f64> does.  Stands in for the rand::random_range call sites (camera.rs:503-504,
// point3/mod.rs:231-233,266-267, material.rs:139, constant_medium.rs:72).
struct Rng {
    uint32_t k0, k1, pixel, sample, ctr;
    uint32_t blk;   // block held in w[], 0xFFFFFFFF when none
    uint32_t w[4];
};
RT_HD Rng rng_make(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr) {
    Rng r;
    r.k0 = (uint32_t)seed;
    r.k1 = (uint32_t)(seed >> 32);
    r.pixel = pixel;
    r.sample = sample;
    r.ctr = ctr;
    r.blk = 0xFFFFFFFFu;
    r.w[0] = r.w[1] = r.w[2] = r.w[3] = 0;
    return r;
}
RT_HD void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];  // one 32x32->64 multiply each
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0; c[1] = (uint32_t)p1; c[2] = n2; c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}
// (bits >> 12) * 2^-52, the 52-bit uniform of rand's UniformFloat<f64>: the 52 bits become the mantissa of a double in
// [1, 2) and 1 is taken off, which is exact and costs one subtraction instead of two conversions and a multiply.
RT_HD double unit_interval(uint64_t bits) {
    return __builtin_bit_cast(double, (bits >> 12) | 0x3FF0000000000000ull) - 1.0;
}
RT_HD double rng_next01(Rng& r) {
    uint32_t b = r.ctr >> 1;
    if (b != r.blk) {
        r.w[0] = r.pixel; r.w[1] = r.sample; r.w[2] = b; r.w[3] = 0u;
        philox4x32_10(r.w, r.k0, r.k1);
        r.blk = b;
    }
    uint64_t bits = (r.ctr & 1u) ? (((uint64_t)r.w[3] << 32) | r.w[2]) : (((uint64_t)r.w[1] << 32) | r.w[0]);
    r.ctr++;
    return unit_interval(bits);
}
RT_HD double rng_range(Rng& r, double lo, double hi) { return rng_next01(r) * (hi - lo) + lo; }
RT_HD bool unit_vector_trial(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t c, V3& p);
RT_HD V3 random_unit_vector(Rng& r) {  // point3/mod.rs:229-250
    // Draws ctr, ctr + 1, ctr + 2 of every trial lie in blocks ctr >> 1 and (ctr >> 1) + 1: both are generated for
    // all lanes alike and the three values picked by the parity of ctr.  Going through rng_next01 instead, lanes of
    // either parity would need their next block at different draws and a wavefront would run the generator three
    // times per trial, each time for half of its lanes.
    for (;;) {
        V3 p;
        const bool ok = unit_vector_trial(r.k0, r.k1, r.pixel, r.sample, r.ctr, p);
        r.ctr += 3u;
        r.blk = 0xFFFFFFFFu;
        if (ok) return unit(p);
    }
}
// One trial of that loop at draw position c of a stream: the candidate point and whether it is accepted.
RT_HD bool unit_vector_trial(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t c, V3& p) {
    const uint32_t b0 = c >> 1;
    uint32_t w0[4] = {pixel, sample, b0, 0u}, w1[4] = {pixel, sample, b0 + 1u, 0u};
    philox4x32_10(w0, k0, k1);
    philox4x32_10(w1, k0, k1);
    const uint64_t lo0 = ((uint64_t)w0[1] << 32) | w0[0], hi0 = ((uint64_t)w0[3] << 32) | w0[2];
    const uint64_t lo1 = ((uint64_t)w1[1] << 32) | w1[0], hi1 = ((uint64_t)w1[3] << 32) | w1[2];
    const bool odd = (c & 1u) != 0u;
    p.x = unit_interval(odd ? hi0 : lo0) * (1.0 - -1.0) + -1.0;
    p.y = unit_interval(odd ? lo1 : hi0) * (1.0 - -1.0) + -1.0;
    p.z = unit_interval(odd ? hi1 : lo1) * (1.0 - -1.0) + -1.0;
    return len2(p) < 1.0;
}
#if defined(__HIP_DEVICE_COMPILE__)
__device__ __forceinline__ double lane_pull(uint32_t src_lane, double v) {  // v of lane src_lane
    const uint64_t b = __builtin_bit_cast(uint64_t, v);
    const uint32_t lo = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)b);
    const uint32_t hi = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(src_lane << 2), (int)(uint32_t)(b >> 32));
    return __builtin_bit_cast(double, ((uint64_t)hi << 32) | lo);
}
// random_unit_vector for the lanes of a wavefront that `want` one, from each lane's own stream at `ctr` (advanced by
// three draws per trial consumed, exactly as the loop above would).  Has to be called by all 64 lanes together.
// The loop above keeps a whole wavefront going until its unluckiest lane is accepted (about six rounds for 64
// lanes at an acceptance rate of pi/6, with ever fewer lanes doing anything useful).  Here, once no more than half
// of the lanes are still waiting, the idle lanes run the waiting lanes' next trials for them -- lane j tries draw
// position ctr + 3 * (j / m) of the (j % m)-th waiting lane's stream, m being the number of waiting lanes -- and each
// waiting lane takes the first accepted of its trials: the same vector from the same draws in about three rounds.
__device__ inline V3 wave_unit_vector(bool want, uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t& ctr) {
    const uint32_t lane = __lane_id();
    const unsigned long long below = (1ull << lane) - 1ull;
    V3 p = mk(0.0, 0.0, 1.0);
    bool pending = want;
    for (;;) {
        const unsigned long long mask = __ballot(pending);
        if (mask == 0ull) break;
        const uint32_t m = (uint32_t)__popcll(mask);
        if (m > 32u) {  // not even two trials per waiting lane: everyone tries for itself
            if (pending) {
                V3 cand;
                const bool ok = unit_vector_trial(k0, k1, pixel, sample, ctr, cand);
                ctr += 3u;
                if (ok) { p = cand; pending = false; }
            }
            continue;
        }
        const uint32_t per = 64u / m < 8u ? 64u / m : 8u;          // trials per waiting lane this round
        const uint32_t rank = (uint32_t)__popcll(mask & below);     // waiting lanes before this one
        // slot s < m learns which lane is the s-th waiting one (the others send their ids past slot m)
        const uint32_t dst = pending ? rank : m + (lane - rank);
        const uint32_t slot_owner = (uint32_t)__builtin_amdgcn_ds_permute((int)(dst << 2), (int)lane);
        const uint32_t q = (lane * ((65536u + m - 1u) / m)) >> 16;  // lane / m for lane < 64, m <= 32
        const uint32_t r = lane - q * m;
        const uint32_t owner = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(r << 2), (int)slot_owner);
        const uint32_t o_pixel = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)pixel);
        const uint32_t o_sample = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)sample);
        const uint32_t o_ctr = (uint32_t)__builtin_amdgcn_ds_bpermute((int)(owner << 2), (int)ctr);
        V3 cand;
        const bool ok = unit_vector_trial(k0, k1, o_pixel, o_sample, o_ctr + 3u * q, cand) && q < per;
        const unsigned long long accepted = __ballot(ok);
        uint32_t src = lane, used = per;
        bool found = false;
        if (pending) {
            for (uint32_t t = 0; t < per; ++t) {
                const uint32_t at = rank + t * m;
                if (!found && ((accepted >> at) & 1ull)) { found = true; src = at; used = t + 1u; }
            }
            ctr += 3u * used;
        }
        const double fx = lane_pull(src, cand.x), fy = lane_pull(src, cand.y), fz = lane_pull(src, cand.z);
        if (found) { p = mk(fx, fy, fz); pending = false; }
    }
    return want ? unit(p) : p;
}
#elif defined(__HIPCC__)
__device__ double lane_pull(uint32_t src_lane, double v);
__device__ V3 wave_unit_vector(bool want, uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t& ctr);
#endif
// One draw at position (pixel, sample, ctr) without carrying a cached block around (the tree walk draws at most
// once per medium and must stay light on registers).
RT_HD double rng_draw_at(uint32_t k0, uint32_t k1, uint32_t pixel, uint32_t sample, uint32_t ctr) {
    uint32_t w[4] = {pixel, sample, ctr >> 1, 0u};
    philox4x32_10(w, k0, k1);
    uint64_t bits = (ctr & 1u) ? (((uint64_t)w[3] << 32) | w[2]) : (((uint64_t)w[1] << 32) | w[0]);
    return unit_interval(bits);
}
struct RngPos {  // what the walk needs of a path's random stream
    uint32_t k0, k1, pixel, sample, ctr;
};

// ------------------------------------------------------------------------------------------------------------
// Scene as the kernels see it: per-type arrays of packed f64 records plus the flattened tree.  Every record array
// starts 16-byte aligned and every record is a whole number of 16-byte pairs, so records are fetched as double2
// (one 16-byte load per pair; a quad or triangle is exactly one 128-byte line).
#define RT_MAX_SIG 2u
// The CULL entries a ray's signature covers (lockstep trees only), passed to the kernels that compute signatures as an
// argument of its own so that everything in it arrives in scalar registers.
struct SigInfo {
    uint32_t n;         // 0..RT_MAX_SIG
    double box[2][6];   // the entry's box (x, y, z: start, end)
    double xf[2][12];   // the map from world space into the entry's chain, p -> A p + b (A row-major in [0..8], b in
                        // [9..11]; directions take A alone); composed on the host, which a signature's tolerance allows
};
struct DScene {
    const uint4* nodes;      // 128 bytes (one cache line) per tree entry: rt_node header | box of a BVH or CULL entry
                             // (3 x double2) | conservative cull box of a BVH entry (3 x double2) | 16 bytes unused
    uint32_t n_nodes;
    const double* spheres;   // 4 per sphere: cx cy cz r
    const double* quads;     // 16 per quad: n(3) d q(3) u(3) v(3) w(3)
    const double* tris;      // 16 per triangle, same layout
    const double* quadrics;  // 10 per quadric: p1(3) p2(3) p3(3) j
    const uint32_t* sphere_mat; const uint32_t* sphere_id;
    const uint32_t* quad_mat;   const uint32_t* quad_id;
    const uint32_t* quad_axis;  // per quad: 1 + the axis its normal lies along when n and w are zero elsewhere, else 0
    const float* leaf_sub;      // per BVH leaf that has them (records, bytes 88..91): cull boxes of its pairs of primitives,
                                // 8 x 6 f32
    uint32_t n_variants;        // copies of the tree in `nodes` (1, 2 or 4), n_nodes entries each: copy m is the tree
                                // without the subtrees of the signature entries in mask m (a pruned CULL entry has
                                // count 1 and the lockstep walk passes its subtree over); indices in copy m count
                                // from m * n_nodes
    const uint32_t* tri_mat;    const uint32_t* tri_id;
    const uint32_t* quadric_mat; const uint32_t* quadric_id;
    const rt_medium* media;
    const rt_xform_op* ops;
    const rt_chain* chains;
    const rt_material* mats;
    const rt_texture* texs;
    const uint8_t* texels;        // all images back to back
    const uint32_t* image_desc;   // per slot: first pixel, width, height
    const double* perlin_vec;     // per slot 256*3
    const uint32_t* perlin_perm;  // per slot 3*256
    uint32_t n_perlin;            // perlin textures in the texture table (0: the shade kernel skips its deferred evaluation)
};

struct Node {  // rt_node unpacked
    uint32_t kind, level, chain, skip, first, count;
};
RT_HD Node load_node(const DScene& S, uint32_t i) {
    const uint4 raw = S.nodes[8u * i];
    Node n;
    n.kind = raw.x & 0xFFu; n.level = (raw.x >> 8) & 0xFFu; n.chain = raw.x >> 16;
    n.skip = raw.y; n.first = raw.z; n.count = raw.w;
    return n;
}

struct Ray {  // src/ray.rs:4-8; `a` caches direction.length_squared()
    V3 o, d, inv;
    double a;
    bool inv_ok;
};
RT_HD V3 ray_at(const Ray& r, double t) { return r.o + r.d * t; }  // ray.rs:11-13
RT_HD void ray_need_inv(Ray& r) {  // ray.rs:14-19; only AABB::hit reads it
    if (!r.inv_ok) {
        r.inv = mk(1.0 / r.d.x, 1.0 / r.d.y, 1.0 / r.d.z);
        r.inv_ok = true;
    }
}
RT_HD Ray ray_world(V3 o, V3 d) {
    Ray r;
    r.o = o; r.d = d; r.a = len2(d); r.inv_ok = false; r.inv = mk(0, 0, 0);
    return r;
}
// The ray an object under `chain` sees: Translate::hit (hittable/mod.rs:80) and RotateY::hit (:121-124) applied
// outermost first.
RT_HD Ray ray_in_chain(const DScene& S, const Ray& w, uint32_t chain) {
    Ray r = w;
    if (chain == 0) return r;
    rt_chain c = S.chains[chain];
    bool rotated = false;
    for (uint32_t k = 0; k < c.n_ops; ++k) {
        const rt_xform_op& op = S.ops[c.first_op + k];
        if (op.kind == RT_XF_TRANSLATE) {
            r.o = r.o - mk(op.v[0], op.v[1], op.v[2]);
        } else {
            r.o = rotate_y(r.o, op.v[1], op.v[0]);
            r.d = rotate_y(r.d, op.v[1], op.v[0]);
            rotated = true;
        }
    }
    if (rotated) {
        r.a = len2(r.d);
        r.inv_ok = false;
    }
    return r;
}
// Where a lane re-reads its world-space ray from when the walk changes chain (kept out of registers meanwhile).
struct RaySoA {  // path queue
    const double *ox, *oy, *oz, *dx, *dy, *dz;
    uint32_t idx;
    RT_HD Ray world() const { return ray_world(mk(ox[idx], oy[idx], oz[idx]), mk(dx[idx], dy[idx], dz[idx])); }
};
struct RayAoS {  // caller-supplied batches, 3 f64 per ray
    const double *o3, *d3;
    uint64_t i;
    RT_HD Ray world() const { return ray_world(ld3(o3 + 3 * i), ld3(d3 + 3 * i)); }
};
struct RayVal {
    V3 o, d;
    RT_HD Ray world() const { return ray_world(o, d); }
};

// Per-lane array of ray intervals, one slot per nesting level of BVH / medium entries.  `stride` is the number
// of lanes sharing the backing store (LDS on the device: slot l of lane i lives at base[l * stride + i]).
struct Levels {
    double2* base;
    uint32_t stride;
};
RT_HD void lv_set(const Levels& L, uint32_t l, double s, double e) {
    double2 v; v.x = s; v.y = e;
    L.base[l * L.stride] = v;
}
RT_HD double2 lv_get(const Levels& L, uint32_t l) { return L.base[l * L.stride]; }

struct Best {
    double t;        // closest accepted t so far (inf when none)
    uint32_t ref;    // RT_NO_HIT when none
    uint32_t chain;
};
struct Tally {
    uint32_t node_tests, prim_tests, prim_bytes, cull_tests;
    uint32_t prim_flops;  // algorithmic: 35 per sphere test, 60 per quad / triangle, 70 per quadric, 30 per medium
    uint32_t wide_tests;  // child boxes of the eight-wide tree tested (f32, rt_wide.h); none in the reference
};
RT_HD Tally tally_zero() {
    Tally t;
    t.node_tests = 0; t.prim_tests = 0; t.prim_bytes = 0; t.cull_tests = 0; t.prim_flops = 0; t.wide_tests = 0;
    return t;
}

// True when every active lane of the wavefront holds the same x (returned in `first`).  Records addressed through
// `first` are then fetched once per wavefront with scalar loads instead of once per lane.
#if defined(__HIP_DEVICE_COMPILE__)
RT_HD uint32_t wave_first(uint32_t x) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)x); }
RT_HD bool wave_all(bool p) { return __all(p) != 0; }
RT_HD bool wave_uniform(uint32_t x, uint32_t& first) {
    first = (uint32_t)__builtin_amdgcn_readfirstlane((int)x);
    return __all(x == first) != 0;
}
#else
inline bool& host_uniform_mode() { static bool m = true; return m; }  // lets CPU tests take either branch
inline unsigned long long& host_fallback_count() { static unsigned long long n = 0; return n; }  // for the host tests
RT_HD uint32_t wave_first(uint32_t x) { return x; }
RT_HD bool wave_all(bool p) { return p; }
RT_HD bool wave_uniform(uint32_t x, uint32_t& first) { first = x; return host_uniform_mode(); }
#endif

// AABB::hit, aabb.rs:164-200: narrows (s, e) axis by axis, miss as soon as e <= s.
RT_HD bool slab3(const double2 bx, const double2 by, const double2 bz, const Ray& r, double& s, double& e);
RT_HD bool slab(const double2* b, const Ray& r, double& s, double& e) { return slab3(b[0], b[1], b[2], r, s, e); }
RT_HD bool slab3(const double2 bx, const double2 by, const double2 bz, const Ray& r, double& s, double& e) {
    {
        double t0 = (bx.x - r.o.x) * r.inv.x, t1 = (bx.y - r.o.x) * r.inv.x;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e <= s) return false;
    }
    {
        double t0 = (by.x - r.o.y) * r.inv.y, t1 = (by.y - r.o.y) * r.inv.y;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e <= s) return false;
    }
    {
        double t0 = (bz.x - r.o.z) * r.inv.z, t1 = (bz.y - r.o.z) * r.inv.z;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e <= s) return false;
    }
    return true;
}

// The same arithmetic as a conservative test (the cull boxes, which the reference does not have): the box is passed over
// only when e < s, NOT when e = s as in AABB::hit.  A quad tests a closed interval (parallelogram.rs:101-103), so with
// s = e -- a ray that enters its leaf's box at the very parameter of the closest hit so far, as at a corner shared by
// neighbouring boxes -- a hit at t = s = e is still possible, and by the reference's order a later quad takes such a tie.
RT_HD bool slab3_cull(const double2 bx, const double2 by, const double2 bz, const Ray& r, double s, double e) {
    {
        double t0 = (bx.x - r.o.x) * r.inv.x, t1 = (bx.y - r.o.x) * r.inv.x;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e < s) return false;
    }
    {
        double t0 = (by.x - r.o.y) * r.inv.y, t1 = (by.y - r.o.y) * r.inv.y;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e < s) return false;
    }
    {
        double t0 = (bz.x - r.o.z) * r.inv.z, t1 = (bz.y - r.o.z) * r.inv.z;
        if (t0 < t1) { s = fmax(s, t0); e = fmin(e, t1); } else { s = fmax(s, t1); e = fmin(e, t0); }
        if (e < s) return false;
    }
    return true;
}

// The same without the early exits (for steps a wavefront takes for lanes in different situations, where a lane's exit
// saves nothing): every axis is worked through, and the three tests AABB::hit makes on the way are combined at the end.
RT_HD bool slab3_flat(const double2 bx, const double2 by, const double2 bz, const Ray& r, double& s, double& e) {
    bool ok = true;
    {
        const double t0 = (bx.x - r.o.x) * r.inv.x, t1 = (bx.y - r.o.x) * r.inv.x;
        const bool up = t0 < t1;
        s = fmax(s, up ? t0 : t1); e = fmin(e, up ? t1 : t0);
        ok = ok & !(e <= s);
    }
    {
        const double t0 = (by.x - r.o.y) * r.inv.y, t1 = (by.y - r.o.y) * r.inv.y;
        const bool up = t0 < t1;
        s = fmax(s, up ? t0 : t1); e = fmin(e, up ? t1 : t0);
        ok = ok & !(e <= s);
    }
    {
        const double t0 = (bz.x - r.o.z) * r.inv.z, t1 = (bz.y - r.o.z) * r.inv.z;
        const bool up = t0 < t1;
        s = fmax(s, up ? t0 : t1); e = fmin(e, up ? t1 : t0);
        ok = ok & !(e <= s);
    }
    return ok;
}

// Sphere::hit, sphere.rs:40-63 (the part that decides t; the record is rebuilt in finalize()).
RT_HD bool sphere_t(const double2* sp, const Ray& r, double s, double e, double& t) {
    const double2 a0 = sp[0], a1 = sp[1];
    V3 oc = mk(a0.x, a0.y, a1.x) - r.o;
    double h = dot(oc, r.d);
    double c = len2(oc) - a1.y * a1.y;
    double disc = h * h - r.a * c;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double root = (h - sq) / r.a;
    if (!(s <= root && root < e)) {
        root = (h + sq) / r.a;
        if (!(s <= root && root < e)) return false;
    }
    t = root;
    return true;
}
// The 16-f64 record of a quad or triangle: n(3) d q(3) u(3) v(3) w(3).
struct Planar {
    V3 n, q, u, v, w;
    double d;
};
RT_HD Planar load_planar(const double2* p) {
    const double2 a0 = p[0], a1 = p[1], a2 = p[2], a3 = p[3], a4 = p[4], a5 = p[5], a6 = p[6], a7 = p[7];
    Planar r;
    r.n = mk(a0.x, a0.y, a1.x); r.d = a1.y;
    r.q = mk(a2.x, a2.y, a3.x);
    r.u = mk(a3.y, a4.x, a4.y);
    r.v = mk(a5.x, a5.y, a6.x);
    r.w = mk(a6.y, a7.x, a7.y);
    return r;
}
// Parallelogram::hit, parallelogram.rs:96-120: closed interval, tested before the parallel check.
RT_HD bool quad_t(const Planar& g, const Ray& r, double s, double e, double& t_out) {
    double den = dot(g.n, r.d);
    double t = (g.d - dot(g.n, r.o)) / den;
    if (e < t || t < s) return false;
    if (fabs(den) < 1e-8) return false;
    V3 ph = ray_at(r, t) - g.q;
    double alpha = dot(g.w, cross(ph, g.v));
    double beta = dot(g.w, cross(g.u, ph));
    if (!(0.0 <= alpha && alpha < 1.0 && 0.0 <= beta && beta < 1.0)) return false;  // :139-143
    t_out = t;
    return true;
}
// Parallelogram::hit without branches, for runs that a whole wavefront works on together (one lane's early exit
// saves the wavefront nothing there, and every branch costs exec-mask bookkeeping on the scalar unit): the same
// operations in the same order, the tests folded into one predicate.  A t that fails the interval or parallel test
// still flows through the remaining arithmetic; whatever it produces is discarded.
RT_HD bool quad_flat_t(const Planar& g, const Ray& r, double s, double e, double& t_out) {
    double den = dot(g.n, r.d);
    double t = (g.d - dot(g.n, r.o)) / den;
    bool ok = !(e < t) & !(t < s) & !(fabs(den) < 1e-8);
    V3 ph = ray_at(r, t) - g.q;
    double alpha = dot(g.w, cross(ph, g.v));
    double beta = dot(g.w, cross(g.u, ph));
    ok = ok & (0.0 <= alpha) & (alpha < 1.0) & (0.0 <= beta) & (beta < 1.0);
    t_out = t;
    return ok;
}
// The same for a quad whose normal lies along coordinate axis `ax` - 1 (n and w exactly zero in the other two
// components, as for every side of create_box and of the Cornell box): the terms of the dot products that multiply
// those zeros are left out.  For a tame ray (below) each of them is +-0, so den, t, alpha and beta come out with the
// same values as in quad_t (at most the sign of a zero differs, which no comparison here sees).
// (a: the normal's axis; b, c: the next two axes, cyclically.  Scalars by value and one call per axis, so that no
// field is ever picked by a run-time index.)
RT_HD bool quad_axis_core(double na, double wa, double gd, double oa, double da, double ob, double db, double qb,
                          double ub, double vb, double oc, double dc, double qc, double uc, double vc, double s,
                          double e, double& t_out) {
    double den = na * da;
    double t = (gd - na * oa) / den;
    bool ok = !(e < t) & !(t < s) & !(fabs(den) < 1e-8);
    double phb = (ob + t * db) - qb, phc = (oc + t * dc) - qc;
    double alpha = wa * (phb * vc - phc * vb);  // w . cross(ph, v)
    double beta = wa * (ub * phc - uc * phb);   // w . cross(u, ph)
    ok = ok & (0.0 <= alpha) & (alpha < 1.0) & (0.0 <= beta) & (beta < 1.0);
    t_out = t;
    return ok;
}
RT_HD bool quad_axis_t(const Planar& g, uint32_t ax, const Ray& r, double s, double e, double& t_out) {
    if (ax == 1u)
        return quad_axis_core(g.n.x, g.w.x, g.d, r.o.x, r.d.x, r.o.y, r.d.y, g.q.y, g.u.y, g.v.y, r.o.z, r.d.z, g.q.z,
                              g.u.z, g.v.z, s, e, t_out);
    if (ax == 2u)
        return quad_axis_core(g.n.y, g.w.y, g.d, r.o.y, r.d.y, r.o.z, r.d.z, g.q.z, g.u.z, g.v.z, r.o.x, r.d.x, g.q.x,
                              g.u.x, g.v.x, s, e, t_out);
    return quad_axis_core(g.n.z, g.w.z, g.d, r.o.z, r.d.z, r.o.x, r.d.x, g.q.x, g.u.x, g.v.x, r.o.y, r.d.y, g.q.y,
                          g.u.y, g.v.y, s, e, t_out);
}
// The shortcut above needs products with zero to be zero: no infinity or NaN in the ray, and magnitudes that
// cannot overflow on the way (record values are checked the same way where quad_axis is filled in).
RT_HD bool ray_is_tame(const Ray& r) {
    return fabs(r.o.x) + fabs(r.o.y) + fabs(r.o.z) + fabs(r.d.x) + fabs(r.d.y) + fabs(r.d.z) < 1e100;
}
// Triangle::hit, triangle.rs:123-144: parallel check first, half-open interval, strict barycentric test.
RT_HD bool tri_t(const Planar& g, const Ray& r, double s, double e, double& t_out) {
    double den = dot(g.n, r.d);
    if (fabs(den) < 1e-8) return false;
    double t = (g.d - dot(g.n, r.o)) / den;
    if (!(s <= t && t < e)) return false;
    V3 ph = ray_at(r, t) - g.q;
    double alpha = dot(g.w, cross(ph, g.v));
    double beta = dot(g.w, cross(g.u, ph));
    if (!(alpha > 0.0 && beta > 0.0 && alpha + beta < 1.0)) return false;  // :163-165
    t_out = t;
    return true;
}
// Both of the above in one instruction stream, for wavefronts whose lanes hold a mix of quads and triangles.  The
// two hit() bodies differ only in the order of the interval and parallel checks (either rejects, so the order does
// not show in the result), in the interval being closed or half-open, and in the interior test.
RT_HD bool planar_t(const Planar& g, bool is_tri, const Ray& r, double s, double e, double& t_out) {
    double den = dot(g.n, r.d);
    double t = (g.d - dot(g.n, r.o)) / den;
    bool in_iv = is_tri ? (s <= t && t < e) : !(e < t || t < s);
    if (fabs(den) < 1e-8 || !in_iv) return false;
    V3 ph = ray_at(r, t) - g.q;
    double alpha = dot(g.w, cross(ph, g.v));
    double beta = dot(g.w, cross(g.u, ph));
    bool inside = is_tri ? (alpha > 0.0 && beta > 0.0 && alpha + beta < 1.0)
                         : (0.0 <= alpha && alpha < 1.0 && 0.0 <= beta && beta < 1.0);
    if (!inside) return false;
    t_out = t;
    return true;
}
// planar_t without branches, for steps a wavefront takes for whichever of its lanes stand on a planar primitive (a
// lane's early exit saves the wavefront nothing there: see quad_flat_t).  The same operations in the same order; what a
// failed test leaves in the remaining arithmetic is discarded.
RT_HD bool planar_flat_t(const Planar& g, bool is_tri, const Ray& r, double s, double e, double& t_out) {
    double den = dot(g.n, r.d);
    double t = (g.d - dot(g.n, r.o)) / den;
    const bool in_iv = is_tri ? ((s <= t) & (t < e)) : (!(e < t) & !(t < s));
    V3 ph = ray_at(r, t) - g.q;
    double alpha = dot(g.w, cross(ph, g.v));
    double beta = dot(g.w, cross(g.u, ph));
    const bool inside = is_tri ? ((alpha > 0.0) & (beta > 0.0) & (alpha + beta < 1.0))
                               : ((0.0 <= alpha) & (alpha < 1.0) & (0.0 <= beta) & (beta < 1.0));
    t_out = t;
    return !(fabs(den) < 1e-8) & in_iv & inside;
}
// Quadric::hit, quadric.rs:26-54.
RT_HD V3 prod1(V3 u, V3 v) { return mk(u.x * v.y, u.x * v.z, u.y * v.z); }  // :91-97
RT_HD V3 anticross(V3 u, V3 v) {                                            // :99-105
    return mk(u.x * v.y + u.y * v.x, u.x * v.z + u.z * v.x, u.y * v.z + u.z * v.y);
}
RT_HD bool quadric_t(const double2* q, const Ray& r, double s, double e, double& t) {
    const double2 a0 = q[0], a1 = q[1], a2 = q[2], a3 = q[3], a4 = q[4];
    V3 p1 = mk(a0.x, a0.y, a1.x), p2 = mk(a1.y, a2.x, a2.y), p3 = mk(a3.x, a3.y, a4.x);
    V3 o = r.o, d = r.d;
    double a = dot(p1, d * d) + dot(p2, prod1(d, d));
    double h = -dot(p1, d * o) - 0.5 * dot(p2, anticross(d, o)) - 0.5 * dot(p3, d);
    double c = dot(p1, o * o) + dot(p2, anticross(o, o)) + dot(p3, o) + a4.y;
    double disc = h * h - a * c;
    if (disc < 0.0) return false;
    double sq = sqrt(disc);
    double root = (h - sq) / a;
    if (!(s <= root && root < e)) {
        root = (h + sq) / a;
        if (!(s <= root && root < e)) return false;
    }
    t = root;
    return true;
}

// One run of `count` primitives of one type, in list order, against the interval (s, e): HittableSlice::hit
// (hittable_list.rs:88-102) with closest_so_far kept in e.
// UNIFORM: every lane of the wavefront is on this same run (its records are wavefront-uniform); per-quad choices
// such as the axis-aligned shortcut are then uniform branches.
template <bool COUNT, bool UNIFORM = false>
RT_HD void run_prims(const DScene& S, uint32_t kind, uint32_t first, uint32_t count, uint32_t chain, const Ray& cur,
                     double s, double e, Best& best, Tally& tally) {
    const uint32_t last = first + count;
    if (kind == RT_NODE_TRIANGLES) {
        const double2* base = reinterpret_cast<const double2*>(S.tris);
        for (uint32_t k = first; k < last; ++k) {
            if (COUNT) { tally.prim_tests++; tally.prim_bytes += 136u; tally.prim_flops += 60u; }
            double t;
            if (tri_t(load_planar(base + 8u * k), cur, s, e, t)) {
                best.t = t; best.ref = make_ref(RT_NODE_TRIANGLES, k); best.chain = chain;
                e = t;  // t < e already
            }
        }
    } else if (kind == RT_NODE_QUADS) {
        const double2* base = reinterpret_cast<const double2*>(S.quads);
        if (UNIFORM) {
            // a wavefront whose rays are not all tame (never, in a render) takes the general form for all of them
            const bool tame = wave_all(ray_is_tame(cur));
            for (uint32_t k = first; k < last; ++k) {
                if (COUNT) { tally.prim_tests++; tally.prim_bytes += 136u; tally.prim_flops += 60u; }
                double t;
                const Planar g = load_planar(base + 8u * k);
                const uint32_t ax = S.quad_axis[k];
                const bool hit = (ax != 0u && tame) ? quad_axis_t(g, ax, cur, s, e, t) : quad_flat_t(g, cur, s, e, t);
                // t <= e already: a later quad at the same t replaces an earlier one
                best.t = hit ? t : best.t;
                best.ref = hit ? make_ref(RT_NODE_QUADS, k) : best.ref;
                best.chain = hit ? chain : best.chain;
                e = hit ? t : e;
            }
        } else {
            for (uint32_t k = first; k < last; ++k) {
                if (COUNT) { tally.prim_tests++; tally.prim_bytes += 136u; tally.prim_flops += 60u; }
                double t;
                if (quad_t(load_planar(base + 8u * k), cur, s, e, t)) {
                    best.t = t; best.ref = make_ref(RT_NODE_QUADS, k); best.chain = chain;
                    e = t;  // t <= e already: a later quad at the same t replaces an earlier one
                }
            }
        }
    } else if (kind == RT_NODE_SPHERES) {
        const double2* base = reinterpret_cast<const double2*>(S.spheres);
        for (uint32_t k = first; k < last; ++k) {
            if (COUNT) { tally.prim_tests++; tally.prim_bytes += 40u; tally.prim_flops += 35u; }
            double t;
            if (sphere_t(base + 2u * k, cur, s, e, t)) {
                best.t = t; best.ref = make_ref(RT_NODE_SPHERES, k); best.chain = chain;
                e = t;
            }
        }
    } else {  // RT_NODE_QUADRICS
        const double2* base = reinterpret_cast<const double2*>(S.quadrics);
        for (uint32_t k = first; k < last; ++k) {
            if (COUNT) { tally.prim_tests++; tally.prim_bytes += 88u; tally.prim_flops += 70u; }
            double t;
            if (quadric_t(base + 5u * k, cur, s, e, t)) {
                best.t = t; best.ref = make_ref(RT_NODE_QUADRICS, k); best.chain = chain;
                e = t;
            }
        }
    }
}

// Conservative cull test (no counterpart in the reference): false when the ray's interval (s, e) cannot touch the
// box stored `at` 16-byte units into entry i's record, which bounds every point the entry's subtree can report a
// hit at.  Nothing is narrowed; skipping a subtree that cannot produce a hit leaves the result unchanged.
template <bool COUNT>
RT_HD bool cull_pass(const DScene& S, uint32_t i, uint32_t at, Ray& cur, double s, double e, Tally& tally) {
    if (COUNT) tally.cull_tests++;
    ray_need_inv(cur);
    const double2* b = reinterpret_cast<const double2*>(S.nodes + 8u * i + at);
    return slab3_cull(b[0], b[1], b[2], cur, s, e);
}

// A ray's signature: bit b is clear when the LINE the ray lies on cannot touch the cull box of the tree's b-th
// signature entry, whatever interval the walk arrives there with.  The lockstep walk, where one
// wavefront visits every entry for all of its lanes, walks a copy of the tree without the entries no lane's signature
// has the bit of; the scheduler groups rays of equal signature into the same wavefronts so that this happens often.
RT_HD bool line_off_axis(double mi, double mj, double di, double dj, double hi, double hj) {
    // separating axis d x e_k, (i, j) the two other coordinates: |m x d|_k against the box's reach along it, with
    // slack far above the rounding of the products and far below the margin the cull boxes are widened by
    const double a = mi * dj, b = mj * di;
    return fabs(a - b) > (hi * fabs(dj) + hj * fabs(di)) * (1.0 + 1e-9) + 1e-9 * (fabs(a) + fabs(b));
}
// Whether the line through (o, d) -- with `half`, its part t >= 0 -- cannot touch `box6` (x, y, z: start, end) once
// taken through the affine map A (12 values: a row-major 3 x 3 matrix, then the translation).
RT_HD bool line_misses_box(const double* A, const double* box6, V3 o, V3 d, bool half) {
    const V3 co = mk(A[0] * o.x + A[1] * o.y + A[2] * o.z + A[9], A[3] * o.x + A[4] * o.y + A[5] * o.z + A[10],
                     A[6] * o.x + A[7] * o.y + A[8] * o.z + A[11]);
    const V3 cd = mk(A[0] * d.x + A[1] * d.y + A[2] * d.z, A[3] * d.x + A[4] * d.y + A[5] * d.z,
                     A[6] * d.x + A[7] * d.y + A[8] * d.z);
    // line against box without a division: a line misses a box exactly when one of the axes d x e_k separates
    // them (or, with d_k = 0, the box's own axis e_k does)
    const V3 h = mk(0.5 * (box6[1] - box6[0]), 0.5 * (box6[3] - box6[2]), 0.5 * (box6[5] - box6[4]));
    const V3 m = mk(co.x - 0.5 * (box6[0] + box6[1]), co.y - 0.5 * (box6[2] + box6[3]), co.z - 0.5 * (box6[4] + box6[5]));
    bool off = line_off_axis(m.y, m.z, cd.y, cd.z, h.y, h.z) || line_off_axis(m.z, m.x, cd.z, cd.x, h.z, h.x) ||
               line_off_axis(m.x, m.y, cd.x, cd.y, h.x, h.y) || (cd.x == 0.0 && fabs(m.x) > h.x * (1.0 + 1e-9)) ||
               (cd.y == 0.0 && fabs(m.y) > h.y * (1.0 + 1e-9)) || (cd.z == 0.0 && fabs(m.z) > h.z * (1.0 + 1e-9));
    if (half) {
        // the whole box behind the origin: (p - o) . d < 0 for every point p of the box (the largest value is taken
        // corner by corner), with slack far above the rounding of the products
        const double ax = (box6[0] - co.x) * cd.x, bx = (box6[1] - co.x) * cd.x;
        const double ay = (box6[2] - co.y) * cd.y, by = (box6[3] - co.y) * cd.y;
        const double az = (box6[4] - co.z) * cd.z, bz = (box6[5] - co.z) * cd.z;
        const double top = fmax(ax, bx) + fmax(ay, by) + fmax(az, bz);
        const double mag = fabs(ax) + fabs(bx) + fabs(ay) + fabs(by) + fabs(az) + fabs(bz);
        off = off || top < -1e-9 * mag;
    }
    // anything out of the ordinary (infinite box, NaN) compares false above: "may touch"
    return off;
}
RT_HD uint32_t ray_signature(const SigInfo& G, V3 o, V3 d) {
    uint32_t sig = 0xFFFFFFFFu;
#pragma unroll
    for (uint32_t b = 0; b < RT_MAX_SIG; ++b) {
        if (b >= G.n) break;
        if (line_misses_box(G.xf[b], G.box[b], o, d, false)) sig &= ~(1u << b);
    }
    return sig;
}
// Gates (trees with BVH entries; no counterpart in the reference, results unchanged): entry b describes the bounds of
// everything below one top-level BVH entry, in that entry's instance chain.  Bit b of a ray's gate mask is clear
// when the ray -- t >= 0, which every ray of a render satisfies (camera.rs:363 starts at 0.001) -- cannot reach them;
// the launch that walks that entry is then handed only the rays whose bit is set.
#define RT_MAX_GATES 12u
struct GateInfo {
    uint32_t n, reserved;
    double box[RT_MAX_GATES][6];
    double xf[RT_MAX_GATES][12];
};
RT_HD uint32_t ray_gates(const GateInfo& G, V3 o, V3 d) {
    uint32_t bits = 0u;
    for (uint32_t b = 0; b < G.n; ++b)
        if (!line_misses_box(G.xf[b], G.box[b], o, d, true)) bits |= 1u << b;
    return bits;
}
// Host side: entry b of the signature from tree entry `n` (a CULL entry), its box, and the scene's chains / ops.
// (S needs its chains and ops only, as host pointers.)
inline void chain_map(const DScene& S, uint32_t chain, double* xf12) {
    const V3 e[4] = {mk(0, 0, 0), mk(1, 0, 0), mk(0, 1, 0), mk(0, 0, 1)};
    V3 img[4];
    for (int k = 0; k < 4; ++k) img[k] = ray_in_chain(S, ray_world(e[k], mk(0, 0, 1)), chain).o;
    for (int k = 0; k < 3; ++k) {
        const V3 col = img[k + 1] - img[0];
        xf12[0 + k] = col.x; xf12[3 + k] = col.y; xf12[6 + k] = col.z;
    }
    xf12[9] = img[0].x; xf12[10] = img[0].y; xf12[11] = img[0].z;
}
inline void signature_entry(SigInfo& G, const DScene& S, uint32_t b, const rt_node& n, const double* box6) {
    for (int k = 0; k < 6; ++k) G.box[b][k] = box6[k];
    chain_map(S, n.chain, G.xf[b]);
}
// What a wavefront of the lockstep walk may pass over: bit b set when no lane's signature has bit b (`sig` of a lane
// without a ray: 0).  The same value in every lane.
RT_HD uint32_t sig_skip_mask(uint32_t sig) {
    uint32_t m = 0u;
    for (uint32_t b = 0; b < RT_MAX_SIG; ++b)
        if (wave_all(((sig >> b) & 1u) == 0u)) m |= 1u << b;
    return wave_first(m);
}
// Where the copy of the tree starts that a wavefront with that mask walks.
RT_HD uint32_t tree_without(const DScene& S, uint32_t skip_mask) {
    return (skip_mask & (S.n_variants - 1u)) * S.n_nodes;
}

// The same against the cull box of a BVH entry, six f32 (rounded outwards when the records were built) in the fifth
// and sixth 16-byte unit of the record.
template <bool COUNT>
RT_HD bool cull_pass_f32(const DScene& S, uint32_t i, Ray& cur, double s, double e, Tally& tally) {
    if (COUNT) tally.cull_tests++;
    ray_need_inv(cur);
    const float* f = reinterpret_cast<const float*>(S.nodes + 8u * i + 4u);
    double2 cx, cy, cz;
    cx.x = f[0]; cx.y = f[1]; cy.x = f[2]; cy.y = f[3]; cz.x = f[4]; cz.y = f[5];
    return slab3_cull(cx, cy, cz, cur, s, e);
}

// BVHNode::hit's box test (bvh.rs:31-37) for the entry `n` at index i: returns the index to go on with.
template <bool COUNT>
RT_HD uint32_t bvh_step(const DScene& S, const Node& n, uint32_t i, const Levels& L, const Best& best, Ray& cur,
                        Tally& tally) {
    ray_need_inv(cur);
    double2 iv = lv_get(L, n.level);
    double s = iv.x, e = fmin(iv.y, best.t);
    if (COUNT) tally.node_tests++;
    if (!cull_pass_f32<COUNT>(S, i, cur, s, e, tally)) return n.skip;
    if (slab(reinterpret_cast<const double2*>(S.nodes + 8u * i + 1u), cur, s, e)) {
        lv_set(L, n.level + 1u, s, e);
        return i + 1u;
    }
    return n.skip;
}

// MEDIA says which ConstantMedium entries the tree may hold: RT_MEDIA_NONE, RT_MEDIA_FLAT (every boundary is
// nothing but spheres, quads and triangles, so the nested walk below is never needed and costs the kernel no
// registers) or RT_MEDIA_ANY.
#define RT_MEDIA_NONE 0
#define RT_MEDIA_FLAT 1
#define RT_MEDIA_ANY 2
template <int MEDIA, bool COUNT, bool LOCKSTEP, class Src>
RT_HD void walk(const DScene& S, const Src& src, uint32_t i, uint32_t iend, const Levels& L, RngPos& rng, Best& best,
                Ray& cur, uint32_t& cur_chain, Tally& tally);

// ConstantMedium::hit's two boundary hits (constant_medium.rs:55-63): boundary.hit over (-inf, inf), then over
// (t1 + 0.0001, inf).  Returns false when either misses.
template <bool COUNT, bool LOCKSTEP, class Src>
RT_HD bool boundary_two_passes(const DScene& S, const Src& src, uint32_t i, const Node& n, const Levels& L, RngPos& rng,
                               Ray& cur, uint32_t& cur_chain, Tally& tally, double& t1, double& t2) {
    double lo = -RT_INF;
    bool ok = true;
    for (int pass = 0; pass < 2 && ok; ++pass) {
        Best b; b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0;
        lv_set(L, n.level + 1u, lo, RT_INF);
        walk<RT_MEDIA_NONE, COUNT, LOCKSTEP>(S, src, i + 1u, n.skip, L, rng, b, cur, cur_chain, tally);
        ok = b.ref != RT_NO_HIT;
        if (pass == 0) t1 = b.t; else t2 = b.t;
        lo = b.t + 0.0001;
    }
    return ok;
}

// The same two values from ONE pass over a boundary that is nothing but spheres, quads and triangles (no BVH, no
// quadric): each primitive's hit parameter does not depend on the interval it is tested against, only whether it is
// accepted does, so the first pass returns the smallest parameter m1 any primitive offers and the second the
// smallest one that is >= m1 + 0.0001 (every hit() accepts start <= t).  A sphere offers both roots: Sphere::hit
// falls back to the far root when the near one is outside the interval (sphere.rs:52-58).  While scanning, a value
// within 0.0001 above the minimum so far is set aside; if the minimum drops afterwards, such a value may have been
// the second hit after all, and `lost` tells the caller to do the two passes instead (two boundary hits closer
// than 0.0001: edges and corners).
struct TwoMin {
    double m1, m2;
    bool dirty, lost;
};
RT_HD void two_min_add(TwoMin& w, double t) {
    if (t < w.m1) {
        if (w.dirty) w.lost = true;
        if (w.m1 >= t + 0.0001) w.m2 = w.m1;  // the old minimum is at most the old second
        else w.dirty = true;
        w.m1 = t;
    } else if (t >= w.m1 + 0.0001) {
        w.m2 = fmin(w.m2, t);
    } else {
        w.dirty = true;
    }
}
// The same update written without branches, applied only where `on` holds (for the wavefront-uniform scan).
RT_HD void two_min_add_flat(TwoMin& w, double t, bool on) {
    const bool lower = on & (t < w.m1);
    const bool old_beyond = w.m1 >= t + 0.0001;
    const bool beyond = t >= w.m1 + 0.0001;
    const bool upper = on & !(t < w.m1);
    w.lost = w.lost | (lower & w.dirty);
    w.dirty = w.dirty | (lower & !old_beyond) | (upper & !beyond);
    const double m2_upper = fmin(w.m2, t);
    w.m2 = lower ? (old_beyond ? w.m1 : w.m2) : ((upper & beyond) ? m2_upper : w.m2);
    w.m1 = lower ? t : w.m1;
}
// SECOND = false: the scan described above.  SECOND = true: the second pass on its own, for the undecided cases --
// the smallest value any primitive offers that is >= lo (into t2); no bookkeeping, and no counting (the scan before
// it has counted both passes).
template <bool COUNT, bool LOCKSTEP, bool SECOND, class Src>
RT_HD bool boundary_scan(const DScene& S, const Src& src, uint32_t i, const Node& n, Ray& cur, uint32_t& cur_chain,
                         Tally& tally, double lo, double& t1, double& t2, bool& lost) {
    TwoMin w; w.m1 = RT_INF; w.m2 = RT_INF; w.dirty = false; w.lost = false;
    uint32_t tests = 0, bytes = 0, flops = 0;
    for (uint32_t j = i + 1u; j < n.skip;) {
        const Node r = load_node(S, LOCKSTEP ? wave_first(j) : j);
        if (LOCKSTEP && r.kind == RT_NODE_CULL) { j = j + 1u; continue; }  // (a medium pruned as a whole never gets here)
        if (r.chain != cur_chain) { cur = ray_in_chain(S, src.world(), r.chain); cur_chain = r.chain; }
        if (r.kind == RT_NODE_CULL) {
            j = cull_pass<COUNT && !SECOND>(S, j, 1u, cur, -RT_INF, RT_INF, tally) ? j + 1u : r.skip;
            continue;
        }
        const uint32_t last = r.first + r.count;
        if (r.kind == RT_NODE_SPHERES) {
            const double2* base = reinterpret_cast<const double2*>(S.spheres);
            for (uint32_t k = r.first; k < last; ++k) {
                const double2 a0 = base[2u * k], a1 = base[2u * k + 1u];
                V3 oc = mk(a0.x, a0.y, a1.x) - cur.o;
                double h = dot(oc, cur.d);
                double c = len2(oc) - a1.y * a1.y;
                double disc = h * h - cur.a * c;
                if (!(disc < 0.0)) {
                    double sq = sqrt(disc);
                    double r1 = (h - sq) / cur.a, r2 = (h + sq) / cur.a;
                    const bool ok1 = -RT_INF <= r1 && r1 < RT_INF, ok2 = -RT_INF <= r2 && r2 < RT_INF;
                    if (SECOND) {
                        if (ok1 && r1 >= lo) w.m2 = fmin(w.m2, r1);
                        if (ok2 && r2 >= lo) w.m2 = fmin(w.m2, r2);
                    } else {
                        if (ok1) two_min_add(w, r1);
                        if (ok2) two_min_add(w, r2);
                    }
                }
            }
            if (COUNT) { tests += r.count; bytes += 40u * r.count; flops += 35u * r.count; }
        } else {
            const bool is_tri = r.kind == RT_NODE_TRIANGLES;
            const double2* base = reinterpret_cast<const double2*>(is_tri ? S.tris : S.quads);
            if (LOCKSTEP && !is_tri) {
                const bool tame = wave_all(ray_is_tame(cur));
                for (uint32_t k = r.first; k < last; ++k) {
                    double t;
                    const Planar g = load_planar(base + 8u * k);
                    const uint32_t ax = S.quad_axis[k];
                    const bool hit = (ax != 0u && tame) ? quad_axis_t(g, ax, cur, -RT_INF, RT_INF, t)
                                                        : quad_flat_t(g, cur, -RT_INF, RT_INF, t);
                    if (SECOND) w.m2 = (hit & (t >= lo)) ? fmin(w.m2, t) : w.m2;
                    else two_min_add_flat(w, t, hit);
                }
            } else {
                for (uint32_t k = r.first; k < last; ++k) {
                    double t;
                    if (planar_t(load_planar(base + 8u * k), is_tri, cur, -RT_INF, RT_INF, t)) {
                        if (SECOND) { if (t >= lo) w.m2 = fmin(w.m2, t); }
                        else two_min_add(w, t);
                    }
                }
            }
            if (COUNT) { tests += r.count; bytes += 136u * r.count; flops += 60u * r.count; }
        }
        j = j + 1u;
    }
    if (SECOND) {
        t2 = w.m2;
        return w.m2 < RT_INF;
    }
    const bool first = w.m1 < RT_INF;
    if (COUNT) {  // what the two passes would have tested: the second runs only after a first hit
        tally.prim_tests += first ? 2u * tests : tests;
        tally.prim_bytes += first ? 2u * bytes : bytes;
        tally.prim_flops += first ? 2u * flops : flops;
    }
    if (first && w.m1 >= w.m1 + 0.0001) w.m2 = fmin(w.m2, w.m1);  // 0.0001 below the spacing of doubles at m1
    t1 = w.m1;
    t2 = w.m2;
    lost = w.lost;
    return first && w.m2 < RT_INF;
}
// Dispatch: entries of a MEDIUM record with count == 1 have a boundary the scan can handle (set where the device
// records are built).  ANY = false promises that every medium of the scene is of that kind.
template <bool COUNT, bool LOCKSTEP, bool ANY, class Src>
RT_HD bool boundary_hits(const DScene& S, const Src& src, uint32_t i, const Node& n, const Levels& L, RngPos& rng,
                         Ray& cur, uint32_t& cur_chain, Tally& tally, double& t1, double& t2) {
    if (!ANY || n.count == 1u) {
        bool lost = false;
        bool ok = boundary_scan<COUNT, LOCKSTEP, false>(S, src, i, n, cur, cur_chain, tally, 0.0, t1, t2, lost);
        if (!lost) return ok;
#if !defined(__HIP_DEVICE_COMPILE__)
        host_fallback_count()++;
#endif
        // undecided: now that the first hit is known, the second from a scan of its own
        return boundary_scan<false, LOCKSTEP, true>(S, src, i, n, cur, cur_chain, tally, t1 + 0.0001, t1, t2, lost);
    }
    if constexpr (ANY) return boundary_two_passes<COUNT, LOCKSTEP>(S, src, i, n, L, rng, cur, cur_chain, tally, t1, t2);
    return false;
}

// Walks entries [i, iend) of the flattened tree in order, which is the order of the reference's recursion:
// HittableSlice::hit in list order with the interval end shrunk to the closest hit so far
// (hittable_list.rs:88-102), BVHNode::hit narrowing the interval by its box and handing it to left, then right
// with end = left's t (bvh.rs:29-60).  The interval an entry sees is (start, min(end, best.t)) of its level: a
// minimum over nested narrowings does not depend on when each was taken, so this equals the value the recursion
// passes down.  MEDIA selects whether ConstantMedium entries may occur (not inside a medium's own boundary).
//
// Shape of the loop: every lane first runs through BVH entries until it stands on a primitive run (or a medium),
// then all lanes work on their runs together; lanes that stand on the same entry take the scalar-load branch.
//
// LOCKSTEP = true is for trees without BVH entries: every lane then visits the same entries in the same order, so
// the per-lane branches are dropped and all scene data comes through scalar loads.
// CULL entries are passed over there without testing: a lane that failed the box could not leave the common
// sequence anyway, and incoherent rays practically never fail it all together.
template <int MEDIA, bool COUNT, bool LOCKSTEP, class Src>
RT_HD void walk(const DScene& S, const Src& src, uint32_t i, uint32_t iend, const Levels& L, RngPos& rng, Best& best,
                Ray& cur, uint32_t& cur_chain, Tally& tally) {
    for (;;) {
        Node n;
        n.kind = RT_NODE_BVH; n.level = 0; n.chain = 0; n.skip = 0; n.first = 0; n.count = 0;
        bool have = false;
        while (i < iend) {
            uint32_t i0;
            if (LOCKSTEP || wave_uniform(i, i0)) {
                if (LOCKSTEP) i0 = wave_first(i);
                const Node nu = load_node(S, i0);
                if (nu.kind != RT_NODE_BVH) { n = nu; have = true; break; }
                if (nu.chain != cur_chain) { cur = ray_in_chain(S, src.world(), nu.chain); cur_chain = nu.chain; }
                i = bvh_step<COUNT>(S, nu, i0, L, best, cur, tally);
            } else {
                const Node nl = load_node(S, i);
                if (nl.kind != RT_NODE_BVH) { n = nl; have = true; break; }
                if (nl.chain != cur_chain) { cur = ray_in_chain(S, src.world(), nl.chain); cur_chain = nl.chain; }
                i = bvh_step<COUNT>(S, nl, i, L, best, cur, tally);
            }
        }
        if (!have) break;
        if (n.chain != cur_chain) { cur = ray_in_chain(S, src.world(), n.chain); cur_chain = n.chain; }
        const double2 iv = lv_get(L, n.level);
        const double s = iv.x, e = fmin(iv.y, best.t);
        if (n.kind == RT_NODE_CULL) {
            if (LOCKSTEP) i = n.count != 0u ? n.skip : i + 1u;
            else i = cull_pass<COUNT>(S, i, 1u, cur, s, e, tally) ? i + 1u : n.skip;
        } else if (n.kind == RT_NODE_MEDIUM) {
            if constexpr (MEDIA != RT_MEDIA_NONE) {
                // ConstantMedium::hit, constant_medium.rs:46-93
                if (COUNT) { tally.prim_tests++; tally.prim_bytes += 16u; tally.prim_flops += 30u; }
                const rt_medium m = S.media[n.first];
                const double ray_length = sqrt(cur.a);  // of the ray the medium itself receives
                double t1 = 0.0, t2 = 0.0;
                const uint32_t i0 = LOCKSTEP ? wave_first(i) : i;
                if (boundary_hits<COUNT, LOCKSTEP, MEDIA == RT_MEDIA_ANY>(S, src, i0, n, L, rng, cur, cur_chain, tally, t1, t2)) {
                    t1 = fmax(t1, s);
                    t2 = fmin(t2, e);
                    t1 = fmax(t1, 0.0);
                    double inside = (t2 - t1) * ray_length;
                    double x = rng_draw_at(rng.k0, rng.k1, rng.pixel, rng.sample, rng.ctr) * (1.0 - 0.0) + 0.0;
                    rng.ctr++;
                    double hit_distance = m.neg_inv_density * log(x);
                    if (!(hit_distance > inside)) {
                        best.t = t1 + hit_distance / ray_length;
                        best.ref = make_ref(RT_NODE_MEDIUM, n.first);
                        best.chain = n.chain;
                    }
                }
            }
            i = n.skip;
        } else {
            uint32_t i0;
            if (LOCKSTEP || wave_uniform(i, i0)) {
                if (LOCKSTEP) i0 = wave_first(i);
                // all lanes stand on the same run (top-level lists, shared leaves): the entry and its records are
                // addressed by a wavefront-uniform index and fetched with scalar loads
                const Node nu = load_node(S, i0);
                run_prims<COUNT, true>(S, nu.kind, nu.first, nu.count, nu.chain, cur, s, e, best, tally);
            } else {
                run_prims<COUNT>(S, n.kind, n.first, n.count, n.chain, cur, s, e, best, tally);
            }
            i = i + 1u;
        }
    }
}

// world.hit(ray, tmin..tmax) (camera.rs:363): returns the closest hit as (t, ref, chain).
template <bool COUNT, bool LOCKSTEP, int MEDIA = RT_MEDIA_ANY, class Src>
RT_HD Best closest_hit(const DScene& S, const Src& src, double tmin, double tmax, const Levels& L, RngPos& rng,
                       Tally& tally, uint32_t first = 0u) {
    Best best; best.t = RT_INF; best.ref = RT_NO_HIT; best.chain = 0;
    lv_set(L, 0u, tmin, tmax);
    Ray cur = src.world();
    uint32_t cur_chain = 0;
    walk<MEDIA, COUNT, LOCKSTEP>(S, src, first, first + S.n_nodes, L, rng, best, cur, cur_chain, tally);
    return best;
}
// The same over entries [first, end) of the top level only, going on from `best` (what the entries before `first`
// have left): the reference's list scan (hittable_list.rs:88-102) cut into stretches.
template <bool COUNT, bool LOCKSTEP, int MEDIA = RT_MEDIA_ANY, class Src>
RT_HD Best closest_hit_range(const DScene& S, const Src& src, double tmin, double tmax, const Levels& L, RngPos& rng,
                             Tally& tally, uint32_t first, uint32_t end, Best best) {
    lv_set(L, 0u, tmin, tmax);
    Ray cur = src.world();
    uint32_t cur_chain = 0;
    walk<MEDIA, COUNT, LOCKSTEP>(S, src, first, end, L, rng, best, cur, cur_chain, tally);
    return best;
}

// The same walk as a per-lane state machine that does one BVH entry and one primitive test per step, for the
// kernel that keeps every lane busy by handing it a new ray as soon as its ray is finished.
#define RT_ST_NODE 0u
#define RT_ST_RUN 1u
#define RT_ST_DONE 2u
#define RT_ST_MEDIUM 3u  // standing on a MEDIUM entry, waiting for lane_medium_step
#define RT_ST_CHAIN 4u   // standing on an entry of another instance chain, waiting for lane_chain_step
#define RT_ST_SUB 5u     // about to start a leaf's run that has cull boxes per pair of primitives, waiting for lane_sub_step
struct Lane {
    Ray cur;
    uint32_t cur_chain;
    uint32_t i;              // next entry of the flattened tree
    uint32_t k, kend, kind;  // primitive run in progress (RT_ST_RUN); kind: bits 0..7 the kind; bit 8 set: bits 16..23
                             // say which of the pairs k, k + 2, ... are worth testing (bit 16 = the pair at k);
                             // RT_ST_SUB: bits 8.. the leaf's index into leaf_sub
    double s, e;             // interval the run in progress is tested against
    Best best;
    uint32_t state;
};
template <class Src>
RT_HD void lane_begin(Lane& ln, const Src& src, double tmin, double tmax, const Levels& L, uint32_t first = 0u) {
    ln.cur = src.world();
    ray_need_inv(ln.cur);
    ln.cur_chain = 0;
    ln.i = first;
    ln.k = ln.kend = 0; ln.kind = 0;
    ln.s = tmin; ln.e = tmax;
    ln.best.t = RT_INF; ln.best.ref = RT_NO_HIT; ln.best.chain = 0;
    ln.state = RT_ST_NODE;
    lv_set(L, 0u, tmin, tmax);
}
// One entry of the tree for a lane in RT_ST_NODE.
// MEDIA = RT_MEDIA_NONE is for scenes without a ConstantMedium: the medium code, its random draw and its logarithm then
// cost the kernel no registers.
template <bool COUNT, int MEDIA = RT_MEDIA_ANY, class Src>
RT_HD void lane_node_step(const DScene& S, const Src& src, Lane& ln, const Levels& L, RngPos& rng, Tally& tally) {
    if (ln.i >= S.n_nodes) { ln.state = RT_ST_DONE; return; }
    // Everything the record holds at once, whatever its kind (96 bytes, six loads): the boxes sit in the same cache
    // line as the header, which also names a leaf's run of primitives, and fetching them together costs one round
    // trip instead of one per dependent step.
    const uint4* rp = S.nodes + 8u * ln.i;
    const double2* bp = reinterpret_cast<const double2*>(rp);
    const uint4 raw = rp[0];
    const double2 b0 = bp[1], b1 = bp[2], b2 = bp[3];
    const float4 cf0 = reinterpret_cast<const float4*>(rp)[4];
    const uint4 tail = rp[5];  // two more f32 of that box, then (CULL entries) the next sibling's skip and flag
    float2 cf1;
    cf1.x = __builtin_bit_cast(float, tail.x); cf1.y = __builtin_bit_cast(float, tail.y);
    const double2 iv0 = lv_get(L, raw.x >> 8 & 0xFFu);
#if defined(__HIP_DEVICE_COMPILE__)
    // keeps the compiler from sinking the loads into the branches below, where each would wait on its own
    asm volatile("" ::"v"(b0.x), "v"(b1.x), "v"(b2.x), "v"(cf0.x), "v"(tail.x));
#endif
    Node n;
    n.kind = raw.x & 0xFFu; n.level = raw.x >> 8 & 0xFFu; n.chain = raw.x >> 16;
    n.skip = raw.y; n.first = raw.z; n.count = raw.w;
    if (n.chain != ln.cur_chain) {
        // re-reading the world ray, running it through the chain's transforms and taking the reciprocal of its new
        // direction is a detour the whole wavefront would take for one lane: collected like the media
        ln.state = RT_ST_CHAIN;
        return;
    }
    const double s = iv0.x, e = fmin(iv0.y, ln.best.t);
    if (n.kind == RT_NODE_BVH) {
        double bs = s, be = e;
        if (COUNT) { tally.node_tests++; tally.cull_tests++; }
        {
            double2 c0, c1, c2;  // (the cull box narrows nothing)
            c0.x = cf0.x; c0.y = cf0.y; c1.x = cf0.z; c1.y = cf0.w; c2.x = cf1.x; c2.y = cf1.y;
            if (!slab3_cull(c0, c1, c2, ln.cur, s, e)) { ln.i = n.skip; return; }
        }
        if (!slab3(b0, b1, b2, ln.cur, bs, be)) { ln.i = n.skip; return; }
        if (n.count != 0u) {
            // a leaf whose children are one run of primitives (the usual case): go straight to the run with the
            // narrowed interval in registers, without a round trip through the level slots
            ln.kind = n.count >> 24; ln.k = n.first; ln.kend = n.first + (n.count & 0xFFFFFFu);
            ln.s = bs; ln.e = be;
            ln.i = ln.i + 1u;
            ln.state = RT_ST_RUN;
            if (tail.z != 0u) { ln.kind |= (tail.z - 1u) << 8; ln.state = RT_ST_SUB; }
            return;
        }
        lv_set(L, n.level + 1u, bs, be);
        ln.i = ln.i + 1u;
        return;
    }
    if (n.kind == RT_NODE_CULL) {
        if (COUNT) tally.cull_tests++;
        if (slab3_cull(b0, b1, b2, ln.cur, s, e)) { ln.i = ln.i + 1u; return; }
        ln.i = n.skip;
        if (tail.w != 0u) {
            // a row of boxes: the record also holds the box of the CULL entry at n.skip (widened to f32), so a miss
            // here decides about that one in the same step -- straight into its subtree, or past it
            if (COUNT) tally.cull_tests++;
            double2 c0, c1, c2;
            c0.x = cf0.x; c0.y = cf0.y; c1.x = cf0.z; c1.y = cf0.w; c2.x = cf1.x; c2.y = cf1.y;
            ln.i = slab3_cull(c0, c1, c2, ln.cur, s, e) ? n.skip + 1u : tail.z;
        }
        return;
    }
    if (MEDIA != RT_MEDIA_NONE && n.kind == RT_NODE_MEDIUM) {
        // A medium costs several times what any other entry does (two boundary hits, a random draw, a logarithm),
        // and the wavefront pays for it whenever a single lane gets here: the lane waits in RT_ST_MEDIUM until the
        // caller has collected enough of them for one lane_medium_step.
        ln.state = RT_ST_MEDIUM;
        return;
    }
    if (n.count == 0u) { ln.i = ln.i + 1u; return; }
    ln.kind = n.kind; ln.k = n.first; ln.kend = n.first + n.count;
    ln.s = s; ln.e = e;
    ln.state = RT_ST_RUN;
}
// Takes a lane in RT_ST_CHAIN into the ray space of the entry it stands on (Translate::hit / RotateY::hit,
// hittable/mod.rs:78-91,118-135), reciprocal direction included (Ray::new, ray.rs:14-19): the state machine keeps
// it ready at all times, so that no box test has to stop for three divisions.
template <class Src>
RT_HD void lane_chain_step(const DScene& S, const Src& src, Lane& ln) {
    const Node n = load_node(S, ln.i);
    ln.cur = ray_in_chain(S, src.world(), n.chain);
    ray_need_inv(ln.cur);
    ln.cur_chain = n.chain;
    ln.state = RT_ST_NODE;
}
// The MEDIUM entry a lane in RT_ST_MEDIUM stands on: ConstantMedium::hit, constant_medium.rs:46-93 (same steps as in
// walk()).
template <bool COUNT, bool ANY = true, class Src>
RT_HD void lane_medium_step(const DScene& S, const Src& src, Lane& ln, const Levels& L, RngPos& rng, Tally& tally) {
    const Node n = load_node(S, ln.i);
    const double2 iv = lv_get(L, n.level);
    const double s = iv.x, e = fmin(iv.y, ln.best.t);
    if (COUNT) { tally.prim_tests++; tally.prim_bytes += 16u; tally.prim_flops += 30u; }
    const rt_medium m = S.media[n.first];
    const double ray_length = sqrt(ln.cur.a);  // lane_node_step has put the ray into the entry's chain
    double t1 = 0.0, t2 = 0.0;
    if (boundary_hits<COUNT, false, ANY>(S, src, ln.i, n, L, rng, ln.cur, ln.cur_chain, tally, t1, t2)) {
        t1 = fmax(t1, s);
        t2 = fmin(t2, e);
        t1 = fmax(t1, 0.0);
        double inside = (t2 - t1) * ray_length;
        double x = rng_draw_at(rng.k0, rng.k1, rng.pixel, rng.sample, rng.ctr) * (1.0 - 0.0) + 0.0;
        rng.ctr++;
        double hit_distance = m.neg_inv_density * log(x);
        if (!(hit_distance > inside)) {
            ln.best.t = t1 + hit_distance / ray_length;
            ln.best.ref = make_ref(RT_NODE_MEDIUM, n.first);
            ln.best.chain = n.chain;
        }
    }
    ray_need_inv(ln.cur);  // the boundary may have left the lane in another chain, whose next entry may be a box
    ln.i = n.skip;
    ln.state = RT_ST_NODE;
}
// The next primitive(s) of the run in progress for a lane in RT_ST_RUN: two per step where the run has two left, so
// that both records are in flight together and the step's fixed costs are shared.
// Moves a lane whose run has a pair mask (bit 0 of `mask` = the pair at ln.k) on to the next pair worth testing.
template <class LaneT>
RT_HD void lane_skip_pairs(LaneT& ln, uint32_t mask) {
    mask &= 0xFFu;
    while (ln.k < ln.kend && (mask & 1u) == 0u) { ln.k += 2u; mask >>= 1; }
    ln.kind = (ln.kind & 0xFFFFu) | mask << 16;
}
// (The tests themselves, for any lane type with the run fields of Lane; the caller moves the lane on once k >= kend.)
// ONLY != 0 promises that every run the lane can stand on holds primitives of that one kind (a tree of triangles only,
// of spheres only, ...): the code of the other kinds is then not compiled in.
template <bool COUNT, int ONLY = 0, class LaneT>
RT_HD void lane_prim_tests(const DScene& S, LaneT& ln, Tally& tally) {
    const bool two = ln.k + 1u < ln.kend;
    const uint32_t kind = ONLY != 0 ? (uint32_t)ONLY : (ln.kind & 0xFFu);
    if (kind == RT_NODE_TRIANGLES || kind == RT_NODE_QUADS) {
        const bool is_tri = kind == RT_NODE_TRIANGLES;
        const double2* rec = reinterpret_cast<const double2*>(is_tri ? S.tris : S.quads) + 8u * ln.k;
        const Planar g0 = load_planar(rec);
        if (COUNT) { tally.prim_tests += two ? 2u : 1u; tally.prim_bytes += two ? 272u : 136u; tally.prim_flops += two ? 120u : 60u; }
        double t0, t1 = 0.0;
        // Triangles without branches (a mesh's rays fail each of a triangle's checks at different points, and a lane's
        // early exit saves the wavefront nothing); quads with them: on the sides of a box most lanes fail the first check,
        // the interval, together (measured: final_scene - 3 % of its walk with, bust - 21 % without).
        bool h1 = false;
        if (is_tri) {
            // (a run of one -- every item of a mesh's wide tree -- reads and tests one record, not the same one twice)
            const bool h0 = planar_flat_t(g0, true, ln.cur, ln.s, ln.e, t0);
            ln.e = h0 ? t0 : ln.e;
            if (two) {
                const Planar g1 = load_planar(rec + 8u);
                h1 = planar_flat_t(g1, true, ln.cur, ln.s, ln.e, t1);
            }
            ln.e = h1 ? t1 : ln.e;
            ln.best.t = h1 ? t1 : (h0 ? t0 : ln.best.t);
            ln.best.ref = h1 ? make_ref(kind, ln.k + 1u) : (h0 ? make_ref(kind, ln.k) : ln.best.ref);
            ln.best.chain = (h0 | h1) ? ln.cur_chain : ln.best.chain;
            ln.k = ln.k + (two ? 2u : 1u);
            if (ln.kind & 0x100u) lane_skip_pairs(ln, (ln.kind >> 16) >> 1);
            return;
        }
        const Planar g1 = load_planar(rec + (two ? 8u : 0u));
        const bool h0 = planar_t(g0, false, ln.cur, ln.s, ln.e, t0);
        ln.e = h0 ? t0 : ln.e;  // (the second primitive is tested against the interval the first has left)
        h1 = two && planar_t(g1, false, ln.cur, ln.s, ln.e, t1);
        ln.e = h1 ? t1 : ln.e;
        ln.best.t = h1 ? t1 : (h0 ? t0 : ln.best.t);
        ln.best.ref = h1 ? make_ref(kind, ln.k + 1u) : (h0 ? make_ref(kind, ln.k) : ln.best.ref);
        ln.best.chain = (h0 | h1) ? ln.cur_chain : ln.best.chain;
    } else if (kind == RT_NODE_SPHERES) {
        // up to four spheres per step: their records (32 bytes each) share one 128-byte line
        const double2* rec = reinterpret_cast<const double2*>(S.spheres) + 2u * ln.k;
        const uint32_t left = ln.kend - ln.k, done = left < 4u ? left : 4u;
        double2 q[8];
#pragma unroll
        for (uint32_t j = 0; j < 8u; ++j) q[j] = rec[j < 2u * done ? j : 0u];
        if (COUNT) { tally.prim_tests += done; tally.prim_bytes += 40u * done; tally.prim_flops += 35u * done; }
#pragma unroll
        for (uint32_t j = 0; j < 4u; ++j) {
            double t;
            // (with its branches: most rays miss most spheres of a run, and the square root and the two divisions behind
            // the discriminant's sign are what a sphere costs)
            if (j < done && sphere_t(q + 2u * j, ln.cur, ln.s, ln.e, t)) {
                ln.best.t = t; ln.best.ref = make_ref(RT_NODE_SPHERES, ln.k + j); ln.best.chain = ln.cur_chain;
                ln.e = t;
            }
        }
        ln.k = ln.k + done;
        return;
    } else {
        double t;
        if (COUNT) { tally.prim_tests++; tally.prim_bytes += 88u; tally.prim_flops += 70u; }
        if (quadric_t(reinterpret_cast<const double2*>(S.quadrics) + 5u * ln.k, ln.cur, ln.s, ln.e, t)) {
            ln.best.t = t; ln.best.ref = make_ref(RT_NODE_QUADRICS, ln.k); ln.best.chain = ln.cur_chain;
            ln.e = t;
        }
        ln.k = ln.k + 1u;
        return;
    }
    ln.k = ln.k + (two ? 2u : 1u);
    if (ln.kind & 0x100u) lane_skip_pairs(ln, (ln.kind >> 16) >> 1);
}
template <bool COUNT>
RT_HD void lane_prim_step(const DScene& S, Lane& ln, Tally& tally) {
    lane_prim_tests<COUNT>(S, ln, tally);
    if (ln.k >= ln.kend) { ln.state = RT_ST_NODE; ln.i = ln.i + 1u; }
}
// A leaf's run with cull boxes per pair of primitives (no counterpart in the reference; results unchanged: a pair is
// passed over only when the ray's interval cannot touch the box around its corners).  One step tests all eight
// boxes and moves the lane to the first pair worth testing, or past the run.
template <bool COUNT>
RT_HD void lane_sub_step(const DScene& S, Lane& ln, Tally& tally) {
    const float4* bp = reinterpret_cast<const float4*>(S.leaf_sub + 48u * (ln.kind >> 8));
    float4 f[12];
#pragma unroll
    for (int j = 0; j < 12; ++j) f[j] = bp[j];
    if (COUNT) tally.cull_tests += 8u;
    const double e = fmin(ln.e, ln.best.t);
    uint32_t mask = 0u;
#pragma unroll
    for (int b = 0; b < 8; b += 2) {  // two boxes in three float4
        const float4 p = f[3 * (b / 2)], q = f[3 * (b / 2) + 1], r = f[3 * (b / 2) + 2];
        double2 x0, y0, z0, x1, y1, z1;
        x0.x = p.x; x0.y = p.y; y0.x = p.z; y0.y = p.w; z0.x = q.x; z0.y = q.y;
        x1.x = q.z; x1.y = q.w; y1.x = r.x; y1.y = r.y; z1.x = r.z; z1.y = r.w;
        if (slab3_cull(x0, y0, z0, ln.cur, ln.s, e)) mask |= 1u << b;
        if (slab3_cull(x1, y1, z1, ln.cur, ln.s, e)) mask |= 2u << b;
    }
    ln.kind = (ln.kind & 0xFFu) | 0x100u;
    ln.state = RT_ST_RUN;
    lane_skip_pairs(ln, mask);
    if (ln.k >= ln.kend) { ln.state = RT_ST_NODE; ln.i = ln.i + 1u; }
}

// ------------------------------------------------------------------------------------------------------------
struct Rec {  // hittable/mod.rs:18-31
    V3 p, n;
    double u, v;
    uint32_t mat, prim_id;
    bool front;
};
RT_HD void face_normal(Rec& r, V3 dir, V3 outward) {  // hittable/mod.rs:51-58
    r.front = dot(outward, dir) < 0.0;
    r.n = r.front ? outward : -outward;
}
RT_HD void sphere_uv(V3 p, double& u, double& v) {  // sphere.rs:90-98
    double theta = acos(-p.y);
    double phi = atan2(-p.z, p.x) + RT_PI;
    u = phi / (2.0 * RT_PI);
    v = theta / RT_PI;
}
// Rebuilds the HitRecord of the closest hit exactly as the primitive's hit() fills it (sphere.rs:65-73,
// parallelogram.rs:113-128, triangle.rs:137-152, quadric.rs:56-82, constant_medium.rs:79-90) in the ray space of
// its chain, then maps p and normal back out through the instances (hittable/mod.rs:88,131-132).
// FULL = false skips (u, v) when the material cannot read them: only an image texture does, and a checker may lead
// to one.
RT_HD bool mat_reads_uv(const DScene& S, uint32_t mat) {
    const rt_material m = S.mats[mat];
    if (m.kind == RT_MAT_METAL || m.kind == RT_MAT_DIELECTRIC) return false;
    const uint32_t k = S.texs[m.texture].kind;
    return k == RT_TEX_IMAGE || k == RT_TEX_CHECKER;
}
// PLAIN = true is for scenes whose materials are all Lambertian, DiffuseLight or Isotropic over SolidColor textures
// (checked where the scene is committed): nothing reads (u, v) then.
template <bool FULL, bool PLAIN = false>
RT_HD Rec finalize(const DScene& S, const Ray& wray, const Best& b) {
    Rec r;
    Ray ray = ray_in_chain(S, wray, b.chain);
    uint32_t kind = b.ref >> RT_REF_KIND_SHIFT, k = b.ref & RT_REF_INDEX_MASK;
    r.u = 0.0; r.v = 0.0;
    r.p = ray_at(ray, b.t);
    switch (kind) {
        case RT_NODE_SPHERES: {
            const double* sp = S.spheres + 4u * k;
            V3 outward = (r.p - mk(sp[0], sp[1], sp[2])) / sp[3];
            face_normal(r, ray.d, outward);
            r.mat = S.sphere_mat[k]; r.prim_id = S.sphere_id[k];
            if (FULL || (!PLAIN && mat_reads_uv(S, r.mat))) sphere_uv(outward, r.u, r.v);
            break;
        }
        case RT_NODE_QUADS:
        case RT_NODE_TRIANGLES: {
            const double* q = (kind == RT_NODE_QUADS ? S.quads : S.tris) + 16u * k;
            if (kind == RT_NODE_QUADS) { r.mat = S.quad_mat[k]; r.prim_id = S.quad_id[k]; }
            else { r.mat = S.tri_mat[k]; r.prim_id = S.tri_id[k]; }
            if (FULL || (!PLAIN && mat_reads_uv(S, r.mat))) {
                V3 ph = r.p - ld3(q + 4);
                r.u = dot(ld3(q + 13), cross(ph, ld3(q + 10)));
                r.v = dot(ld3(q + 13), cross(ld3(q + 7), ph));
            }
            face_normal(r, ray.d, ld3(q));
            break;
        }
        case RT_NODE_QUADRICS: {
            const double* q = S.quadrics + 10u * k;
            V3 p1 = ld3(q), p2 = ld3(q + 3), p3 = ld3(q + 6), p = r.p;
            V3 g = mk(2.0 * p1.x * p.x + p2.x * p.y + p2.y * p.z, 2.0 * p1.y * p.y + p2.x * p.x + p2.z * p.z,
                      2.0 * p1.z * p.z + p2.y * p.x + p2.z * p.y) + p3;
            face_normal(r, ray.d, unit(g));
            r.mat = S.quadric_mat[k]; r.prim_id = S.quadric_id[k];
            break;
        }
        default: {  // RT_NODE_MEDIUM
            r.n = mk(1.0, 0.0, 0.0);
            r.front = true;
            r.mat = S.media[k].material; r.prim_id = S.media[k].prim_id;
            break;
        }
    }
    if (b.chain != 0) {
        rt_chain c = S.chains[b.chain];
        for (uint32_t j = c.n_ops; j > 0; --j) {
            const rt_xform_op& op = S.ops[c.first_op + j - 1u];
            if (op.kind == RT_XF_TRANSLATE) {
                r.p = r.p + mk(op.v[0], op.v[1], op.v[2]);
            } else {
                r.p = rotate_y(r.p, op.v[1], -op.v[0]);
                r.n = rotate_y(r.n, op.v[1], -op.v[0]);
            }
        }
    }
    return r;
}

// ------------------------------------------------------------------------------------------------------------
// perlin.rs:46-143
RT_HD double perlin_noise(const double* vec, const uint32_t* perm, V3 p) {
    double fx = floor(p.x), fy = floor(p.y), fz = floor(p.z);
    double u = p.x - fx, v = p.y - fy, w = p.z - fz;
    long long i = (long long)fx, j = (long long)fy, k = (long long)fz;
    double uu = u * u * (3.0 - 2.0 * u);
    double vv = v * v * (3.0 - 2.0 * v);
    double ww = w * w * (3.0 - 2.0 * w);
    double accum = 0.0;
    for (int di = 0; di < 2; ++di)
        for (int dj = 0; dj < 2; ++dj)
            for (int dk = 0; dk < 2; ++dk) {
                uint32_t fi = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
                V3 c = ld3(vec + 3u * fi);
                double a = (double)di, b = (double)dj, cc = (double)dk;
                V3 wv = mk(u - a, v - b, w - cc);
                accum += (a * uu + (1.0 - a) * (1.0 - uu)) * (b * vv + (1.0 - b) * (1.0 - vv)) *
                         (cc * ww + (1.0 - cc) * (1.0 - ww)) * dot(c, wv);
            }
    return accum;
}
RT_HD double perlin_turbulence(const double* vec, const uint32_t* perm, V3 p, int depth) {  // :79-92
    double accum = 0.0, weight = 1.0;
    V3 tp = p;
    for (int i = 0; i < depth; ++i) {
        accum += weight * perlin_noise(vec, perm, tp);
        weight *= 0.5;
        tp = 2.0 * tp;
    }
    return fabs(accum);
}
// CheckerTexture::value's choice of child (texture.rs:59-80), followed down to a texture that is not a checker.
RT_HD uint32_t tex_resolve(const DScene& S, uint32_t tid, V3 p) {
    for (int guard = 0; guard < 64; ++guard) {
        if (S.texs[tid].kind != RT_TEX_CHECKER) return tid;
        const double sc = S.texs[tid].scale;
        long long xi = (long long)floor(sc * p.x);
        long long yi = (long long)floor(sc * p.y);
        long long zi = (long long)floor(sc * p.z);
        tid = ((xi + yi + zi) % 2 == 0) ? S.texs[tid].a : S.texs[tid].b;
    }
    return tid;
}
// Texture::value, texture.rs:25-29 (solid), :59-80 (checker), :116-136 (image), :143-157 (perlin).
// PERLIN = false: the caller has made sure that `tid` is a solid colour or an image (tex_resolve).
template <bool PERLIN = true>
RT_HD V3 tex_value(const DScene& S, uint32_t tid, double u, double v, V3 p) {
    for (int guard = 0; guard < 64; ++guard) {
        const rt_texture t = S.texs[tid];
        if (t.kind == RT_TEX_SOLID) return mk(t.color[0], t.color[1], t.color[2]);
        if (t.kind == RT_TEX_CHECKER) {
            long long xi = (long long)floor(t.scale * p.x);
            long long yi = (long long)floor(t.scale * p.y);
            long long zi = (long long)floor(t.scale * p.z);
            tid = ((xi + yi + zi) % 2 == 0) ? t.a : t.b;
            continue;
        }
        if (t.kind == RT_TEX_IMAGE) {
            uint32_t off = S.image_desc[3u * t.a], w = S.image_desc[3u * t.a + 1u], h = S.image_desc[3u * t.a + 2u];
            double cu = u < 0.0 ? 0.0 : (u > 1.0 ? 1.0 : u);
            double cv0 = v < 0.0 ? 0.0 : (v > 1.0 ? 1.0 : v);
            double cv = 1.0 - cv0;
            uint32_t ui = (uint32_t)((double)w * cu), vi = (uint32_t)((double)h * cv);
            if (ui >= w) ui = w - 1u;  // texture.rs:124-127 would index one past the edge at u == 1 / v == 0
            if (vi >= h) vi = h - 1u;
            const uint8_t* px = S.texels + 3u * ((size_t)off + (size_t)vi * w + ui);
            double sc = 1.0 / 255.0;
            return mk((double)px[0] * sc, (double)px[1] * sc, (double)px[2] * sc);
        }
        // RT_TEX_PERLIN
        if (!PERLIN) return mk(0, 0, 0);
        const double* vec = S.perlin_vec + 768u * t.a;
        const uint32_t* perm = S.perlin_perm + 768u * t.a;
        return mk(0.5, 0.5, 0.5) * (1.0 + sin(t.scale * p.z + 10.0 * perlin_turbulence(vec, perm, p, 7)));
    }
    return mk(0, 0, 0);
}

// Material::emitted + Material::scatter (material.rs).  Returns false when the path ends here; `emit` is the
// emitted colour (zero for everything but DiffuseLight).
RT_HD double reflectance(double cosine, double ri) {  // material.rs:161-166
    double r0 = (1.0 - ri) / (1.0 + ri);
    double r0s = r0 * r0;
    // (1 - cosine)^5 by multiplication: within an ulp or two of the reference's powf(5.0) (material.rs:165), which the
    // device's pow does not reproduce bit for bit either, at a fraction of its instructions and registers
    const double x = 1.0 - cosine, x2 = x * x;
    return r0s + (1.0 - r0s) * (x2 * x2 * x);
}
// Whether scatter's first draw is random_unit_vector (Lambertian, Metal, Isotropic; in a plain scene everything
// but a light).
template <bool PLAIN = false>
RT_HD bool scatter_wants_unit_vector(const rt_material& m) {
    if (PLAIN) return m.kind != RT_MAT_DIFFUSE_LIGHT;
    return m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_METAL || m.kind == RT_MAT_ISOTROPIC;
}
// scatter with that vector already drawn (`ruv`, and rng past it); see scatter below.
// `tex_ready`: the material's texture value when the caller has evaluated it already (`tex_is_ready`).
RT_HD bool material_wants_texture(const rt_material& m) {
    return m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
}
// The materials the PLAIN code below covers: Lambertian, DiffuseLight, Isotropic over a solid colour.
RT_HD bool material_is_plain(const DScene& S, const rt_material& m) {
    return (m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC) &&
           S.texs[m.texture].kind == RT_TEX_SOLID;
}
// INLINE_PERLIN = false: the caller evaluates perlin textures itself (tex_is_ready for those hits) and this code holds
// none of their arithmetic.
template <bool PLAIN = false, bool INLINE_PERLIN = true>
RT_HD bool scatter_given(const DScene& S, const rt_material& m, V3 ray_d, const Rec& rec, Rng& rng, V3 ruv, V3& out_d,
                         V3& att, V3& emit, bool tex_is_ready = false, V3 tex_ready = V3{0.0, 0.0, 0.0}) {
    emit = mk(0, 0, 0);
    if (PLAIN) {
        // the three materials of a plain scene over solid colours: no checker / image / perlin code, no metal, no
        // dielectric (whose pow alone would set the kernel's register count)
        const rt_texture t = S.texs[m.texture];
        const V3 colour = mk(t.color[0], t.color[1], t.color[2]);
        if (m.kind == RT_MAT_DIFFUSE_LIGHT) { emit = colour; return false; }
        if (m.kind == RT_MAT_LAMBERTIAN) {
            V3 temp = rec.n + ruv;
            out_d = near_zero(temp) ? rec.n : temp;
        } else {
            out_d = ruv;
        }
        att = colour;
        return true;
    }
    // What several materials share is evaluated once, ahead of the switch, so that a wavefront with a mix of
    // materials runs the texture code once instead of once per material (and the caller the rejection loop: the
    // random unit vector is the first and only thing Lambertian, Metal and Isotropic draw).
    const bool wants_tex = material_wants_texture(m);
    V3 tex = mk(0, 0, 0);
    if (tex_is_ready) tex = tex_ready;
    else if (wants_tex) tex = tex_value<INLINE_PERLIN>(S, m.texture, rec.u, rec.v, rec.p);
    switch (m.kind) {
        case RT_MAT_LAMBERTIAN: {  // :45-68
            V3 temp = rec.n + ruv;
            out_d = near_zero(temp) ? rec.n : temp;
            att = tex;
            return true;
        }
        case RT_MAT_METAL: {  // :88-104
            V3 reflected = reflect(ray_d, rec.n);
            out_d = unit(reflected) + (m.param * ruv);
            att = mk(m.albedo[0], m.albedo[1], m.albedo[2]);
            return true;
        }
        case RT_MAT_DIELECTRIC: {  // :120-159
            double ratio = rec.front ? 1.0 / m.param : m.param;
            V3 ud = unit(ray_d);
            double cos_theta = fmin(dot(rec.n, -ud), 1.0);
            double sin_theta = sqrt(1.0 - cos_theta * cos_theta);
            bool cannot_refract = ratio * sin_theta > 1.0;
            bool refl = reflectance(cos_theta, ratio) > rng_range(rng, 0.0, 1.0);
            out_d = (cannot_refract | refl) ? reflect(ud, rec.n) : refract(ud, rec.n, ratio);
            att = mk(1.0, 1.0, 1.0);
            return true;
        }
        case RT_MAT_DIFFUSE_LIGHT: {  // :185-189
            emit = tex;
            return false;
        }
        default: {  // RT_MAT_ISOTROPIC :196-208
            out_d = ruv;
            att = tex;
            return true;
        }
    }
}
template <bool PLAIN = false>
RT_HD bool scatter(const DScene& S, const rt_material& m, V3 ray_d, const Rec& rec, Rng& rng, V3& out_d, V3& att,
                   V3& emit) {
    V3 ruv = mk(0, 0, 0);
    if (scatter_wants_unit_vector<PLAIN>(m)) ruv = random_unit_vector(rng);
    return scatter_given<PLAIN>(S, m, ray_d, rec, rng, ruv, out_d, att, emit);
}

// Camera::get_ray, camera.rs:479-507.
RT_HD void camera_ray(const rt_camera& c, uint32_t i, uint32_t j, Rng& rng, V3& o, V3& d) {
    double ox = rng_range(rng, -0.5, 0.5);
    double oy = rng_range(rng, -0.5, 0.5);
    V3 ps = ld3(c.pixel00) + (ld3(c.delta_u) * ((double)i + ox)) + (ld3(c.delta_v) * ((double)j + oy));
    if (c.defocus_angle <= 0.0) {
        o = ld3(c.center);
    } else {
        double px, py;
        for (;;) {  // random_in_unit_disk, point3/mod.rs:263-274
            px = rng_range(rng, -1.0, 1.0);
            py = rng_range(rng, -1.0, 1.0);
            if (px * px + py * py + 0.0 * 0.0 < 1.0) break;
        }
        o = ld3(c.center) + px * ld3(c.disk_u) + py * ld3(c.disk_v);
    }
    d = ps - o;
}

// point3/color.rs:17-22,49-68
RT_HD uint8_t to_byte(double lin) {
    double g = lin > 0.0 ? sqrt(lin) : 0.0;
    double c = g < 0.0 ? 0.0 : (g > 0.999 ? 0.999 : g);
    double s = 256.0 * c;
    if (!(s == s)) return 0;
    if (s >= 255.0) return 255;
    if (s <= 0.0) return 0;
    return (uint8_t)s;
}

}  // namespace rt
