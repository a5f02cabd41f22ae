// rt_oracle.cpp -- CPU restatement of villarjorge/raytracer's rendering path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing in the shipped library links, loads or calls
// this file; it exists so that tests and the benchmark's CPU-baseline leg have an
// independent statement of the reference algorithm to compare the HIP path with.
//
// Every function cites the reference file:line it follows (paths relative to the
// reference checkout).  Arithmetic is f64 with the reference's operation order and
// no FMA contraction (build with -ffp-contract=off), so surface t / normal / uv
// and every accept/reject branch are reproducible bit for bit.
//
// Two deliberate differences from the reference, both about randomness:
//  * the reference draws from an OS-seeded thread-local ChaCha12 (rand 0.9.2
//    ThreadRng, not present in the reference tree), so its sample streams cannot
//    be reproduced by anyone.  Here every draw comes from Philox4x32-10 keyed by
//    (seed; pixel, sample, draw index), mapped to f64 the way rand 0.9's
//    UniformFloat does it (52 mantissa bits, value*scale+low).
//  * primitive ids do not exist in the reference; they are assigned here by a
//    depth-first walk of the scene tree (first visit wins) so that "which primitive
//    was hit" can be compared.
//
// Parity pin status: src/tests.rs holds no test of hit/scatter.  The known answers
// the reference does hold (sphere.rs:87-89 uv table, scripts/sympy_quadric.py, the
// Point3 tests, block means of images/cornell_box_two_cylinders_medium_quality.png)
// are checked in tests/test_oracle_pins.py; everything else is "parity unpinned"
// and rests on this restatement following the source line by line.

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <limits>
#include <memory>
#include <sstream>
#include <string>
#include <thread>
#include <vector>

namespace orc {

static const double INF = std::numeric_limits<double>::infinity();
static const double PI = 3.14159265358979323846264338327950288;  // std::f64::consts::PI

// ---------------------------------------------------------------- point3/mod.rs
struct V3 {
    double x, y, z;
};
static inline V3 v3(double x, double y, double z) { return V3{x, y, z}; }
static inline V3 operator-(V3 a) { return V3{-a.x, -a.y, -a.z}; }                      // :70-81
static inline V3 operator+(V3 a, V3 b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }  // :83-94
static inline V3 operator-(V3 a, V3 b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }  // :111-122
static inline V3 operator*(V3 a, double s) { return V3{a.x * s, a.y * s, a.z * s}; }    // :124-135
static inline V3 operator*(double s, V3 a) { return V3{s * a.x, s * a.y, s * a.z}; }    // :146-157
static inline V3 operator*(V3 a, V3 b) { return V3{a.x * b.x, a.y * b.y, a.z * b.z}; }  // :159-170
static inline V3 operator/(V3 a, double s) { return a * (1.0 / s); }                    // :137-144
static inline double length_squared(V3 a) { return a.x * a.x + a.y * a.y + a.z * a.z; }  // :175-177
static inline double length(V3 a) { return std::sqrt(length_squared(a)); }              // :179-181
static inline double dot(V3 u, V3 v) { return u.x * v.x + u.y * v.y + u.z * v.z; }       // :183-191
static inline V3 cross(V3 u, V3 v) {                                                    // :194-200
    return V3{u.y * v.z - u.z * v.y, u.z * v.x - u.x * v.z, u.x * v.y - u.y * v.x};
}
static inline V3 rotate_y(V3 p, double c, double s) {  // :203-209
    return V3{c * p.x - s * p.z, p.y, s * p.x + c * p.z};
}
static inline bool is_near_zero(V3 a) {  // :214-217
    const double T = 1e-8;
    return std::fabs(a.x) < T && std::fabs(a.y) < T && std::fabs(a.z) < T;
}
static inline V3 unit_vector(V3 u) { return u / length(u); }  // :221-223
static inline V3 reflect(V3 v, V3 n) { return v - (2.0 * dot(v, n)) * n; }  // :279-281
static inline V3 refract(V3 uv, V3 n, double eta) {  // :284-293
    double cos_theta = std::fmin(dot(-uv, n), 1.0);
    V3 perp = (uv + n * cos_theta) * eta;
    V3 par = (-n) * std::sqrt(std::fabs(1.0 - length_squared(perp)));
    return perp + par;
}
static inline double idx(V3 a, int i) { return i == 0 ? a.x : (i == 1 ? a.y : a.z); }  // :54-68

// ---------------------------------------------------------------- random numbers
// Philox4x32-10 (Salmon et al., "Parallel random numbers: as easy as 1, 2, 3", SC'11).
static inline void philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1) {
    for (int r = 0; r < 10; ++r) {
        uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        uint32_t n1 = (uint32_t)p1;
        uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        uint32_t n3 = (uint32_t)p0;
        c[0] = n0; c[1] = n1; c[2] = n2; c[3] = n3;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

// One stream per (pixel, sample).  Draw number d lives in Philox block d>>1, words
// {0,1} for even d and {2,3} for odd d.  Replaces rand::random_range call sites
// camera.rs:503-504, point3/mod.rs:231-233,266-267, material.rs:139,
// constant_medium.rs:72.
struct Rng {
    uint32_t k0, k1, pixel, sample, stream, ctr;
    Rng(uint64_t seed, uint32_t pixel_, uint32_t sample_, uint32_t stream_, uint32_t ctr_ = 0)
        : k0((uint32_t)seed), k1((uint32_t)(seed >> 32)), pixel(pixel_), sample(sample_), stream(stream_), ctr(ctr_) {}
    double next01() {
        uint32_t c[4] = {pixel, sample, ctr >> 1, stream};
        philox4x32_10(c, k0, k1);
        uint64_t bits = (ctr & 1u) ? ((uint64_t)c[3] << 32 | c[2]) : ((uint64_t)c[1] << 32 | c[0]);
        ctr++;
        // rand 0.9 UniformFloat<f64>: 52 random mantissa bits -> [1,2) - 1.0.
        return (double)(bits >> 12) * (1.0 / 4503599627370496.0);
    }
    // rand 0.9 sample_single: value0_1 * (high - low) + low.
    double range(double lo, double hi) { return next01() * (hi - lo) + lo; }
};

struct Counters {
    uint64_t rays = 0, node_tests = 0, prim_tests = 0;
};
struct Ctx {
    Rng* rng;
    Counters* cnt;
};

static inline V3 random_vector(Rng& r, double a, double b) {  // point3/mod.rs:229-235
    double x = r.range(a, b);
    double y = r.range(a, b);
    double z = r.range(a, b);
    return V3{x, y, z};
}
static inline V3 random_in_unit_sphere(Rng& r) {  // :238-245
    for (;;) {
        V3 p = random_vector(r, -1.0, 1.0);
        if (length_squared(p) < 1.0) return p;
    }
}
static inline V3 random_unit_vector(Rng& r) { return unit_vector(random_in_unit_sphere(r)); }  // :248-250
static inline V3 random_in_unit_disk(Rng& r) {  // :263-274
    for (;;) {
        double x = r.range(-1.0, 1.0);
        double y = r.range(-1.0, 1.0);
        V3 p{x, y, 0.0};
        if (length_squared(p) < 1.0) return p;
    }
}

// ---------------------------------------------------------------- ray.rs
struct Ray {
    V3 origin, direction, inverse_direction;
    V3 at(double t) const { return origin + direction * t; }  // ray.rs:11-13
};
static inline Ray make_ray(V3 o, V3 d) {  // ray.rs:14-19
    return Ray{o, d, V3{1.0 / d.x, 1.0 / d.y, 1.0 / d.z}};
}

// ---------------------------------------------------------------- aabb.rs
struct Range {
    double start, end;
};
static Range pad_to_minimums(Range x) {  // aabb.rs:71-85
    const double DELTA = 0.0001;
    if (x.end < x.start) {
        std::fprintf(stderr, "oracle: tried to pad reversed range\n");
        std::abort();
    } else if (x.end - x.start < DELTA) {
        return Range{x.start + DELTA, x.end + DELTA};
    }
    return x;
}
struct AABB {
    Range x, y, z;
    const Range& axis(int n) const { return n == 0 ? x : (n == 1 ? y : z); }  // :240-248
    int longest_axis() const {                                                // :251-264
        double xs = x.end - x.start, ys = y.end - y.start, zs = z.end - z.start;
        if (xs > ys && xs > zs) return 0;
        if (ys > xs && ys > zs) return 1;
        return 2;
    }
    // aabb.rs:164-200.  Narrows ray_t in place; f64::max/min ignore a NaN operand,
    // which std::fmax/fmin also do.
    bool hit(const Ray& ray, Range& ray_t) const {
        for (int a = 0; a < 3; ++a) {
            const Range& ax = axis(a);
            double inv = idx(ray.inverse_direction, a);
            double t0 = (ax.start - idx(ray.origin, a)) * inv;
            double t1 = (ax.end - idx(ray.origin, a)) * inv;
            if (t0 < t1) {
                ray_t.start = std::fmax(ray_t.start, t0);
                ray_t.end = std::fmin(ray_t.end, t1);
            } else {
                ray_t.start = std::fmax(ray_t.start, t1);
                ray_t.end = std::fmin(ray_t.end, t0);
            }
            if (ray_t.end <= ray_t.start) return false;
        }
        return true;
    }
};
static AABB aabb_from_points(V3 a, V3 b) {  // aabb.rs:16-26
    Range x = a.x <= b.x ? Range{a.x, b.x} : Range{b.x, a.x};
    Range y = a.y <= b.y ? Range{a.y, b.y} : Range{b.y, a.y};
    Range z = a.z <= b.z ? Range{a.z, b.z} : Range{b.z, a.z};
    return AABB{pad_to_minimums(x), pad_to_minimums(y), pad_to_minimums(z)};
}
static AABB aabb_default() { return aabb_from_points(V3{0, 0, 0}, V3{0, 0, 0}); }  // :29-33
static AABB aabb_shift(const AABB& b, V3 r) {                                      // :53-68
    return AABB{Range{b.x.start + r.x, b.x.end + r.x}, Range{b.y.start + r.y, b.y.end + r.y},
                Range{b.z.start + r.z, b.z.end + r.z}};
}
static Range unite_ranges(const Range& a, const Range& b) {  // :88-90
    return Range{std::fmin(a.start, b.start), std::fmax(a.end, b.end)};
}
static AABB join_aabbs(const AABB& a, const AABB& b) {  // :93-103
    return AABB{pad_to_minimums(unite_ranges(a.x, b.x)), pad_to_minimums(unite_ranges(a.y, b.y)),
                pad_to_minimums(unite_ranges(a.z, b.z))};
}
static inline bool contains(double start, double end, double t) { return start <= t && t < end; }  // Range::contains

// ---------------------------------------------------------------- texture.rs / perlin.rs
struct Texture {
    virtual ~Texture() {}
    virtual V3 value(double u, double v, V3 p) const = 0;  // texture.rs:11-13
};
struct SolidColor : Texture {  // texture.rs:15-29
    V3 albedo;
    explicit SolidColor(V3 c) : albedo(c) {}
    V3 value(double, double, V3) const override { return albedo; }
};
struct CheckerTexture : Texture {  // texture.rs:31-80
    double inverse_scale;
    std::shared_ptr<Texture> even, odd;
    V3 value(double u, double v, V3 p) const override {
        int64_t xi = (int64_t)std::floor(inverse_scale * p.x);
        int64_t yi = (int64_t)std::floor(inverse_scale * p.y);
        int64_t zi = (int64_t)std::floor(inverse_scale * p.z);
        bool is_even = (xi + yi + zi) % 2 == 0;
        return is_even ? even->value(u, v, p) : odd->value(u, v, p);
    }
};
static inline double clamp01(double x) { return x < 0.0 ? 0.0 : (x > 1.0 ? 1.0 : x); }
struct ImageTexture : Texture {  // texture.rs:84-136
    std::vector<uint8_t> rgb;
    uint32_t w = 0, h = 0;
    V3 value(double u, double v, V3) const override {
        double cu = clamp01(u);
        double cv = 1.0 - clamp01(v);
        uint32_t ui = (uint32_t)((double)w * cu);
        uint32_t vi = (uint32_t)((double)h * cv);
        // texture.rs:124-127 indexes without clamping and panics at u==1 / v==0
        // exactly; clamp to the last texel instead (measure-zero deviation).
        if (ui >= w) ui = w - 1;
        if (vi >= h) vi = h - 1;
        const uint8_t* px = &rgb[3 * ((size_t)vi * w + ui)];
        double s = 1.0 / 255.0;
        return V3{(double)px[0] * s, (double)px[1] * s, (double)px[2] * s};
    }
};
struct PerlinNoise {  // perlin.rs:11-16
    V3 random_vectors[256];
    uint32_t x_perm[256], y_perm[256], z_perm[256];
    static double interp(const V3 c[2][2][2], double u, double v, double w) {  // :115-143
        double uu = u * u * (3.0 - 2.0 * u);
        double vv = v * v * (3.0 - 2.0 * v);
        double ww = w * w * (3.0 - 2.0 * w);
        double accum = 0.0;
        for (int i = 0; i < 2; ++i)
            for (int j = 0; j < 2; ++j)
                for (int k = 0; k < 2; ++k) {
                    double fi = (double)i, fj = (double)j, fk = (double)k;
                    V3 weight_v{u - fi, v - fj, w - fk};
                    accum += (fi * uu + (1.0 - fi) * (1.0 - uu)) * (fj * vv + (1.0 - fj) * (1.0 - vv)) *
                             (fk * ww + (1.0 - fk) * (1.0 - ww)) * dot(c[i][j][k], weight_v);
                }
        return accum;
    }
    double noise(V3 p) const {  // :46-77
        double u = p.x - std::floor(p.x);
        double v = p.y - std::floor(p.y);
        double w = p.z - std::floor(p.z);
        int64_t i = (int64_t)std::floor(p.x);
        int64_t j = (int64_t)std::floor(p.y);
        int64_t k = (int64_t)std::floor(p.z);
        V3 c[2][2][2];
        for (int64_t di = 0; di < 2; ++di)
            for (int64_t dj = 0; dj < 2; ++dj)
                for (int64_t dk = 0; dk < 2; ++dk) {
                    uint32_t fi = x_perm[(i + di) & 255] ^ y_perm[(j + dj) & 255] ^ z_perm[(k + dk) & 255];
                    c[di][dj][dk] = random_vectors[fi];
                }
        return interp(c, u, v, w);
    }
    double turbulence(V3 p, int depth) const {  // :79-92
        double accum = 0.0;
        V3 temp_p = p;
        double weight = 1.0;
        for (int i = 0; i < depth; ++i) {
            accum += weight * noise(temp_p);
            weight *= 0.5;
            temp_p = 2.0 * temp_p;
        }
        return std::fabs(accum);
    }
};
struct PerlinNoiseTexture : Texture {  // texture.rs:138-158
    PerlinNoise noise;
    double scale;
    V3 value(double, double, V3 p) const override {
        return V3{0.5, 0.5, 0.5} * (1.0 + std::sin(scale * p.z + 10.0 * noise.turbulence(p, 7)));
    }
};

// ---------------------------------------------------------------- hittable/mod.rs, material.rs
struct Material;
struct HitRecord {  // hittable/mod.rs:18-31 (+ prim_id, see header)
    V3 p{0, 0, 0}, normal{0, 0, 0};
    const Material* material = nullptr;
    double t = 0.0;
    double u = 0.0, v = 0.0;
    bool front_face = false;
    uint32_t prim_id = 0xFFFFFFFFu;
    void set_face_normal(const Ray& ray, V3 outward_normal) {  // :51-58
        front_face = dot(outward_normal, ray.direction) < 0.0;
        normal = front_face ? outward_normal : -outward_normal;
    }
};

struct Material {  // material.rs:18-31
    virtual ~Material() {}
    virtual bool scatter(const Ray&, const HitRecord&, Ray&, V3&, Rng&) const { return false; }
    virtual V3 emitted(double, double, V3) const { return V3{0, 0, 0}; }
};
struct Lambertian : Material {  // material.rs:41-68
    std::shared_ptr<Texture> texture;
    bool scatter(const Ray&, const HitRecord& rec, Ray& out, V3& att, Rng& rng) const override {
        V3 temp = rec.normal + random_unit_vector(rng);
        V3 dir = is_near_zero(temp) ? rec.normal : temp;
        out = make_ray(rec.p, dir);
        att = texture->value(rec.u, rec.v, rec.p);
        return true;
    }
};
struct Metal : Material {  // material.rs:83-104
    V3 albedo;
    double fuzz;
    bool scatter(const Ray& in, const HitRecord& rec, Ray& out, V3& att, Rng& rng) const override {
        V3 reflected = reflect(in.direction, rec.normal);
        V3 fuzzed = unit_vector(reflected) + (fuzz * random_unit_vector(rng));
        out = make_ray(rec.p, fuzzed);
        att = albedo;
        return true;
    }
};
static double reflectance(double cosine, double ri) {  // material.rs:161-166
    double r0 = (1.0 - ri) / (1.0 + ri);
    double r0s = r0 * r0;
    return r0s + (1.0 - r0s) * std::pow(1.0 - cosine, 5.0);
}
struct Dielectric : Material {  // material.rs:114-159
    double refraction_index;
    bool scatter(const Ray& in, const HitRecord& rec, Ray& out, V3& att, Rng& rng) const override {
        double ratio = rec.front_face ? 1.0 / refraction_index : refraction_index;
        V3 unit_direction = unit_vector(in.direction);
        double cos_theta = std::fmin(dot(rec.normal, -unit_direction), 1.0);
        double sin_theta = std::sqrt(1.0 - cos_theta * cos_theta);
        bool cannot_refract = ratio * sin_theta > 1.0;
        bool refl = reflectance(cos_theta, ratio) > rng.range(0.0, 1.0);  // always drawn (non-short-circuit |)
        V3 direction = (cannot_refract | refl) ? reflect(unit_direction, rec.normal)
                                               : refract(unit_direction, rec.normal, ratio);
        out = make_ray(rec.p, direction);
        att = V3{1.0, 1.0, 1.0};
        return true;
    }
};
struct DiffuseLight : Material {  // material.rs:173-189
    std::shared_ptr<Texture> texture;
    V3 emitted(double u, double v, V3 p) const override { return texture->value(u, v, p); }
};
struct Isotropic : Material {  // material.rs:192-208
    std::shared_ptr<Texture> texture;
    bool scatter(const Ray&, const HitRecord& rec, Ray& out, V3& att, Rng& rng) const override {
        out = make_ray(rec.p, random_unit_vector(rng));
        att = texture->value(rec.u, rec.v, rec.p);
        return true;
    }
};

struct Hittable {  // hittable/mod.rs:65-68
    AABB bbox;
    uint32_t prim_id = 0xFFFFFFFFu;
    virtual ~Hittable() {}
    virtual bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const = 0;
    // depth-first id assignment (first visit wins); not in the reference.
    virtual void assign_ids(uint32_t& next) {
        if (prim_id == 0xFFFFFFFFu) prim_id = next++;
    }
    virtual void count(uint64_t* c) const = 0;  // c[0..7] per-kind object counts
};
typedef std::shared_ptr<Hittable> HPtr;

struct Translate : Hittable {  // hittable/mod.rs:71-107
    HPtr object;
    V3 offset;
    Translate(HPtr o, V3 off) : object(o), offset(off) { bbox = aabb_shift(o->bbox, off); }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        Ray offset_ray = make_ray(ray.origin - offset, ray.direction);
        if (!object->hit(offset_ray, ray_t, rec, cx)) return false;
        rec.p = rec.p + offset;
        return true;
    }
    void assign_ids(uint32_t& next) override { object->assign_ids(next); }
    void count(uint64_t* c) const override { c[6]++; object->count(c); }
};
struct RotateY : Hittable {  // hittable/mod.rs:110-189
    HPtr object;
    double sin_theta, cos_theta;
    RotateY(HPtr o, double angle_in_degrees) : object(o) {
        double radians = angle_in_degrees * (PI / 180.0);  // f64::to_radians
        sin_theta = std::sin(radians);
        cos_theta = std::cos(radians);
        const AABB& bb = o->bbox;
        double mn[3] = {INF, INF, INF}, mx[3] = {-INF, -INF, -INF};
        for (int ii = 0; ii < 2; ++ii)
            for (int jj = 0; jj < 2; ++jj)
                for (int kk = 0; kk < 2; ++kk) {
                    double i = ii, j = jj, k = kk;
                    double x = i * bb.x.end + (1.0 - i) * bb.x.start;
                    double y = j * bb.y.end + (1.0 - j) * bb.y.start;
                    double z = k * bb.z.end + (1.0 - k) * bb.z.start;
                    double xn = cos_theta * x + sin_theta * z;
                    double zn = -sin_theta * x + cos_theta * z;
                    double tester[3] = {xn, y, zn};
                    // :174-177 as written: only x and y, and `min` for the maximum.
                    for (int c = 0; c < 2; ++c) {
                        mn[c] = std::fmin(mn[c], tester[c]);
                        mx[c] = std::fmin(mx[c], tester[c]);
                    }
                }
        bbox = aabb_from_points(V3{mn[0], mn[1], mn[2]}, V3{mx[0], mx[1], mx[2]});
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        V3 o = rotate_y(ray.origin, cos_theta, sin_theta);
        V3 d = rotate_y(ray.direction, cos_theta, sin_theta);
        Ray rotated = make_ray(o, d);
        if (!object->hit(rotated, ray_t, rec, cx)) return false;
        rec.p = rotate_y(rec.p, cos_theta, -sin_theta);
        rec.normal = rotate_y(rec.normal, cos_theta, -sin_theta);
        return true;
    }
    void assign_ids(uint32_t& next) override { object->assign_ids(next); }
    void count(uint64_t* c) const override { c[7]++; object->count(c); }
};

// ---------------------------------------------------------------- sphere.rs
static void get_sphere_uv(V3 p, double& u, double& v) {  // sphere.rs:90-98
    double theta = std::acos(-p.y);
    double phi = std::atan2(-p.z, p.x) + PI;
    u = phi / (2.0 * PI);
    v = theta / PI;
}
struct Sphere : Hittable {  // sphere.rs:12-80
    V3 center;
    double radius;
    std::shared_ptr<Material> material;
    Sphere(V3 c, double r, std::shared_ptr<Material> m) : center(c), radius(std::fmax(r, 0.0)), material(m) {
        V3 rv{r, r, r};
        bbox = aabb_from_points(c - rv, c + rv);
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        cx.cnt->prim_tests++;
        V3 oc = center - ray.origin;
        double a = length_squared(ray.direction);
        double h = dot(oc, ray.direction);
        double c = length_squared(oc) - radius * radius;
        double discriminant = h * h - a * c;
        if (discriminant < 0.0) return false;
        double sq = std::sqrt(discriminant);
        double root = (h - sq) / a;
        if (!contains(ray_t.start, ray_t.end, root)) {
            root = (h + sq) / a;
            if (!contains(ray_t.start, ray_t.end, root)) return false;
        }
        rec.t = root;
        rec.p = ray.at(rec.t);
        V3 outward = (ray.at(root) - center) / radius;
        rec.set_face_normal(ray, outward);
        rec.material = material.get();
        rec.prim_id = prim_id;
        get_sphere_uv(outward, rec.u, rec.v);
        return true;
    }
    void count(uint64_t* c) const override { c[0]++; }
};

// ---------------------------------------------------------------- parallelogram.rs / triangle.rs
struct Planar : Hittable {
    double d;
    V3 normal, q, u, v, w;
    std::shared_ptr<Material> material;
    void derive(V3 n) {  // parallelogram.rs:43-46, triangle.rs:42-45
        normal = unit_vector(n);
        d = dot(normal, q);
        w = n / dot(n, n);
    }
};
struct Parallelogram : Planar {  // parallelogram.rs:12-136
    Parallelogram(V3 q_, V3 u_, V3 v_, std::shared_ptr<Material> m) {
        q = q_; u = u_; v = v_; material = m;
        AABB b0 = aabb_from_points(q, q + u + v);  // :31-37
        AABB b1 = aabb_from_points(q + u, q + v);
        bbox = join_aabbs(b0, b1);
        derive(cross(u, v));
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        cx.cnt->prim_tests++;
        double denominator = dot(normal, ray.direction);
        double t = (d - dot(normal, ray.origin)) / denominator;
        if (ray_t.end < t || t < ray_t.start) return false;  // closed interval, :103
        if (std::fabs(denominator) < 1e-8) return false;
        V3 intersection = ray.at(t);
        V3 ph = intersection - q;
        double alpha = dot(w, cross(ph, v));
        double beta = dot(w, cross(u, ph));
        // is_interior :139-143
        if (!(contains(0.0, 1.0, alpha) && contains(0.0, 1.0, beta))) return false;
        rec.u = alpha;
        rec.v = beta;
        rec.t = t;
        rec.p = intersection;
        rec.material = material.get();
        rec.prim_id = prim_id;
        rec.set_face_normal(ray, normal);
        return true;
    }
    void count(uint64_t* c) const override { c[1]++; }
};
struct Triangle : Planar {  // triangle.rs:11-165
    Triangle(V3 q_, V3 u_, V3 v_, std::shared_ptr<Material> m) {
        q = q_; u = u_; v = v_; material = m;
        AABB b0 = aabb_from_points(q, q + u);  // :30-36
        AABB b1 = aabb_from_points(q, q + v);
        bbox = join_aabbs(b0, b1);
        derive(cross(u, v));
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        cx.cnt->prim_tests++;
        double denominator = dot(normal, ray.direction);
        if (std::fabs(denominator) < 1e-8) return false;
        double t = (d - dot(normal, ray.origin)) / denominator;
        if (!contains(ray_t.start, ray_t.end, t)) return false;
        V3 intersection = ray.at(t);
        V3 ph = intersection - q;
        double alpha = dot(w, cross(ph, v));
        double beta = dot(w, cross(u, ph));
        if (!(alpha > 0.0 && beta > 0.0 && alpha + beta < 1.0)) return false;  // :163-165
        rec.u = alpha;
        rec.v = beta;
        rec.t = t;
        rec.p = intersection;
        rec.material = material.get();
        rec.prim_id = prim_id;
        rec.set_face_normal(ray, normal);
        return true;
    }
    void count(uint64_t* c) const override { c[2]++; }
};

// ---------------------------------------------------------------- quadric.rs
static inline V3 prod1(V3 u, V3 v) { return V3{u.x * v.y, u.x * v.z, u.y * v.z}; }  // :91-97
static inline V3 anticross(V3 u, V3 v) {                                            // :99-105
    return V3{u.x * v.y + u.y * v.x, u.x * v.z + u.z * v.x, u.y * v.z + u.z * v.y};
}
struct Quadric : Hittable {  // quadric.rs:14-89
    V3 p1, p2, p3;
    double j;
    std::shared_ptr<Material> material;
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        cx.cnt->prim_tests++;
        V3 o = ray.origin, d = ray.direction;
        double a = dot(p1, d * d) + dot(p2, prod1(d, d));
        double h = -dot(p1, d * o) - 0.5 * dot(p2, anticross(d, o)) - 0.5 * dot(p3, d);
        double c = dot(p1, o * o) + dot(p2, anticross(o, o)) + dot(p3, o) + j;
        double discriminant = h * h - a * c;
        if (discriminant < 0.0) return false;
        double sq = std::sqrt(discriminant);
        double root = (h - sq) / a;
        if (!contains(ray_t.start, ray_t.end, root)) {
            root = (h + sq) / a;
            if (!contains(ray_t.start, ray_t.end, root)) return false;
        }
        V3 p = ray.at(root);
        V3 n = V3{2.0 * p1.x * p.x + p2.x * p.y + p2.y * p.z, 2.0 * p1.y * p.y + p2.x * p.x + p2.z * p.z,
                  2.0 * p1.z * p.z + p2.y * p.x + p2.z * p.y} +
               p3;
        rec.t = root;
        rec.p = p;
        rec.set_face_normal(ray, unit_vector(n));
        rec.material = material.get();
        rec.prim_id = prim_id;
        rec.u = 0.0;
        rec.v = 0.0;
        return true;
    }
    void count(uint64_t* c) const override { c[3]++; }
};
static std::shared_ptr<Quadric> y_cylinder(V3 center, double radius, std::shared_ptr<Material> m) {  // :110-135
    auto q = std::make_shared<Quadric>();
    V3 rv{radius, radius, radius};
    q->bbox = aabb_from_points(center - rv, center + rv);
    q->p1 = V3{1.0, 0.0, 1.0};
    q->p2 = V3{0, 0, 0};
    q->p3 = center * V3{-2.0, 0.0, -2.0};
    q->j = -radius * radius + center.x * center.x + center.z * center.z;
    q->material = m;
    return q;
}
static std::shared_ptr<Quadric> quadric_sphere(V3 center, double radius, std::shared_ptr<Material> m) {  // :138-158
    auto q = std::make_shared<Quadric>();
    V3 rv{radius, radius, radius};
    q->bbox = aabb_from_points(center - rv, center + rv);
    q->p1 = V3{1.0, 1.0, 1.0};
    q->p2 = V3{0, 0, 0};
    q->p3 = -2.0 * center;
    q->j = -radius * radius + dot(center, center);
    q->material = m;
    return q;
}
static std::shared_ptr<Quadric> y_cone(V3 center, V3 offset, std::shared_ptr<Material> m) {  // :161-177
    auto q = std::make_shared<Quadric>();
    q->bbox = aabb_from_points(center - offset, center + offset);
    V3 ind{1.0, -1.0, 1.0};
    q->p1 = ind;
    q->p2 = V3{0, 0, 0};
    q->p3 = (-2.0 * ind) * center;
    q->j = dot(center * ind, center);
    q->material = m;
    return q;
}

// ---------------------------------------------------------------- hittable_list.rs
struct HittableList {  // :13-47
    std::vector<HPtr> objects;
    AABB bounding_box = aabb_default();
    void add(HPtr o) {
        bounding_box = join_aabbs(bounding_box, o->bbox);
        objects.push_back(o);
    }
};
struct HittableSlice : Hittable {  // :72-106
    std::vector<HPtr> objects;
    explicit HittableSlice(const HittableList& l) : objects(l.objects) { bbox = l.bounding_box; }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        HitRecord temp = rec;
        bool hit_anything = false;
        double closest_so_far = ray_t.end;
        for (const HPtr& o : objects) {
            if (o->hit(ray, Range{ray_t.start, closest_so_far}, temp, cx)) {
                hit_anything = true;
                closest_so_far = temp.t;
                rec = temp;
            }
        }
        return hit_anything;
    }
    void assign_ids(uint32_t& next) override {
        for (auto& o : objects) o->assign_ids(next);
    }
    void count(uint64_t* c) const override {
        for (auto& o : objects) o->count(c);
    }
};

// ---------------------------------------------------------------- bvh.rs
static inline int64_t total_key(double x) {  // f64::total_cmp ordering key
    int64_t b;
    std::memcpy(&b, &x, 8);
    b ^= (int64_t)((uint64_t)(b >> 63) >> 1);
    return b;
}
struct BVHNode : Hittable {  // bvh.rs:13-139
    bool leaf = false;
    std::shared_ptr<HittableSlice> objects;  // Leaf
    HPtr left, right;                        // Internal
    explicit BVHNode(std::vector<HPtr> objs) {
        AABB bb = aabb_default();
        for (auto& o : objs) bb = join_aabbs(bb, o->bbox);
        int axis = bb.longest_axis();
        const size_t THRESHOLD = 16;
        if (objs.size() <= THRESHOLD) {
            HittableList l;
            for (auto& o : objs) l.add(o);
            bb = l.bounding_box;
            leaf = true;
            objects = std::make_shared<HittableSlice>(l);
            bbox = bb;
        } else {
            std::stable_sort(objs.begin(), objs.end(), [axis](const HPtr& a, const HPtr& b) {
                return total_key(a->bbox.axis(axis).start) < total_key(b->bbox.axis(axis).start);
            });
            size_t mid = objs.size() / 2;
            std::vector<HPtr> upper(objs.begin() + mid, objs.end());  // split_off(mid)
            objs.resize(mid);
            left = std::make_shared<BVHNode>(upper);
            right = std::make_shared<BVHNode>(objs);
            bbox = bb;
        }
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {  // :29-60
        cx.cnt->node_tests++;
        Range reduced = ray_t;
        if (!bbox.hit(ray, reduced)) return false;
        if (leaf) return objects->hit(ray, reduced, rec, cx);
        bool hit_left = left->hit(ray, reduced, rec, cx);
        bool hit_right = right->hit(ray, Range{reduced.start, hit_left ? rec.t : reduced.end}, rec, cx);
        return hit_left || hit_right;
    }
    void assign_ids(uint32_t& next) override {
        if (leaf) objects->assign_ids(next);
        else { left->assign_ids(next); right->assign_ids(next); }
    }
    void count(uint64_t* c) const override {
        c[5]++;
        if (leaf) objects->count(c);
        else { left->count(c); right->count(c); }
    }
    int depth() const {
        if (leaf) return 1;
        int l = 1, r = 1;
        if (auto* b = dynamic_cast<BVHNode*>(left.get())) l = b->depth();
        if (auto* b = dynamic_cast<BVHNode*>(right.get())) r = b->depth();
        return 1 + std::max(l, r);
    }
};

// ---------------------------------------------------------------- constant_medium.rs
struct ConstantMedium : Hittable {  // :14-98
    HPtr boundary;
    double neg_inv_density;
    std::shared_ptr<Material> phase_function;
    ConstantMedium(HPtr b, double density, std::shared_ptr<Texture> tex) : boundary(b), neg_inv_density(-1.0 / density) {
        auto iso = std::make_shared<Isotropic>();
        iso->texture = tex;
        phase_function = iso;
        bbox = b->bbox;
    }
    bool hit(const Ray& ray, Range ray_t, HitRecord& rec, Ctx& cx) const override {
        cx.cnt->prim_tests++;
        HitRecord r1 = rec, r2 = rec;
        if (!boundary->hit(ray, Range{-INF, INF}, r1, cx)) return false;
        if (!boundary->hit(ray, Range{r1.t + 0.0001, INF}, r2, cx)) return false;
        r1.t = std::fmax(r1.t, ray_t.start);
        r2.t = std::fmin(r2.t, ray_t.end);
        r1.t = std::fmax(r1.t, 0.0);
        double ray_length = length(ray.direction);
        double distance_inside_boundary = (r2.t - r1.t) * ray_length;
        double x = cx.rng->range(0.0, 1.0);
        double hit_distance = neg_inv_density * std::log(x);
        if (hit_distance > distance_inside_boundary) return false;
        rec.t = r1.t + hit_distance / ray_length;
        rec.p = ray.at(rec.t);
        rec.normal = V3{1.0, 0.0, 0.0};
        rec.front_face = true;
        rec.u = 0.0;
        rec.v = 0.0;
        rec.material = phase_function.get();
        rec.prim_id = prim_id;
        return true;
    }
    void assign_ids(uint32_t& next) override {
        if (prim_id == 0xFFFFFFFFu) prim_id = next++;
        boundary->assign_ids(next);
    }
    void count(uint64_t* c) const override { c[4]++; boundary->count(c); }
};

// ---------------------------------------------------------------- load_obj.rs
static bool load_model(const char* path, double scale, std::shared_ptr<Material> m, std::vector<HPtr>& tris,
                       std::string& err) {  // :15-115
    std::ifstream f(path);
    if (!f) { err = std::string("cannot open ") + path; return false; }
    std::vector<V3> verts;
    std::vector<std::vector<size_t>> faces;
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ls(line);
        std::string first;
        if (!(ls >> first)) continue;
        if (first == "v") {
            std::vector<double> c;
            std::string tok;
            while (ls >> tok) c.push_back(std::strtod(tok.c_str(), nullptr));
            if (c.size() != 3) { err = "unable to parse non 3d coordinates"; return false; }
            verts.push_back(V3{c[0], c[1], c[2]});
        } else if (first == "f") {
            std::vector<size_t> vs;
            std::string tok;
            while (ls >> tok) {
                size_t slash = tok.find('/');
                std::string head = slash == std::string::npos ? tok : tok.substr(0, slash);
                char* endp = nullptr;
                long long i = std::strtoll(head.c_str(), &endp, 10);
                if (endp == head.c_str() || *endp != 0 || i < 1) { err = "bad face index: " + tok; return false; }
                vs.push_back((size_t)i - 1);
            }
            faces.push_back(vs);
        } else if (first == "vn" || first == "#" || first == "vt" || first == "s" || first == "o" || first == "vp" ||
                   first == "usemtl" || first == "mtllib" || first == "g") {
            // ignored, :55-73
        } else {
            err = "Unexpected line in .obj file: " + line;
            return false;
        }
    }
    for (auto& face : faces) {  // fan triangulation :88-106
        if (face.empty()) continue;
        for (size_t ix : face)
            if (ix >= verts.size()) { err = "face index out of range"; return false; }
        V3 c0 = verts[face[0]];
        V3 v1{scale * c0.x, scale * c0.y, scale * c0.z};
        for (size_t k = 1; k + 1 < face.size(); ++k) {
            V3 c2 = verts[face[k]], c3 = verts[face[k + 1]];
            V3 v2{scale * c2.x, scale * c2.y, scale * c2.z};
            V3 v3_{scale * c3.x, scale * c3.y, scale * c3.z};
            // Triangle::from_vertex_locations, triangle.rs:58-87
            tris.push_back(std::make_shared<Triangle>(v1, v2 - v1, v3_ - v1, m));
        }
    }
    return true;
}

// ---------------------------------------------------------------- camera.rs
struct Camera {  // camera.rs:23-37
    uint32_t image_width, image_height;
    V3 pixel00_loc, pixel_delta_u, pixel_delta_v, camera_center;
    double defocus_angle;
    V3 defocus_disk_u, defocus_disk_v;
};
static Camera camera_new(double aspect_ratio, uint32_t image_width, double vfov, double defocus_angle,
                         double focus_distance, V3 look_from, V3 look_at, V3 view_up) {  // :80-143
    Camera c;
    uint32_t h_ = (uint32_t)((double)image_width / aspect_ratio);
    c.image_width = image_width;
    c.image_height = h_ < 1 ? 1 : h_;
    c.camera_center = look_from;
    double theta = vfov * (PI / 180.0);
    double h = std::tan(theta * 0.5);
    double viewport_height = 2.0 * h * focus_distance;
    double viewport_width = viewport_height * ((double)image_width / (double)c.image_height);
    V3 w = unit_vector(look_from - look_at);
    V3 u = unit_vector(cross(view_up, w));
    V3 v = unit_vector(cross(w, u));
    V3 viewport_u = u * viewport_width;
    V3 viewport_v = v * (-viewport_height);
    c.pixel_delta_u = viewport_u / (double)image_width;
    c.pixel_delta_v = viewport_v / (double)c.image_height;
    V3 upper_left = look_from - w * focus_distance - viewport_u * 0.5 - viewport_v * 0.5;
    c.pixel00_loc = upper_left + (c.pixel_delta_u + c.pixel_delta_v) * 0.5;
    double defocus_radius = focus_distance * std::tan((defocus_angle * 0.5) * (PI / 180.0));
    c.defocus_disk_u = u * defocus_radius;
    c.defocus_disk_v = v * defocus_radius;
    c.defocus_angle = defocus_angle;
    return c;
}
static Ray get_ray(const Camera& c, uint32_t i, uint32_t j, Rng& rng) {  // :479-507
    double ox = rng.range(-0.5, 0.5);
    double oy = rng.range(-0.5, 0.5);
    V3 pixel_sample = c.pixel00_loc + (c.pixel_delta_u * ((double)i + ox)) + (c.pixel_delta_v * ((double)j + oy));
    V3 origin;
    if (c.defocus_angle <= 0.0) {
        origin = c.camera_center;
    } else {
        V3 p = random_in_unit_disk(rng);
        origin = c.camera_center + p.x * c.defocus_disk_u + p.y * c.defocus_disk_v;
    }
    return make_ray(origin, pixel_sample - origin);
}

// camera.rs:349-408 (ray_color; ray_color2 at :410-474 is the same body).
static V3 ray_color(const Ray& ray, uint32_t depth, const Hittable& world, V3 background, Ctx& cx,
                    Ray* capture, uint32_t capture_bounce, uint32_t bounce, bool* captured, uint32_t* capture_ctr) {
    if (depth == 0) return V3{0, 0, 0};
    if (capture && bounce == capture_bounce) {
        *capture = ray;
        *captured = true;
        *capture_ctr = cx.rng->ctr;
    }
    HitRecord rec;
    cx.cnt->rays++;
    if (!world.hit(ray, Range{0.001, INF}, rec, cx)) return background;
    V3 emission = rec.material->emitted(rec.u, rec.v, rec.p);
    Ray scattered = make_ray(V3{0, 0, 0}, V3{1, 1, 1});
    V3 attenuation{1.0, 1.0, 1.0};
    if (!rec.material->scatter(ray, rec, scattered, attenuation, *cx.rng)) return emission;
    V3 from_scatter = attenuation * ray_color(scattered, depth - 1, world, background, cx, capture, capture_bounce,
                                              bounce + 1, captured, capture_ctr);
    return emission + from_scatter;
}

// ---------------------------------------------------------------- point3/color.rs
static inline double linear_to_gamma(double x) { return x > 0.0 ? std::sqrt(x) : 0.0; }  // :17-22
static inline uint8_t to_byte(double g) {                                                 // :61-65
    // f64::clamp keeps NaN; `as u8` saturates and maps NaN to 0.
    double c = g < 0.0 ? 0.0 : (g > 0.999 ? 0.999 : g);
    double s = 256.0 * c;
    if (!(s == s)) return 0;
    if (s <= 0.0) return 0;
    if (s >= 255.0) return 255;
    return (uint8_t)s;
}
static void proccess_color(V3 c, uint8_t out[3]) {  // :49-68
    out[0] = to_byte(linear_to_gamma(c.x));
    out[1] = to_byte(linear_to_gamma(c.y));
    out[2] = to_byte(linear_to_gamma(c.z));
}

// ---------------------------------------------------------------- builder (test harness plumbing)
struct Scene {
    std::vector<std::shared_ptr<Texture>> textures;
    std::vector<std::shared_ptr<Material>> materials;
    std::vector<HPtr> objects;
    std::vector<HittableList> lists;
    HPtr world;
    uint32_t n_prims = 0;
    std::string err;
};

}  // namespace orc

using namespace orc;

#define ORC_API extern "C" __attribute__((visibility("default")))

static V3 rd3(const double* p) { return V3{p[0], p[1], p[2]}; }
static bool ok_tex(Scene* s, int t) { return t >= 0 && (size_t)t < s->textures.size(); }
static bool ok_mat(Scene* s, int m) { return m >= 0 && (size_t)m < s->materials.size(); }
static bool ok_obj(Scene* s, int o) { return o >= 0 && (size_t)o < s->objects.size(); }
static bool ok_list(Scene* s, int l) { return l >= 0 && (size_t)l < s->lists.size(); }
static int push_obj(Scene* s, HPtr o) { s->objects.push_back(o); return (int)s->objects.size() - 1; }

ORC_API Scene* orc_scene_new() { return new Scene(); }
ORC_API void orc_scene_free(Scene* s) { delete s; }
ORC_API const char* orc_scene_error(Scene* s) { return s->err.c_str(); }

ORC_API int orc_tex_solid(Scene* s, double r, double g, double b) {
    s->textures.push_back(std::make_shared<SolidColor>(V3{r, g, b}));
    return (int)s->textures.size() - 1;
}
ORC_API int orc_tex_checker(Scene* s, double scale, int even, int odd) {
    if (!ok_tex(s, even) || !ok_tex(s, odd)) return -1;
    auto t = std::make_shared<CheckerTexture>();
    t->inverse_scale = 1.0 / scale;
    t->even = s->textures[even];
    t->odd = s->textures[odd];
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
ORC_API int orc_tex_image(Scene* s, const uint8_t* rgb, uint32_t w, uint32_t h) {
    if (!rgb || !w || !h) return -1;
    auto t = std::make_shared<ImageTexture>();
    t->rgb.assign(rgb, rgb + (size_t)3 * w * h);
    t->w = w;
    t->h = h;
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
// vec768: 256 vectors (x,y,z); perm is the identity as at perlin.rs:33-43 (the shuffle loop never runs).
ORC_API int orc_tex_perlin(Scene* s, double scale, const double* vec768) {
    auto t = std::make_shared<PerlinNoiseTexture>();
    for (int i = 0; i < 256; ++i) {
        t->noise.random_vectors[i] = V3{vec768[3 * i], vec768[3 * i + 1], vec768[3 * i + 2]};
        t->noise.x_perm[i] = t->noise.y_perm[i] = t->noise.z_perm[i] = (uint32_t)i;
    }
    t->scale = scale;
    s->textures.push_back(t);
    return (int)s->textures.size() - 1;
}
ORC_API int orc_mat_lambertian(Scene* s, int tex) {
    if (!ok_tex(s, tex)) return -1;
    auto m = std::make_shared<Lambertian>();
    m->texture = s->textures[tex];
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}
ORC_API int orc_mat_metal(Scene* s, double r, double g, double b, double fuzz) {
    auto m = std::make_shared<Metal>();
    m->albedo = V3{r, g, b};
    m->fuzz = fuzz;
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}
ORC_API int orc_mat_dielectric(Scene* s, double ri) {
    auto m = std::make_shared<Dielectric>();
    m->refraction_index = ri;
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}
ORC_API int orc_mat_diffuse_light(Scene* s, int tex) {
    if (!ok_tex(s, tex)) return -1;
    auto m = std::make_shared<DiffuseLight>();
    m->texture = s->textures[tex];
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}
ORC_API int orc_mat_isotropic(Scene* s, int tex) {
    if (!ok_tex(s, tex)) return -1;
    auto m = std::make_shared<Isotropic>();
    m->texture = s->textures[tex];
    s->materials.push_back(m);
    return (int)s->materials.size() - 1;
}
ORC_API int orc_sphere(Scene* s, const double* c, double r, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, std::make_shared<Sphere>(rd3(c), r, s->materials[mat]));
}
ORC_API int orc_quad(Scene* s, const double* q, const double* u, const double* v, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, std::make_shared<Parallelogram>(rd3(q), rd3(u), rd3(v), s->materials[mat]));
}
ORC_API int orc_triangle(Scene* s, const double* q, const double* u, const double* v, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, std::make_shared<Triangle>(rd3(q), rd3(u), rd3(v), s->materials[mat]));
}
ORC_API int orc_y_cylinder(Scene* s, const double* c, double r, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, y_cylinder(rd3(c), r, s->materials[mat]));
}
ORC_API int orc_quadric_sphere(Scene* s, const double* c, double r, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, quadric_sphere(rd3(c), r, s->materials[mat]));
}
ORC_API int orc_y_cone(Scene* s, const double* c, const double* off, int mat) {
    if (!ok_mat(s, mat)) return -1;
    return push_obj(s, y_cone(rd3(c), rd3(off), s->materials[mat]));
}
ORC_API int orc_list_new(Scene* s) {
    s->lists.emplace_back();
    return (int)s->lists.size() - 1;
}
ORC_API int orc_list_add(Scene* s, int list, int obj) {
    if (!ok_list(s, list) || !ok_obj(s, obj)) return -1;
    s->lists[list].add(s->objects[obj]);
    return 0;
}
// create_box, parallelogram.rs:146-195 -> a HittableList handle
ORC_API int orc_box(Scene* s, const double* a_, const double* b_, int mat) {
    if (!ok_mat(s, mat)) return -1;
    V3 a = rd3(a_), b = rd3(b_);
    auto m = s->materials[mat];
    HittableList sides;
    V3 mn{std::fmin(a.x, b.x), std::fmin(a.y, b.y), std::fmin(a.z, b.z)};
    V3 mx{std::fmax(a.x, b.x), std::fmax(a.y, b.y), std::fmax(a.z, b.z)};
    V3 dx{mx.x - mn.x, 0.0, 0.0}, dy{0.0, mx.y - mn.y, 0.0}, dz{0.0, 0.0, mx.z - mn.z};
    sides.add(std::make_shared<Parallelogram>(V3{mn.x, mn.y, mx.z}, dx, dy, m));
    sides.add(std::make_shared<Parallelogram>(V3{mx.x, mn.y, mx.z}, -dz, dy, m));
    sides.add(std::make_shared<Parallelogram>(V3{mx.x, mn.y, mn.z}, -dx, dy, m));
    sides.add(std::make_shared<Parallelogram>(V3{mn.x, mn.y, mn.z}, dz, dy, m));
    sides.add(std::make_shared<Parallelogram>(V3{mn.x, mx.y, mx.z}, dx, -dz, m));
    sides.add(std::make_shared<Parallelogram>(V3{mn.x, mn.y, mn.z}, dx, dz, m));
    s->lists.push_back(sides);
    return (int)s->lists.size() - 1;
}
ORC_API int orc_slice(Scene* s, int list) {
    if (!ok_list(s, list)) return -1;
    return push_obj(s, std::make_shared<HittableSlice>(s->lists[list]));
}
ORC_API int orc_bvh(Scene* s, int list) {
    if (!ok_list(s, list)) return -1;
    return push_obj(s, std::make_shared<BVHNode>(s->lists[list].objects));
}
ORC_API int orc_translate(Scene* s, int obj, const double* off) {
    if (!ok_obj(s, obj)) return -1;
    return push_obj(s, std::make_shared<Translate>(s->objects[obj], rd3(off)));
}
ORC_API int orc_rotate_y(Scene* s, int obj, double deg) {
    if (!ok_obj(s, obj)) return -1;
    return push_obj(s, std::make_shared<RotateY>(s->objects[obj], deg));
}
ORC_API int orc_constant_medium(Scene* s, int boundary, double density, int tex) {
    if (!ok_obj(s, boundary) || !ok_tex(s, tex)) return -1;
    return push_obj(s, std::make_shared<ConstantMedium>(s->objects[boundary], density, s->textures[tex]));
}
ORC_API int orc_load_model(Scene* s, const char* path, double scale, int mat) {
    if (!ok_mat(s, mat)) return -1;
    std::vector<HPtr> tris;
    if (!load_model(path, scale, s->materials[mat], tris, s->err)) return -1;
    return push_obj(s, std::make_shared<BVHNode>(tris));
}
ORC_API int orc_set_world(Scene* s, int obj) {
    if (!ok_obj(s, obj)) return -1;
    s->world = s->objects[obj];
    uint32_t next = 0;
    s->world->assign_ids(next);
    s->n_prims = next;
    return 0;
}
ORC_API uint32_t orc_num_prims(Scene* s) { return s->n_prims; }
// counts[8]: spheres, quads, triangles, quadrics, media, bvh nodes, translates, rotates (tree-expanded)
ORC_API void orc_counts(Scene* s, uint64_t* counts) {
    std::memset(counts, 0, 8 * sizeof(uint64_t));
    if (s->world) s->world->count(counts);
}
ORC_API int orc_bvh_depth(Scene* s, int obj) {
    if (!ok_obj(s, obj)) return -1;
    auto* b = dynamic_cast<BVHNode*>(s->objects[obj].get());
    return b ? b->depth() : 0;
}
// bounding box of an object handle: x.start,x.end,y.start,...
ORC_API int orc_bbox(Scene* s, int obj, double* out6) {
    if (!ok_obj(s, obj)) return -1;
    const AABB& b = s->objects[obj]->bbox;
    out6[0] = b.x.start; out6[1] = b.x.end; out6[2] = b.y.start; out6[3] = b.y.end; out6[4] = b.z.start; out6[5] = b.z.end;
    return 0;
}

// Camera ------------------------------------------------------------------------------------------
struct orc_camera {
    uint32_t width, height;
    double pixel00[3], du[3], dv[3], center[3];
    double defocus_angle;
    double disk_u[3], disk_v[3];
};
static void st3(double* o, V3 v) { o[0] = v.x; o[1] = v.y; o[2] = v.z; }
ORC_API void orc_camera_new(double aspect, uint32_t width, double vfov, double defocus_angle, double focus_dist,
                            const double* from, const double* at, const double* up, orc_camera* out) {
    Camera c = camera_new(aspect, width, vfov, defocus_angle, focus_dist, rd3(from), rd3(at), rd3(up));
    out->width = c.image_width;
    out->height = c.image_height;
    st3(out->pixel00, c.pixel00_loc);
    st3(out->du, c.pixel_delta_u);
    st3(out->dv, c.pixel_delta_v);
    st3(out->center, c.camera_center);
    out->defocus_angle = c.defocus_angle;
    st3(out->disk_u, c.defocus_disk_u);
    st3(out->disk_v, c.defocus_disk_v);
}
static Camera cam_from(const orc_camera* c) {
    Camera k;
    k.image_width = c->width;
    k.image_height = c->height;
    k.pixel00_loc = rd3(c->pixel00);
    k.pixel_delta_u = rd3(c->du);
    k.pixel_delta_v = rd3(c->dv);
    k.camera_center = rd3(c->center);
    k.defocus_angle = c->defocus_angle;
    k.defocus_disk_u = rd3(c->disk_u);
    k.defocus_disk_v = rd3(c->disk_v);
    return k;
}

// Hittable::hit on a batch of rays (hittable/mod.rs:65-68).  rng3 (nullable): per-ray (pixel, sample, first
// draw index); when null ray i uses (i, 0, 0).  Outputs may be null.
ORC_API int orc_intersect_batch(Scene* s, uint64_t seed, const double* o3, const double* d3, const double* tmin,
                                const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim,
                                double* t, double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out,
                                uint64_t* counters3) {
    if (!s->world) return -1;
    Counters cnt;
    for (uint64_t i = 0; i < n; ++i) {
        Rng rng(seed, rng3 ? rng3[3 * i] : (uint32_t)i, rng3 ? rng3[3 * i + 1] : 0u, 0u, rng3 ? rng3[3 * i + 2] : 0u);
        Ctx cx{&rng, &cnt};
        Ray ray = make_ray(rd3(o3 + 3 * i), rd3(d3 + 3 * i));
        HitRecord rec;
        cnt.rays++;
        bool h = s->world->hit(ray, Range{tmin[i], tmax[i]}, rec, cx);
        if (hit) hit[i] = h ? 1 : 0;
        if (prim) prim[i] = h ? rec.prim_id : 0xFFFFFFFFu;
        if (t) t[i] = h ? rec.t : 0.0;
        if (p3) st3(p3 + 3 * i, h ? rec.p : V3{0, 0, 0});
        if (n3) st3(n3 + 3 * i, h ? rec.normal : V3{0, 0, 0});
        if (uv2) { uv2[2 * i] = h ? rec.u : 0.0; uv2[2 * i + 1] = h ? rec.v : 0.0; }
        if (front) front[i] = (h && rec.front_face) ? 1 : 0;
        if (ctr_out) ctr_out[i] = rng.ctr;
    }
    if (counters3) { counters3[0] = cnt.rays; counters3[1] = cnt.node_tests; counters3[2] = cnt.prim_tests; }
    return 0;
}

struct orc_render_opts {
    uint32_t spp_begin, spp_count, max_depth, threads;
    uint64_t seed;
    double background[3];
};

// Per-sample radiance of chosen (pixel, sample) pairs: Camera::get_ray + ray_color (camera.rs:258-264).
// Optionally captures the ray entering bounce `capture_bounce` (0 = camera ray) with its rng draw index.
ORC_API int orc_trace_samples(Scene* s, const orc_camera* cam_, const orc_render_opts* o, uint64_t n,
                              const uint32_t* pixel, const uint32_t* sample, double* radiance3, uint32_t* nrays,
                              int capture_bounce, double* cap_o3, double* cap_d3, uint8_t* cap_valid,
                              uint32_t* cap_ctr) {
    if (!s->world) return -1;
    Camera cam = cam_from(cam_);
    V3 bg = rd3(o->background);
    for (uint64_t k = 0; k < n; ++k) {
        Counters cnt;
        Rng rng(o->seed, pixel[k], sample[k], 0u);
        Ctx cx{&rng, &cnt};
        uint32_t i = pixel[k] % cam.image_width, j = pixel[k] / cam.image_width;
        Ray r = get_ray(cam, i, j, rng);
        Ray cap = r;
        bool captured = false;
        uint32_t cctr = 0;
        V3 c = ray_color(r, o->max_depth, *s->world, bg, cx, capture_bounce >= 0 ? &cap : nullptr,
                         (uint32_t)(capture_bounce < 0 ? 0 : capture_bounce), 0, &captured, &cctr);
        if (radiance3) st3(radiance3 + 3 * k, c);
        if (nrays) nrays[k] = (uint32_t)cnt.rays;
        if (capture_bounce >= 0) {
            if (cap_o3) st3(cap_o3 + 3 * k, cap.origin);
            if (cap_d3) st3(cap_d3 + 3 * k, cap.direction);
            if (cap_valid) cap_valid[k] = captured ? 1 : 0;
            if (cap_ctr) cap_ctr[k] = cctr;
        }
    }
    return 0;
}

// Camera::render / thrender (camera.rs:217-271): film[j][i] = sum over samples (in sample order, starting from
// -0.0 as Point3's Sum does, point3/mod.rs:96-109).  The film holds sums, not averages.  A pool of threads pulls
// rows dynamically; this stands in for the Rayon pool and does not change any pixel's value.
ORC_API int orc_render(Scene* s, const orc_camera* cam_, const orc_render_opts* o, double* film3, uint64_t* counters3) {
    if (!s->world) return -1;
    Camera cam = cam_from(cam_);
    V3 bg = rd3(o->background);
    uint32_t W = cam.image_width, H = cam.image_height;
    uint32_t nthreads = o->threads ? o->threads : 1;
    std::atomic<uint32_t> next_row(0);
    std::vector<Counters> cnts(nthreads);
    auto work = [&](uint32_t tid) {
        Counters& cnt = cnts[tid];
        for (;;) {
            uint32_t j = next_row.fetch_add(1);
            if (j >= H) break;
            for (uint32_t i = 0; i < W; ++i) {
                V3 sum{-0.0, -0.0, -0.0};
                uint32_t pix = j * W + i;
                for (uint32_t k = 0; k < o->spp_count; ++k) {
                    Rng rng(o->seed, pix, o->spp_begin + k, 0u);
                    Ctx cx{&rng, &cnt};
                    Ray r = get_ray(cam, i, j, rng);
                    bool cap = false;
                    uint32_t cc = 0;
                    sum = sum + ray_color(r, o->max_depth, *s->world, bg, cx, nullptr, 0, 0, &cap, &cc);
                }
                st3(film3 + 3 * (size_t)pix, sum);
            }
        }
    };
    std::vector<std::thread> th;
    for (uint32_t t = 1; t < nthreads; ++t) th.emplace_back(work, t);
    work(0);
    for (auto& x : th) x.join();
    if (counters3) {
        counters3[0] = counters3[1] = counters3[2] = 0;
        for (auto& c : cnts) { counters3[0] += c.rays; counters3[1] += c.node_tests; counters3[2] += c.prim_tests; }
    }
    return 0;
}

// pixel_color / spp then proccess_color (camera.rs:265-267, color.rs:49-68)
ORC_API void orc_tonemap_rgb8(const double* film3, uint32_t w, uint32_t h, uint32_t spp, uint8_t* rgb8) {
    for (size_t p = 0; p < (size_t)w * h; ++p) {
        V3 c = rd3(film3 + 3 * p) / (double)spp;
        proccess_color(c, rgb8 + 3 * p);
    }
}

// ---- unit-level entry points for known-answer tests -------------------------------------------------
ORC_API void orc_sphere_uv(const double* p, double* uv) { get_sphere_uv(rd3(p), uv[0], uv[1]); }
ORC_API void orc_philox(const uint32_t* ctr4, const uint32_t* key2, uint32_t* out4) {
    uint32_t c[4] = {ctr4[0], ctr4[1], ctr4[2], ctr4[3]};
    philox4x32_10(c, key2[0], key2[1]);
    std::memcpy(out4, c, 16);
}
ORC_API double orc_rng_draw(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t stream, uint32_t ctr) {
    Rng r(seed, pixel, sample, stream, ctr);
    return r.next01();
}
// op: 0 add, 1 sub, 2 neg(a), 3 scale a*b[0], 4 cmul, 5 cross, 6 reflect, 7 unit(a), 8 rotate_y(a, cos=b[0], sin=b[1])
ORC_API void orc_vec_op(int op, const double* a_, const double* b_, double* out) {
    V3 a = rd3(a_), b = rd3(b_), r{0, 0, 0};
    switch (op) {
        case 0: r = a + b; break;
        case 1: r = a - b; break;
        case 2: r = -a; break;
        case 3: r = a * b.x; break;
        case 4: r = a * b; break;
        case 5: r = cross(a, b); break;
        case 6: r = reflect(a, b); break;
        case 7: r = unit_vector(a); break;
        case 8: r = rotate_y(a, b.x, b.y); break;
    }
    st3(out, r);
}
ORC_API double orc_dot(const double* a, const double* b) { return dot(rd3(a), rd3(b)); }
ORC_API void orc_refract(const double* uv, const double* n, double eta, double* out) { st3(out, refract(rd3(uv), rd3(n), eta)); }
ORC_API double orc_reflectance(double c, double ri) { return reflectance(c, ri); }
ORC_API void orc_process_color(const double* c, uint8_t* out) { proccess_color(rd3(c), out); }
// AABB::hit on one box: returns 1/0 and the narrowed interval.
ORC_API int orc_aabb_hit(const double* box6, const double* o, const double* d, double* range2) {
    AABB b{Range{box6[0], box6[1]}, Range{box6[2], box6[3]}, Range{box6[4], box6[5]}};
    Ray r = make_ray(rd3(o), rd3(d));
    Range rt{range2[0], range2[1]};
    bool h = b.hit(r, rt);
    range2[0] = rt.start;
    range2[1] = rt.end;
    return h ? 1 : 0;
}
// Evaluate a texture handle at (u, v, p).
ORC_API int orc_tex_value(Scene* s, int tex, double u, double v, const double* p, double* out) {
    if (!ok_tex(s, tex)) return -1;
    st3(out, s->textures[tex]->value(u, v, rd3(p)));
    return 0;
}
// Material::scatter of a material handle on a hand-made hit record.  Returns 1 if scattered.
ORC_API int orc_mat_scatter(Scene* s, int mat, uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t ctr,
                            const double* ray_o, const double* ray_d, const double* p, const double* normal, double u,
                            double v, int front, double* out_o, double* out_d, double* att, double* emitted,
                            uint32_t* ctr_out) {
    if (!ok_mat(s, mat)) return -1;
    Rng rng(seed, pixel, sample, 0u, ctr);
    HitRecord rec;
    rec.p = rd3(p);
    rec.normal = rd3(normal);
    rec.u = u;
    rec.v = v;
    rec.front_face = front != 0;
    Ray in = make_ray(rd3(ray_o), rd3(ray_d));
    Ray out = make_ray(V3{0, 0, 0}, V3{1, 1, 1});
    V3 a{1, 1, 1};
    bool sc = s->materials[mat]->scatter(in, rec, out, a, rng);
    st3(emitted, s->materials[mat]->emitted(u, v, rec.p));
    st3(out_o, out.origin);
    st3(out_d, out.direction);
    st3(att, a);
    if (ctr_out) *ctr_out = rng.ctr;
    return sc ? 1 : 0;
}
