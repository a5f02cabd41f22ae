// rt_amd.hip -- kernels and C ABI of librt_amd.so (see include/rt_amd.h).
//
// Wavefront path tracer for gfx950: one path per lane, path state in SoA queues in HBM, per bounce one
// traverse+intersect launch and one shade+compact launch.  Build with -ffp-contract=off.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>  // types only: the library is loaded at run time (rccl())
#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <map>
#include <type_traits>
#include <mutex>
#include <string>
#include <thread>
#include <numeric>
#include <vector>

#include "rt_core.h"
#include "rt_records.h"
#include "rt_wide.h"

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

// Every device allocation of the library goes through here and is counted (rt_debug_device_allocs: the tests of the
// entry point that must not allocate read the count before and after a call).
static std::atomic<uint64_t> g_device_allocs{0};
static hipError_t rt_hip_malloc(void** p, size_t bytes) {
    g_device_allocs.fetch_add(1);
    return hipMalloc(p, bytes);
}
template <class T>
static hipError_t rt_hip_malloc(T** p, size_t bytes) { return rt_hip_malloc(reinterpret_cast<void**>(p), bytes); }

// ------------------------------------------------------------------------------------------------ kernels
#define TRAV_BLOCK 64    // one wavefront per workgroup: no barriers, LDS interval slots per lane
#define SHADE_BLOCK 256

// Path state queue, SoA: consecutive lanes read consecutive elements of every array.
struct Queue {
    double *ox, *oy, *oz, *dx, *dy, *dz;  // ray
    double *br, *bg, *bb;                 // throughput (product of attenuations so far)
    uint32_t* pix;                        // pixel index j * width + i
    uint32_t* smp;                        // global sample index (with pix: the path's random stream)
    uint32_t* slot;                       // where the path's radiance goes in the luminance buffer
    uint32_t* born;                       // iteration the path was generated in (depth = iteration - born)
    uint32_t* ctr;                        // next draw index of the path's random stream
    uint32_t* sig;                        // ray_signature of the ray (written by k_raygen and k_bounce_lockstep only)
};
// Closest hit per queue slot, one 16-byte record each: t (f64), reference, chain -- written and read whole by one store /
// load per lane (the walks' gathers and scattered stores are what this code pays for), t alone by an 8-byte one.
struct HitBuf {
    uint4* rec;
};
__device__ __forceinline__ void hb_put(const HitBuf& hb, size_t i, double t, uint32_t ref, uint32_t chain) {
    const unsigned long long bits = (unsigned long long)__double_as_longlong(t);
    uint4 v; v.x = (uint32_t)bits; v.y = (uint32_t)(bits >> 32); v.z = ref; v.w = chain;
    hb.rec[i] = v;
}
__device__ __forceinline__ Best hb_get(const HitBuf& hb, size_t i) {
    const uint4 v = hb.rec[i];
    Best b; b.t = __longlong_as_double((long long)((unsigned long long)v.y << 32 | v.x)); b.ref = v.z; b.chain = v.w;
    return b;
}
__device__ __forceinline__ double hb_t(const HitBuf& hb, size_t i) { return reinterpret_cast<const double*>(hb.rec)[2u * i]; }
__device__ __forceinline__ void hb_set_t(const HitBuf& hb, size_t i, double t) { reinterpret_cast<double*>(hb.rec)[2u * i] = t; }

extern __shared__ double2 g_levels[];

// Work lists of the gated launches (trees with BVH entries): list b holds the queue slots of the rays whose gate bit b
// is set, in no particular order; its length is counted in word RT_COUNT_STRIDE * (4 + b) of a counter set.
#define RT_COUNT_STRIDE 64u  // counters sit 256 bytes apart: their atomics go to different cache lines
struct Gates {
    const GateInfo* info;  // what the bits are made of
    uint32_t* lists;       // list b starts at lists + b * stride
    uint32_t stride, n;    // n = 0: no gates
    uint32_t init_hits;    // the kernel that makes a ray also writes "no hit" for it (0: the first launch of the
                           // traverse phase visits every ray and starts from nothing by itself)
    uint32_t tag;          // 1: the ray's gate bits travel in the high bits of its `born` word, and the kernel that makes
                           // a ray writes t = infinity ("no hit so far") for it if it is in any list at all: every
                           // launch that walks a list goes on from there and writes a hit only when it finds one
};
#define RT_BORN_BITS 20u           // with Gates::tag the iteration a path was generated in takes the low 20 bits of `born`,
#define RT_BORN_MASK 0xFFFFFu      // the gate bits of its current ray (RT_MAX_GATES <= 12) the rest
// Appends slot `dst` of every `alive` thread to the lists its ray (o, d) belongs in: ballot + prefix count per wave and
// list, one atomic per workgroup and list.  Called by all threads of a 256-thread workgroup together.
__device__ __forceinline__ uint32_t gates_of(const Gates& G, bool alive, V3 o, V3 d) {
    return (G.n != 0u && alive) ? ray_gates(*G.info, o, d) : 0u;
}
__device__ __forceinline__ void gates_append(const Gates& G, uint32_t bits, uint32_t dst, uint32_t* counts) {
    __shared__ uint32_t g_wave[4][RT_MAX_GATES];
    __shared__ uint32_t g_base[RT_MAX_GATES];
    if (G.n == 0u) return;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    for (uint32_t b = 0; b < G.n; ++b) {
        const unsigned long long mask = __ballot((bits >> b) & 1u);
        if (lane == 0) g_wave[wave][b] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < G.n) {
        uint32_t tot = 0;
        for (int k = 0; k < 4; ++k) { const uint32_t c = g_wave[k][threadIdx.x]; g_wave[k][threadIdx.x] = tot; tot += c; }
        g_base[threadIdx.x] = tot ? atomicAdd(counts + RT_COUNT_STRIDE * (4u + threadIdx.x), tot) : 0u;
    }
    __syncthreads();
    for (uint32_t b = 0; b < G.n; ++b) {
        const unsigned long long mask = __ballot((bits >> b) & 1u);
        if ((bits >> b) & 1u)
            G.lists[(size_t)b * G.stride + g_base[b] + g_wave[wave][b] + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull))] =
                dst;
    }
}

// Camera::get_ray (camera.rs:479-492) for one batch of `nsamples` consecutive samples of every pixel, appended to
// the queue at `base`.  Batch-local path j = s_local * npix + pixel keeps primary rays of neighbouring pixels in
// neighbouring lanes.  With gates the new rays also go into their work lists and start from "no hit" in `hb`.
__global__ void __launch_bounds__(256) k_raygen(const SigInfo* __restrict__ Gp, rt_camera cam, uint64_t seed, uint32_t npix, uint32_t sample0,
                                                 uint32_t n, uint32_t base, uint32_t slot0, uint32_t iter, Queue q,
                                                 Gates gates, HitBuf hb, uint32_t* counts) {
    uint32_t j = blockIdx.x * blockDim.x + threadIdx.x;
    const bool alive = j < n;
    V3 o = mk(0, 0, 0), d = mk(1, 1, 1);
    const uint32_t idx = base + j;
    if (alive) {
        uint32_t pixel = j % npix, s_local = j / npix;
        Rng rng = rng_make(seed, pixel, sample0 + s_local, 0u);
        camera_ray(cam, pixel % cam.width, pixel / cam.width, rng, o, d);
        q.ox[idx] = o.x; q.oy[idx] = o.y; q.oz[idx] = o.z;
        q.dx[idx] = d.x; q.dy[idx] = d.y; q.dz[idx] = d.z;
        q.br[idx] = 1.0; q.bg[idx] = 1.0; q.bb[idx] = 1.0;
        q.pix[idx] = pixel;
        q.smp[idx] = sample0 + s_local;
        q.slot[idx] = slot0 + j;
        q.ctr[idx] = rng.ctr;
        if (Gp->n != 0u) q.sig[idx] = ray_signature(*Gp, o, d);
        if (gates.n && gates.init_hits) hb_put(hb, idx, RT_INF, RT_NO_HIT, 0u);
    }
    const uint32_t bits = gates_of(gates, alive, o, d);
    if (alive) q.born[idx] = iter | (gates.tag ? bits << RT_BORN_BITS : 0u);
    if (gates.tag && bits != 0u) hb_set_t(hb, idx, RT_INF);
    gates_append(gates, bits, idx, counts);
}

// Where the rays of one launch are.  The queue is a row of regions of `stride` slots: one per signature class (a single
// one for trees with BVH entries), filled from slot 0 of the region by the launch before, then one for the camera
// rays generated since.  How many rays wait in each class region is read from device memory (`in_count`, counters
// RT_COUNT_STRIDE apart, written by the launch before as its `out_count`), so that the host can queue launches
// without waiting for those numbers; `fresh` it knows.  Every stretch starts a new workgroup (lockstep kernels).
// in_count = null: one stretch, slots 0 .. n - 1 (caller-supplied batches).
#define RT_RING 4            // counter sets in rotation: the host queues launches RT_LAG iterations ahead of the counts it has read
#define RT_LAG 2
#define RT_RING_WORDS ((5u + RT_MAX_GATES) * RT_COUNT_STRIDE)  // class counters, work-list lengths, hits left to the second shade launch
#define RT_DEFER_WORD ((4u + RT_MAX_GATES) * RT_COUNT_STRIDE)
struct Segs {
    const uint32_t* __restrict__ in_count;
    uint32_t n;
    uint32_t nclass, stride;
    uint32_t fresh;
};
// Queue slot of this thread, or `end` (one past the last slot of its stretch) when it has none.
__device__ __forceinline__ uint32_t seg_slot(const Segs& segs, uint32_t& end) {
    const uint32_t bs = blockDim.x;
    if (segs.in_count == nullptr) {
        end = segs.n;
        const uint32_t idx = blockIdx.x * bs + threadIdx.x;
        return idx < segs.n ? idx : segs.n;
    }
    uint32_t base = 0, count = 0, block0 = 0, next = 0;
#pragma unroll
    for (uint32_t k = 0; k < 5u; ++k) {  // classes 0..3, then the fresh rays
        if (k < 4u && k >= segs.nclass) continue;
        const uint32_t c = k < 4u ? segs.in_count[RT_COUNT_STRIDE * k] : segs.fresh;
        const uint32_t r = k < 4u ? k : segs.nclass;
        if (blockIdx.x >= next) { base = r * segs.stride; count = c; block0 = next; }
        next += (c + bs - 1u) / bs;
    }
    if (blockIdx.x >= next) count = 0;  // the host sized the grid for more rays than there are
    const uint32_t local = (blockIdx.x - block0) * bs + threadIdx.x;
    end = base + count;
    return local < count ? base + local : end;
}
// The same rays as one numbered sequence 0 .. seg_total - 1 (the kernel that hands rays out; one class).
__device__ __forceinline__ uint32_t seg_total(const Segs& segs) {
    return segs.in_count == nullptr ? segs.n : segs.in_count[0] + segs.fresh;
}
__device__ __forceinline__ uint32_t seg_nth(const Segs& segs, uint32_t v) {
    if (segs.in_count == nullptr) return v;
    const uint32_t c0 = segs.in_count[0];
    return v < c0 ? v : segs.stride + (v - c0);
}

// The same over every class region (the tail kernel, whatever the tree).
__device__ __forceinline__ uint32_t seg_total_all(const Segs& segs) {
    if (segs.in_count == nullptr) return segs.n;
    uint32_t total = segs.fresh;
    for (uint32_t k = 0; k < segs.nclass; ++k) total += segs.in_count[RT_COUNT_STRIDE * k];
    return total;
}
__device__ __forceinline__ uint32_t seg_nth_all(const Segs& segs, uint32_t v) {
    if (segs.in_count == nullptr) return v;
    for (uint32_t k = 0; k < segs.nclass; ++k) {
        const uint32_t c = segs.in_count[RT_COUNT_STRIDE * k];
        if (v < c) return k * segs.stride + v;
        v -= c;
    }
    return segs.nclass * segs.stride + v;
}

// What a traverse launch works on: the queue of rays, where the closest hits go, and -- for caller-supplied batches
// only -- per-ray intervals (a render uses 0.001..inf, camera.rs:363).
struct TravArgs {
    Queue q;
    HitBuf hb;
    uint64_t seed;
    const double* tmin;     // nullable
    const double* tmax;     // nullable
    uint32_t chunk;         // rays a wavefront of the persistent kernel takes per hand-out
    Segs segs;              // where the rays are
    uint32_t first, end;    // the stretch of top-level entries this launch walks (the whole tree: 0, n_nodes)
    uint32_t resume;        // 1: go on from the closest hit in `hb` (what the stretches before this one have left)
    const uint32_t* wl;       // gated launch: the queue slots to work on and where their number is; null: every ray
    const uint32_t* wl_count;
    int thr_fetch, thr_step;  // wide kernel: lanes that must be waiting before rays are handed out / a step is taken
    uint32_t fast;            // wide kernel: children nearest first (RT_FLAG_FAST_BVH)
    unsigned long long* list_total;  // wide kernel, nullable: += the length of the work list this launch went through
};
__device__ __forceinline__ Best trav_best_setup(const TravArgs& a, uint32_t idx) {
    Best b; b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0;
    if (a.resume) b = hb_get(a.hb, idx);
    return b;
}
__device__ __forceinline__ void trav_ray_setup(const TravArgs& a, uint32_t idx, RngPos& rng, double& tmin, double& tmax) {
    rng.k0 = (uint32_t)a.seed; rng.k1 = (uint32_t)(a.seed >> 32);
    rng.pixel = a.q.pix[idx]; rng.sample = a.q.smp[idx];
    rng.ctr = a.q.ctr[idx];
    tmin = a.tmin ? a.tmin[idx] : 0.001;
    tmax = a.tmax ? a.tmax[idx] : RT_INF;
}
__device__ __forceinline__ void tally_flush(const Tally& tally, unsigned long long* tallies) {
    unsigned long long v[6] = {tally.node_tests, tally.prim_tests, tally.prim_bytes, tally.cull_tests, tally.prim_flops, tally.wide_tests};
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        for (int off = 32; off > 0; off >>= 1) v[k] += __shfl_down(v[k], off);
        if ((threadIdx.x & 63) == 0) atomicAdd(&tallies[k], v[k]);
    }
}

// world.hit(ray, 0.001..inf) for every queued path (camera.rs:363), for trees WITHOUT BVH entries: one lane per
// ray, all lanes visit the same entries in the same order, scene data comes through scalar loads.
template <bool COUNT, int MEDIA>
__global__ void __launch_bounds__(TRAV_BLOCK, MEDIA == RT_MEDIA_ANY ? 3 : 4) k_traverse_lockstep(DScene S, TravArgs a, unsigned long long* tallies) {
    uint32_t n;
    const uint32_t idx = seg_slot(a.segs, n);
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    // with sorted queues the wavefront walks the copy of the tree without everything its rays' signatures rule out
    uint32_t first = a.first;
    if (a.segs.in_count != nullptr && S.n_variants > 1u) first += tree_without(S, sig_skip_mask(idx < n ? a.q.sig[idx] : 0u));
    if (idx < n) {
        RngPos rng;
        double tmin, tmax;
        trav_ray_setup(a, idx, rng, tmin, tmax);
        RaySoA src;
        src.ox = a.q.ox; src.oy = a.q.oy; src.oz = a.q.oz; src.dx = a.q.dx; src.dy = a.q.dy; src.dz = a.q.dz; src.idx = idx;
        Best b = closest_hit_range<COUNT, true, MEDIA>(S, src, tmin, tmax, L, rng, tally, first, first + (a.end - a.first),
                                                       trav_best_setup(a, idx));
        hb_put(a.hb, idx, b.t, b.ref, b.chain);
        if (MEDIA != RT_MEDIA_NONE) a.q.ctr[idx] = rng.ctr;  // (only a medium draws during the walk)
    }
    if (COUNT) tally_flush(tally, tallies);
}

// The same for trees WITH BVH entries, where rays of one wavefront need very different numbers of steps: a fixed
// grid of wavefronts pulls rays from the queue; each lane runs the per-lane state machine (one tree entry and one
// primitive test per iteration) and takes the next ray the moment its own is finished, so lanes do not idle
// waiting for the slowest ray.  Rays are handed out in chunks of RT_CHUNK consecutive queue slots per wavefront
// (one atomic per chunk).  Every wavefront leaves the loop once the queue is exhausted and its lanes are done.
#define RT_CHUNK 256u
#define RT_STEP_THRESHOLD 20
#define RT_FETCH_THRESHOLD 16
#define RT_MEDIUM_THRESHOLD 18
#define RT_CHAIN_THRESHOLD 12
#define RT_SUB_THRESHOLD 12
template <bool COUNT, int MEDIA>
__global__ void __launch_bounds__(TRAV_BLOCK, MEDIA == RT_MEDIA_ANY ? 3 : 4) k_traverse_persistent(DScene S, TravArgs a, uint32_t* work,
                                                                         unsigned long long* tallies) {
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint32_t idx = 0xFFFFFFFFu;
    const uint32_t total = seg_total(a.segs);
    Lane ln;
    ln.cur = ray_world(mk(0, 0, 0), mk(1, 1, 1));
    ln.cur_chain = 0; ln.i = 0; ln.k = 0; ln.kend = 0; ln.kind = 0; ln.s = 0; ln.e = 0;
    ln.best.t = RT_INF; ln.best.ref = RT_NO_HIT; ln.best.chain = 0;
    ln.state = RT_ST_DONE;
    RngPos rng; rng.k0 = rng.k1 = rng.pixel = rng.sample = rng.ctr = 0;
    RaySoA src;
    src.ox = a.q.ox; src.oy = a.q.oy; src.oz = a.q.oz; src.dx = a.q.dx; src.dy = a.q.dy; src.dz = a.q.dz; src.idx = 0;
    for (;;) {
        const bool need = ln.state == RT_ST_DONE;
        const unsigned long long mask = __ballot(need);
        int n_node = __popcll(__ballot(ln.state == RT_ST_NODE));
        int n_run = __popcll(__ballot(ln.state == RT_ST_RUN));
        int n_med = MEDIA != RT_MEDIA_NONE ? __popcll(__ballot(ln.state == RT_ST_MEDIUM)) : 0;
        const int n_chain = __popcll(__ballot(ln.state == RT_ST_CHAIN));
        const int n_sub = __popcll(__ballot(ln.state == RT_ST_SUB));
        // hand out rays only when enough lanes are free (or nobody has work): the hand-out code costs the whole
        // wavefront its instructions however few lanes it serves
        if (mask != 0ull && (__popcll(mask) >= RT_FETCH_THRESHOLD || n_node + n_run + n_med + n_chain + n_sub == 0)) {
            if (need && idx != 0xFFFFFFFFu) {
                hb_put(a.hb, idx, ln.best.t, ln.best.ref, ln.best.chain);
                a.q.ctr[idx] = rng.ctr;
                idx = 0xFFFFFFFFu;
            }
            if (!exhausted) {
                if (pool_next == pool_end) {
                    const int leader = __ffsll((long long)mask) - 1;
                    uint32_t base = 0;
                    if ((int)lane == leader) base = atomicAdd(work, a.chunk);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                    pool_next = base < total ? base : total;
                    pool_end = (base + a.chunk) < total ? (base + a.chunk) : total;
                    if (base >= total) { pool_end = pool_next; exhausted = true; }
                }
                if (!exhausted) {
                    if (need) {
                        const uint32_t mine = pool_next + (uint32_t)__popcll(mask & lt_mask);
                        if (mine < pool_end) {
                            idx = seg_nth(a.segs, mine);
                            src.idx = idx;
                            double tmin, tmax;
                            trav_ray_setup(a, idx, rng, tmin, tmax);
                            lane_begin(ln, src, tmin, tmax, L, a.first);
                            ln.best = trav_best_setup(a, idx);
                        }
                    }
                    const uint32_t want = pool_next + (uint32_t)__popcll(mask);
                    pool_next = want < pool_end ? want : pool_end;
                }
            }
            if (exhausted && __ballot(ln.state != RT_ST_DONE) == 0ull) break;
            n_node = __popcll(__ballot(ln.state == RT_ST_NODE));
            n_run = __popcll(__ballot(ln.state == RT_ST_RUN));
        }
        // Likewise a step runs only when enough lanes are waiting for it, or when it is what most lanes want; a
        // medium, the most expensive step by far, when enough lanes have gathered or nothing else can move.
        const bool do_node = n_node >= RT_STEP_THRESHOLD || n_node >= n_run;
        const bool do_run = n_run >= RT_STEP_THRESHOLD || n_run > n_node;
        const bool do_med = MEDIA != RT_MEDIA_NONE && (n_med >= RT_MEDIUM_THRESHOLD || (n_med > 0 && n_node + n_run == 0));
        const bool do_chain = n_chain >= RT_CHAIN_THRESHOLD || (n_chain > 0 && n_node + n_run == 0);
        const bool do_sub = n_sub >= RT_SUB_THRESHOLD || (n_sub > 0 && n_node + n_run == 0);
        if (do_chain && ln.state == RT_ST_CHAIN) lane_chain_step(S, src, ln);
        if (do_sub && ln.state == RT_ST_SUB) lane_sub_step<COUNT>(S, ln, tally);
        if (do_node && ln.state == RT_ST_NODE) lane_node_step<COUNT, MEDIA>(S, src, ln, L, rng, tally);
        if (do_run && ln.state == RT_ST_RUN) lane_prim_step<COUNT>(S, ln, tally);
        if (MEDIA != RT_MEDIA_NONE) {
            if (do_med && ln.state == RT_ST_MEDIUM) lane_medium_step<COUNT, MEDIA == RT_MEDIA_ANY>(S, src, ln, L, rng, tally);
        }
    }
    if (COUNT) tally_flush(tally, tallies);
}

// One top-level BVH entry whose leaves the eight-wide tree of rt_wide.h covers: the same scheme as above -- a fixed grid
// of wavefronts pulls rays, every lane takes a new ray the moment its own is done -- with two kinds of step only, a
// node of the wide tree (eight child boxes, for a leaf's node the reference's narrowing first) or the next primitives of
// an item.  The ray is taken into the tree's instance chain where it is handed out, by all lanes that take one
// together; no random numbers are drawn here (no medium below such an entry).
#define RT_WIDE_STEP_THRESHOLD 24
#define RT_WIDE_RAYS_PER_LANE 6u  // of a tree's list, from which its launch has every wavefront the device holds
#define RT_WIDE_FETCH_THRESHOLD 16
// PROF: also adds up, per wavefront, the clock cycles spent handing out rays / in node steps / in primitive steps, the
// number of each and the lanes that took part (development aid: RT_WIDE_PROF=1 prints the sums after a render).
// What the primitive tests of a lane do not touch -- the reciprocal direction, the grid coefficients, the interval at
// the top level, the cursor -- stays in LDS while they run: they are the kernel's register peak, and without these 25
// registers a fourth wavefront fits each SIMD.  A node step brings everything back and stores what it changed.
struct WideAway {
    float (*f)[64];       // [6]: ax .. bz
    double (*d)[64];      // [5]: 1 / d, tmin, tmax
    uint32_t (*u)[64];    // [9]: pending, rev, slot, level, item_first, item_kind, item_counts
    uint32_t lane;
    __device__ __forceinline__ void fetch_ray(WLane& ln) const {
        ln.cur.inv = mk(d[0][lane], d[1][lane], d[2][lane]);
        ln.tmin = d[3][lane]; ln.tmax = d[4][lane];
    }
    __device__ __forceinline__ void fetch_cursor(WLane& ln) const {
        ln.pending = (unsigned long long)u[1][lane] << 32 | u[0][lane];
        ln.rev = u[2][lane]; ln.slot = u[3][lane]; ln.level = u[4][lane];
        ln.item_first = u[5][lane]; ln.item_kind = u[6][lane];
        ln.item_counts = (unsigned long long)u[8][lane] << 32 | u[7][lane];
    }
    __device__ __forceinline__ void stash_cursor(const WLane& ln) const {
        u[0][lane] = (uint32_t)ln.pending; u[1][lane] = (uint32_t)(ln.pending >> 32);
        u[2][lane] = ln.rev; u[3][lane] = ln.slot; u[4][lane] = ln.level;
        u[5][lane] = ln.item_first; u[6][lane] = ln.item_kind;
        u[7][lane] = (uint32_t)ln.item_counts; u[8][lane] = (uint32_t)(ln.item_counts >> 32);
    }
    __device__ __forceinline__ void fetch_grid(WLane& ln) const {
        ln.ax = f[0][lane]; ln.ay = f[1][lane]; ln.az = f[2][lane]; ln.bx = f[3][lane]; ln.by = f[4][lane]; ln.bz = f[5][lane];
    }
    __device__ __forceinline__ void fetch_all(WLane& ln) const {
        fetch_ray(ln);
        fetch_cursor(ln);
        fetch_grid(ln);
    }
    __device__ __forceinline__ void stash_all(const WLane& ln) const {
        stash_cursor(ln);
        d[0][lane] = ln.cur.inv.x; d[1][lane] = ln.cur.inv.y; d[2][lane] = ln.cur.inv.z; d[3][lane] = ln.tmin; d[4][lane] = ln.tmax;
        f[0][lane] = ln.ax; f[1][lane] = ln.ay; f[2][lane] = ln.az; f[3][lane] = ln.bx; f[4][lane] = ln.by; f[5][lane] = ln.bz;
    }
};
// The same for a walk that has the registers to keep all but the reciprocal direction (the triangle walk: 104 without,
// 113 with): no LDS traffic in a node step at all (cornell_model's walk - 2 %).
struct WideAwayInv {
    double (*d)[64];      // [3]: 1 / d
    uint32_t lane;
    __device__ __forceinline__ void fetch_ray(WLane& ln) const { ln.cur.inv = mk(d[0][lane], d[1][lane], d[2][lane]); }
    __device__ __forceinline__ void fetch_cursor(WLane&) const {}
    __device__ __forceinline__ void stash_cursor(const WLane&) const {}
    __device__ __forceinline__ void fetch_grid(WLane&) const {}
    __device__ __forceinline__ void fetch_all(WLane&) const {}
    __device__ __forceinline__ void stash_all(const WLane& ln) const { d[0][lane] = ln.cur.inv.x; d[1][lane] = ln.cur.inv.y; d[2][lane] = ln.cur.inv.z; }
};
#define RT_REF_CARRIED 0xFFFFFFFEu  // "the closest hit so far is the one the hit buffer already holds" (no such kind)
// COUNT: also tallies the tests done (counting passes; the timed launches are the instantiation without it).
// ONLY: the one kind of primitive the tree's items hold (its tests alone are compiled in), or 0 for any mix.
template <bool PROF, bool COUNT = false, int ONLY = 0>
__global__ void __launch_bounds__(TRAV_BLOCK, COUNT ? 2 : 4) k_traverse_wide(DScene S, DWide W, TravArgs a, uint32_t* work, unsigned long long* prof,
                                                                            unsigned long long* tallies) {
    static_assert(TRAV_BLOCK == 64, "one wavefront per workgroup: a column of the arrays below per lane");
    __shared__ float away_f[6][64];
    __shared__ double away_d[5][64];
    __shared__ uint32_t away_u[9][64];
    using AwayT = typename std::conditional<ONLY == RT_NODE_TRIANGLES && !COUNT && !PROF, WideAwayInv, WideAway>::type;
    AwayT away;
    if constexpr (std::is_same<AwayT, WideAway>::value) { away.f = away_f; away.d = away_d; away.u = away_u; }
    else away.d = away_d;
    away.lane = threadIdx.x & 63u;
    unsigned long long pc[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    const unsigned long long t_begin = PROF ? clock64() : 0ull;
    Tally tally = tally_zero();
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    uint32_t idx = 0xFFFFFFFFu;
    const uint32_t total = a.wl ? *a.wl_count : seg_total(a.segs);
    if (a.wl && a.list_total && blockIdx.x == 0u && lane == 0u) atomicAdd(a.list_total, (unsigned long long)total);
    // Rays per hand-out: the host's figure is sized by the rays in the queue; the list this launch works through may be
    // far shorter, and its rays are then spread over all the wavefronts there are, down to sixteen each.
    const uint32_t share = (total / gridDim.x + 15u) & ~15u;
    const uint32_t chunk = share >= a.chunk ? a.chunk : (share < 16u ? 16u : share);
    uint32_t seen = 0u;  // how far the list had been handed out when this wavefront last took from it
    WLane ln;
    {
        Best none; none.t = RT_INF; none.ref = RT_NO_HIT; none.chain = 0;
        wide_begin(ln, S, W, ray_world(mk(0, 0, 0), mk(1, 1, 1)), 0.0, 0.0, none);
        ln.state = RT_W_DONE;
    }
    for (;;) {
        const bool need = ln.state == RT_W_DONE;
        const unsigned long long mask = __ballot(need);
        int n_node = __popcll(__ballot(ln.state == RT_W_NODE));
        int n_prim = __popcll(__ballot(ln.state == RT_W_PRIM));
        if (mask != 0ull && (__popcll(mask) >= a.thr_fetch || n_node + n_prim == 0)) {
            const unsigned long long t0 = PROF ? clock64() : 0ull;
            if (PROF) { pc[3]++; pc[6] += (unsigned long long)__popcll(mask); }
            // The finished ray's result is stored after the next ray has been read: loads and stores of a wavefront
            // complete in the order issued, and the new ray's loads would otherwise wait for the stores' round trip too.
            const uint32_t done_idx = need ? idx : 0xFFFFFFFFu;
            const Best done_best = ln.best;
            if (need) idx = 0xFFFFFFFFu;
            if (!exhausted) {
                if (pool_next == pool_end) {
                    // shorter hand-outs as the list runs out (half of an even share of what was left at this wavefront's
                    // last one), so that the launch does not end on a few wavefronts working through long ones
                    const uint32_t left = total > seen ? total - seen : 0u;
                    const uint32_t even = (left / gridDim.x / 2u + 15u) & ~15u;
                    const uint32_t take = even >= chunk ? chunk : (even < 16u ? 16u : even);
                    const int leader = __ffsll((long long)mask) - 1;
                    uint32_t base = 0;
                    if ((int)lane == leader) base = atomicAdd(work, take);
                    base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                    seen = base + take;
                    pool_next = base < total ? base : total;
                    pool_end = (base + take) < total ? (base + take) : total;
                    if (base >= total) { pool_end = pool_next; exhausted = true; }
                }
                if (!exhausted) {
                    if (need) {
                        const uint32_t mine = pool_next + (uint32_t)__popcll(mask & lt_mask);
                        if (mine < pool_end) {
                            idx = a.wl ? a.wl[mine] : seg_nth(a.segs, mine);
                            const bool carried = a.resume != 0u;
                            const double tmin = a.tmin ? a.tmin[idx] : 0.001, tmax = a.tmax ? a.tmax[idx] : RT_INF;
                            const Ray w = ray_world(mk(a.q.ox[idx], a.q.oy[idx], a.q.oz[idx]), mk(a.q.dx[idx], a.q.dy[idx], a.q.dz[idx]));
                            // going on from an earlier launch's closest hits: only t of that hit is read, and the
                            // hit is written back only if the tree holds a closer one (the buffer keeps what it has)
                            Best b0; b0.t = carried ? hb_t(a.hb, idx) : RT_INF; b0.ref = carried ? RT_REF_CARRIED : RT_NO_HIT; b0.chain = 0u;
                            wide_begin(ln, S, W, w, tmin, tmax, b0);
                            away.stash_all(ln);
                        }
                    }
                    const uint32_t want = pool_next + (uint32_t)__popcll(mask);
                    pool_next = want < pool_end ? want : pool_end;
                }
            }
            if (done_idx != 0xFFFFFFFFu && done_best.ref != RT_REF_CARRIED) {
                hb_put(a.hb, done_idx, done_best.t, done_best.ref, done_best.chain);
            }
            if (exhausted && __ballot(ln.state != RT_W_DONE) == 0ull) break;
            n_node = __popcll(__ballot(ln.state == RT_W_NODE));
            n_prim = __popcll(__ballot(ln.state == RT_W_PRIM));
            if (PROF) { __builtin_amdgcn_s_waitcnt(0); pc[0] += clock64() - t0; }
        }
        const bool do_node = n_node >= a.thr_step || n_node >= n_prim;
        const bool do_prim = n_prim >= a.thr_step || n_prim > n_node;
        if (do_node) {
            const unsigned long long t0 = PROF ? clock64() : 0ull;
            if (ln.state == RT_W_NODE) { away.fetch_all(ln); wide_node_step(W, ln, COUNT ? &tally : nullptr); away.stash_cursor(ln); }
            if (PROF) { __builtin_amdgcn_s_waitcnt(0); pc[1] += clock64() - t0; pc[4]++; pc[7] += (unsigned long long)n_node; }
        }
        if (do_prim) {
            const unsigned long long t0 = PROF ? clock64() : 0ull;
            if (ln.state == RT_W_PRIM) wide_prim_step<COUNT, AwayT, ONLY>(S, W, ln, tally, away);
            if (PROF) { __builtin_amdgcn_s_waitcnt(0); pc[2] += clock64() - t0; pc[5]++; pc[8] += (unsigned long long)n_prim; }
        }
    }
    if (PROF && lane == 0) {
        for (int k = 0; k < 9; ++k) atomicAdd(&prof[k], pc[k]);
        atomicAdd(&prof[9], clock64() - t_begin);
        atomicAdd(&prof[10], 1ull);
    }
    if (COUNT) tally_flush(tally, tallies);
}

// Whole paths for trees WITHOUT BVH entries, where every ray does the same work: get_ray, then ray_color as a
// loop (camera.rs:349-408), one path per lane, and a lane takes the next path of the batch the moment its own ends
// (paths are handed out in chunks of RT_CHUNK per wavefront, one atomic per chunk).  Path state never leaves the
// registers: no queues, no hit buffer, no compaction.  The only traffic is one radiance per path into lum[] at the
// path's batch index j = s_local * npix + pixel, which k_film_accumulate then adds up in sample order.
// counters[0] += rays traced, counters[1] = max(deepest bounce).
template <bool COUNT, int MEDIA, bool PLAIN>
__global__ void __launch_bounds__(TRAV_BLOCK, (MEDIA == RT_MEDIA_NONE && PLAIN && !COUNT) ? 3 : 2) k_paths_lockstep(DScene S, rt_camera cam, uint64_t seed, uint32_t npix,
                                                                   uint32_t sample0, uint32_t npaths, uint32_t max_depth,
                                                                   double bgr, double bgg, double bgb, double* lum_r,
                                                                   double* lum_g, double* lum_b, uint32_t* work,
                                                                   unsigned long long* counters,
                                                                   unsigned long long* tallies) {
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false, alive = false;
    V3 o = mk(0, 0, 0), d = mk(1, 1, 1), beta = mk(0, 0, 0);
    uint32_t pix = 0, smp = 0, ctr = 0, depth = 0, slot = 0;
    uint32_t rays = 0, deepest = 0;
    for (;;) {
        const unsigned long long mask = __ballot(!alive);
        if (mask != 0ull && !exhausted) {
            if (pool_next == pool_end) {
                const int leader = __ffsll((long long)mask) - 1;
                uint32_t base = 0;
                if ((int)lane == leader) base = atomicAdd(work, RT_CHUNK);
                base = (uint32_t)__builtin_amdgcn_readlane((int)base, leader);
                pool_next = base < npaths ? base : npaths;
                pool_end = (base + RT_CHUNK) < npaths ? (base + RT_CHUNK) : npaths;
                if (base >= npaths) exhausted = true;
            }
            if (!exhausted) {
                if (!alive) {
                    const uint32_t mine = pool_next + (uint32_t)__popcll(mask & lt_mask);
                    if (mine < pool_end) {
                        slot = mine;
                        pix = mine % npix;
                        smp = sample0 + mine / npix;
                        Rng rng = rng_make(seed, pix, smp, 0u);
                        camera_ray(cam, pix % cam.width, pix / cam.width, rng, o, d);
                        ctr = rng.ctr;
                        beta = mk(1.0, 1.0, 1.0);
                        depth = 0;
                        alive = true;
                    }
                }
                const uint32_t want = pool_next + (uint32_t)__popcll(mask);
                pool_next = want < pool_end ? want : pool_end;
            }
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted) break;
            continue;
        }
        // three stretches entered by whole wavefronts; between them all 64 lanes are back together, which the shared
        // rejection loop in the middle needs
        bool hit = false;
        Rec rec;
        rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
        rt_material m;
        m.kind = RT_MAT_DIFFUSE_LIGHT; m.texture = 0; m.param = 0; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0;
        if (alive) {
            RayVal src; src.o = o; src.d = d;
            Ray w = src.world();
            rays++;
            RngPos rp; rp.k0 = (uint32_t)seed; rp.k1 = (uint32_t)(seed >> 32); rp.pixel = pix; rp.sample = smp; rp.ctr = ctr;
            Best b = closest_hit<COUNT, true, MEDIA>(S, src, 0.001, RT_INF, L, rp, tally);
            ctr = rp.ctr;
            if (b.ref == RT_NO_HIT) {
                const V3 lum = beta * mk(bgr, bgg, bgb);
                lum_r[3u * (size_t)slot] = lum.x; lum_g[3u * (size_t)slot] = lum.y; lum_b[3u * (size_t)slot] = lum.z;
                alive = false;
            } else {
                rec = finalize<false, PLAIN>(S, w, b);
                m = S.mats[rec.mat];
                hit = true;
            }
        }
        const V3 ruv = wave_unit_vector(hit && scatter_wants_unit_vector<PLAIN>(m), (uint32_t)seed, (uint32_t)(seed >> 32),
                                        pix, smp, ctr);
        if (hit) {
            Rng rng = rng_make(seed, pix, smp, ctr);
            V3 nd, att, emit;
            V3 lum = mk(0, 0, 0);
            if (!scatter_given<PLAIN>(S, m, d, rec, rng, ruv, nd, att, emit)) {
                lum = beta * emit;
                alive = false;
            } else if (depth + 1u == max_depth) {
                alive = false;  // the scattered ray would be traced with depth 0: black (camera.rs:350-352)
            } else {
                beta = beta * att;
                o = rec.p;
                d = nd;
                ctr = rng.ctr;
                depth++;
                deepest = deepest > depth ? deepest : depth;
            }
            if (!alive) { lum_r[3u * (size_t)slot] = lum.x; lum_g[3u * (size_t)slot] = lum.y; lum_b[3u * (size_t)slot] = lum.z; }
        }
    }
    unsigned long long r = rays;
    uint32_t dp = deepest;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        uint32_t other = __shfl_down(dp, off);
        dp = dp > other ? dp : other;
    }
    if (lane == 0) { atomicAdd(&counters[0], r); atomicMax(&counters[1], (unsigned long long)dp); }
    if (COUNT) tally_flush(tally, tallies);
}

// Full HitRecord of every ray of a caller-supplied batch from its closest hit (Hittable::hit's out-parameter).
__global__ void k_finalize_batch(DScene S, Queue q, HitBuf hb, uint32_t n, uint8_t* hit, uint32_t* prim, double* t,
                                 double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Best b = hb_get(hb, i);
    bool h = b.ref != RT_NO_HIT;
    Rec rec;
    rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
    if (h) {
        Ray w = ray_world(mk(q.ox[i], q.oy[i], q.oz[i]), mk(q.dx[i], q.dy[i], q.dz[i]));
        rec = finalize<true>(S, w, b);
    }
    if (hit) hit[i] = h ? 1 : 0;
    if (prim) prim[i] = rec.prim_id;
    if (t) t[i] = h ? b.t : 0.0;
    if (p3) { p3[3 * i] = rec.p.x; p3[3 * i + 1] = rec.p.y; p3[3 * i + 2] = rec.p.z; }
    if (n3) { n3[3 * i] = rec.n.x; n3[3 * i + 1] = rec.n.y; n3[3 * i + 2] = rec.n.z; }
    if (uv2) { uv2[2 * i] = rec.u; uv2[2 * i + 1] = rec.v; }
    if (front) front[i] = rec.front ? 1 : 0;
    if (ctr_out) ctr_out[i] = q.ctr[i];
}

// Texture::value (texture.rs) for the hits of one workgroup, called by all its threads together.  A perlin texture costs
// some two thousand f64 instructions (seven octaves of eight gradient lookups, perlin.rs:79-92, then a sine), and a
// wavefront pays them in full whenever a single one of its rays has hit such a texture -- in final_scene nearly every
// wavefront, for a lane or two.  So those hits are collected in LDS, evaluated side by side by as few wavefronts as they
// fill (the same arithmetic on the same point: the same value), and handed back; every other texture is evaluated in place.
// Returns whether this thread's hit is on a perlin texture, and then its value in `tex`.
template <int BLOCK>
__device__ __forceinline__ bool block_perlin_value(const DScene& S, bool wants, uint32_t tex_id, V3 p, V3& tex) {
    __shared__ double pq[BLOCK][4];       // point, scale; the value comes back in [0]
    __shared__ uint32_t pslot[BLOCK];     // perlin table
    __shared__ uint32_t pcount[BLOCK / 64];
    if (wants) tex_id = tex_resolve(S, tex_id, p);  // past any checker: a solid colour, an image or a perlin texture
    const bool perl = wants && S.texs[tex_id].kind == RT_TEX_PERLIN;
    tex = mk(0, 0, 0);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    const unsigned long long mask = __ballot(perl);
    if (lane == 0u) pcount[wave] = (uint32_t)__popcll(mask);
    __syncthreads();
    uint32_t base = 0, total = 0;
#pragma unroll
    for (uint32_t w = 0; w < BLOCK / 64; ++w) { const uint32_t c = pcount[w]; base += w < wave ? c : 0u; total += c; }
    if (total == 0u) return false;  // (the same for every thread of the workgroup)
    const uint32_t pos = base + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
    if (perl) { pq[pos][0] = p.x; pq[pos][1] = p.y; pq[pos][2] = p.z; pq[pos][3] = S.texs[tex_id].scale; pslot[pos] = S.texs[tex_id].a; }
    __syncthreads();
    // One wavefront per collected hit, the hits dealt round the workgroup's wavefronts: the 7 x 8 gradient terms of the
    // turbulence (perlin.rs:46-92) are independent of each other, so lane 8 i + c computes the term of octave i and
    // corner c; the sums are then taken in the reference's order (corner after corner, octave after octave).
    for (uint32_t e = wave; e < total; e += BLOCK / 64) {
        const V3 at = mk(pq[e][0], pq[e][1], pq[e][2]);
        const double scale = pq[e][3];
        const double* vec = S.perlin_vec + 768u * pslot[e];
        const uint32_t* perm = S.perlin_perm + 768u * pslot[e];
        const uint32_t octave = lane >> 3, corner = lane & 7u;
        const double two_i = (double)(1u << octave);               // temp_p doubles from octave to octave: exact
        const V3 tp = mk(at.x * two_i, at.y * two_i, at.z * two_i);
        const double fx = floor(tp.x), fy = floor(tp.y), fz = floor(tp.z);
        const double u = tp.x - fx, v = tp.y - fy, w = tp.z - fz;
        const long long i = (long long)fx, j = (long long)fy, k = (long long)fz;
        const double uu = u * u * (3.0 - 2.0 * u), vv = v * v * (3.0 - 2.0 * v), ww = w * w * (3.0 - 2.0 * w);
        const int di = (int)(corner >> 2), dj = (int)((corner >> 1) & 1u), dk = (int)(corner & 1u);
        const uint32_t fi = perm[(i + di) & 255] ^ perm[256 + ((j + dj) & 255)] ^ perm[512 + ((k + dk) & 255)];
        const V3 c = ld3(vec + 3u * fi);
        const double a = (double)di, b = (double)dj, cc = (double)dk;
        const V3 wv = mk(u - a, v - b, w - cc);
        const double term = (a * uu + (1.0 - a) * (1.0 - uu)) * (b * vv + (1.0 - b) * (1.0 - vv)) *
                            (cc * ww + (1.0 - cc) * (1.0 - ww)) * dot(c, wv);
        double noise = 0.0;  // perlin_noise's accum, its eight terms added in loop order
#pragma unroll
        for (uint32_t q = 0; q < 8u; ++q) noise += lane_pull((lane & ~7u) + q, term);
        double accum = 0.0, weight = 1.0;  // perlin_turbulence's
#pragma unroll
        for (uint32_t o = 0; o < 7u; ++o) { accum += weight * lane_pull(8u * o, noise); weight *= 0.5; }
        if (lane == 0u) pq[e][0] = 1.0 + sin(scale * at.z + 10.0 * fabs(accum));  // texture.rs:152-156
    }
    __syncthreads();
    if (perl) tex = mk(0.5, 0.5, 0.5) * pq[pos][0];
    return perl;
}

// Material::emitted / scatter on the closest hit, throughput update, termination, and compaction of the
// surviving paths into the next queue (camera.rs:349-408 unrolled into a loop).  A path that ends writes its
// radiance once into lum[] at its batch index.
// WALLS (trees whose top level is a run of quads followed by eight-wide trees, see rt_scene::walls): the launches
// before this one have walked the trees only, for the rays whose gate bits say they can reach one, and left the closest
// hit there in `hb`; the quads of entries [walls.first, walls.end) -- which the reference's list scan visits FIRST
// (hittable_list.rs:88-102) -- are tested here, where the ray is in registers anyway, through scalar loads like the
// lockstep walk's.  Why the order does not show in the result: in the list's order the trees come after the quads and
// are handed (tmin, t_quad), t_quad the closest quad's parameter.  The trees hold primitives that test a half-open
// interval only (triangles, spheres: triangle.rs:131, sphere.rs:55; rt_scene::walls is not set otherwise), so under
// that cap a tree reports its closest hit below t_quad if it has one, and nothing else: a leaf's primitives are offered
// (s, min(e, cap, closest so far)) with (s, e) the leaf box's interval (bvh.rs:31-37), a hit at s <= t < min(e, cap)
// passes whether the cap is there or not, and the leaf is entered for it either way (s <= t < cap leaves the capped
// interval non-empty).  Walked without the cap, a tree reports its closest hit t_tree whatever t_quad is; that hit
// replaces the quad's exactly when t_tree < t_quad.  (Trees of quads are another matter: those test a closed interval,
// parallelogram.rs:101-103, and whether a hit AT the cap passes depends on the leaf box's interval being non-empty
// under the cap, which the hit's parameter alone does not tell.)
struct Walls { uint32_t first, end; };
// MODE (scenes whose materials are not all plain, i.e. Lambertian / DiffuseLight / Isotropic over a solid colour): a
// wavefront of the full-material code pays for every material and texture any of its 64 hits has, and keeps too many
// registers to hide the dependent loads of the hit record.  Where most hits ARE on plain materials, the shade phase of a
// bounce is two launches instead: MODE 1, the plain code over every ray, which leaves a hit on any other material
// alone and notes its queue slot in `defer_list` (their number in *defer_count); then MODE 2, the full code over that
// list.  Both append the rays that go on to the same queue through the same counters; which launch a path was shaded by
// shows nowhere in its radiance.  MODE 0: one launch does everything.
#define RT_SHADE_ALL 0
#define RT_SHADE_PLAIN_DEFER 1
#define RT_SHADE_DEFERRED 2
template <bool PLAIN, bool WALLS = false, int MODE = RT_SHADE_ALL>
__global__ void __launch_bounds__(SHADE_BLOCK, PLAIN ? 6 : 4) k_shade(DScene S, const SigInfo* __restrict__ Gp, uint64_t seed,
                                                       Segs segs, Queue q, HitBuf hb,
                                                       Queue out, uint32_t* out_count, double* lum_r, double* lum_g,
                                                       double* lum_b, double bgr, double bgg, double bgb,
                                                       uint32_t iter, uint32_t max_depth, Gates gates, HitBuf hb_out,
                                                       Walls walls, uint32_t* defer_list, uint32_t* defer_count) {
    static_assert(MODE == RT_SHADE_ALL || PLAIN == (MODE == RT_SHADE_PLAIN_DEFER), "plain code defers, full code takes the deferred");
    __shared__ uint32_t wave_total[SHADE_BLOCK / 64][4];
    __shared__ uint32_t block_base[4];
    __shared__ uint32_t defer_wave[SHADE_BLOCK / 64];
    __shared__ uint32_t defer_base;
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t n, idx;
    if constexpr (MODE == RT_SHADE_DEFERRED) {
        // (the grid is sized for every ray of the bounce; the workgroups past the end of the list leave at once)
        const uint32_t listed = *defer_count, base = blockIdx.x * SHADE_BLOCK;
        if (base >= listed) return;
        const bool mine = base + threadIdx.x < listed;
        idx = mine ? defer_list[base + threadIdx.x] : 0u;
        n = mine ? idx + 1u : idx;
    } else {
        idx = seg_slot(segs, n);
    }
    bool alive = false, deferred = false;
    V3 no = mk(0, 0, 0), nd = mk(0, 0, 0), beta = mk(0, 0, 0);
    uint32_t pix = 0, smp = 0, slot = 0, born = 0, ctr = 0;
    bool hit = false;
    Rec rec;
    rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
    rt_material m;
    m.kind = RT_MAT_DIFFUSE_LIGHT; m.texture = 0; m.param = 0; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0;
    V3 wd = mk(0, 0, 0);
    if (idx < n) {
        slot = q.slot[idx];
        born = q.born[idx];
        beta = mk(q.br[idx], q.bg[idx], q.bb[idx]);
        Best b; b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0u;
        Ray w = ray_world(mk(0, 0, 0), mk(1, 1, 1));
        if constexpr (WALLS) {
            const uint32_t reached = born >> RT_BORN_BITS;  // the gate bits of this ray: which trees' launches have seen it
            born &= RT_BORN_MASK;
            w = ray_world(mk(q.ox[idx], q.oy[idx], q.oz[idx]), mk(q.dx[idx], q.dy[idx], q.dz[idx]));
            if (reached != 0u) {
                b = hb_get(hb, idx);
                if (!(b.t < RT_INF)) { b.ref = RT_NO_HIT; b.chain = 0u; }  // (still infinity: no tree was hit, the rest of the record is stale)
            }
            Best bw; bw.t = RT_INF; bw.ref = RT_NO_HIT; bw.chain = 0u;
            Tally unused = tally_zero();
            for (uint32_t i = walls.first; i < walls.end; ++i) {  // (the same entries for every lane: scalar loads)
                const Node nu = load_node(S, i);
                run_prims<false, true>(S, RT_NODE_QUADS, nu.first, nu.count, 0u, w, 0.001, fmin(RT_INF, bw.t), bw, unused);
            }
            const bool tree_wins = b.ref != RT_NO_HIT && b.t < bw.t;
            b.t = tree_wins ? b.t : bw.t; b.ref = tree_wins ? b.ref : bw.ref; b.chain = tree_wins ? b.chain : bw.chain;
        } else {
            b = hb_get(hb, idx);
        }
        if (b.ref == RT_NO_HIT) {
            lum_r[3u * (size_t)slot] = beta.x * bgr; lum_g[3u * (size_t)slot] = beta.y * bgg; lum_b[3u * (size_t)slot] = beta.z * bgb;
        } else {
            pix = q.pix[idx];
            smp = q.smp[idx];
            ctr = q.ctr[idx];
            if constexpr (!WALLS) {
                w = ray_world(mk(q.ox[idx], q.oy[idx], q.oz[idx]), mk(q.dx[idx], q.dy[idx], q.dz[idx]));
            }
            rec = finalize<false, PLAIN>(S, w, b);
            m = S.mats[rec.mat];
            wd = w.d;
            hit = true;
            if constexpr (MODE == RT_SHADE_PLAIN_DEFER) {
                deferred = !material_is_plain(S, m);  // not for this code: the launch after this one takes it
                hit = !deferred;
            }
        }
    }
    // all threads of the workgroup are back together here, which the deferred textures and the shared rejection loop need
    V3 tex_ready = mk(0, 0, 0);
    bool tex_is_ready = false;
    if constexpr (!PLAIN) {
        if (S.n_perlin != 0u)
            tex_is_ready = block_perlin_value<SHADE_BLOCK>(S, hit && material_wants_texture(m), m.texture, rec.p, tex_ready);
    }
    const V3 ruv = wave_unit_vector(hit && scatter_wants_unit_vector<PLAIN>(m), (uint32_t)seed, (uint32_t)(seed >> 32),
                                    pix, smp, ctr);
    if (hit) {
        Rng rng = rng_make(seed, pix, smp, ctr);
        V3 att, emit;
        if (!scatter_given<PLAIN, false>(S, m, wd, rec, rng, ruv, nd, att, emit, tex_is_ready, tex_ready)) {
            lum_r[3u * (size_t)slot] = beta.x * emit.x; lum_g[3u * (size_t)slot] = beta.y * emit.y; lum_b[3u * (size_t)slot] = beta.z * emit.z;
        } else if (iter - born + 1u == max_depth) {
            // the scattered ray would be traced with depth 0: ray_color returns black (camera.rs:350-352)
            lum_r[3u * (size_t)slot] = 0.0; lum_g[3u * (size_t)slot] = 0.0; lum_b[3u * (size_t)slot] = 0.0;
        } else {
            beta = beta * att;
            no = rec.p;
            ctr = rng.ctr;
            alive = true;
        }
    }
    // compaction by signature class (one class unless the queues are sorted): ballot + prefix count per wave and
    // class, one atomic per workgroup and class; the hits left to the next launch likewise, into its list
    uint32_t sig = 0xFFFFFFFFu;
    if (alive) sig = ray_signature(*Gp, no, nd);
    const uint32_t cls = sig & (segs.nclass - 1u);
    uint32_t prefix = 0, defer_prefix = 0;
    for (uint32_t c = 0; c < segs.nclass; ++c) {
        const unsigned long long mask = __ballot(alive && cls == c);
        if (cls == c) prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wave][c] = (uint32_t)__popcll(mask);
    }
    if constexpr (MODE == RT_SHADE_PLAIN_DEFER) {
        const unsigned long long mask = __ballot(deferred);
        defer_prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) defer_wave[wave] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < segs.nclass) {
        uint32_t tot = 0;
        for (int k = 0; k < SHADE_BLOCK / 64; ++k) { uint32_t c = wave_total[k][threadIdx.x]; wave_total[k][threadIdx.x] = tot; tot += c; }
        block_base[threadIdx.x] = tot ? atomicAdd(out_count + RT_COUNT_STRIDE * threadIdx.x, tot) : 0u;
    }
    if constexpr (MODE == RT_SHADE_PLAIN_DEFER) {
        if (threadIdx.x == 64u) {
            uint32_t tot = 0;
            for (int k = 0; k < SHADE_BLOCK / 64; ++k) { uint32_t c = defer_wave[k]; defer_wave[k] = tot; tot += c; }
            defer_base = tot ? atomicAdd(defer_count, tot) : 0u;
        }
    }
    __syncthreads();
    if constexpr (MODE == RT_SHADE_PLAIN_DEFER) {
        if (deferred) defer_list[defer_base + defer_wave[wave] + defer_prefix] = idx;
    }
    uint32_t dst = 0u;
    if (alive) {
        dst = cls * segs.stride + block_base[cls] + wave_total[wave][cls] + prefix;
        out.ox[dst] = no.x; out.oy[dst] = no.y; out.oz[dst] = no.z;
        out.dx[dst] = nd.x; out.dy[dst] = nd.y; out.dz[dst] = nd.z;
        out.br[dst] = beta.x; out.bg[dst] = beta.y; out.bb[dst] = beta.z;
        out.pix[dst] = pix; out.smp[dst] = smp; out.slot[dst] = slot;
        out.ctr[dst] = ctr;
        if (Gp->n != 0u) out.sig[dst] = sig;
        if (gates.n && gates.init_hits) hb_put(hb_out, dst, RT_INF, RT_NO_HIT, 0u);
    }
    // (the gate bits last: by now the new ray is all that is left in registers)
    const uint32_t bits = gates_of(gates, alive, no, nd);
    if (alive) out.born[dst] = born | (gates.tag ? bits << RT_BORN_BITS : 0u);
    if (gates.tag && bits != 0u) hb_set_t(hb_out, dst, RT_INF);
    gates_append(gates, bits, dst, out_count);
}

// One whole bounce for trees WITHOUT BVH entries: the lockstep walk of k_traverse_lockstep followed, in the same
// lane, by what k_shade does with the closest hit.  The hit never goes through memory and the path state is read
// once; the walk's registers are dead by the time the material code starts, so the kernel keeps the walk's occupancy.
#define BOUNCE_BLOCK 256
template <int MEDIA, bool PLAIN>
__global__ void __launch_bounds__(BOUNCE_BLOCK, MEDIA == RT_MEDIA_ANY ? 3 : 4) k_bounce_lockstep(
    DScene S, const SigInfo* __restrict__ Gp, uint64_t seed, Segs segs, Queue q, Queue out, uint32_t* out_count, double* lum_r, double* lum_g,
    double* lum_b, double bgr, double bgg, double bgb, uint32_t iter, uint32_t max_depth) {
    __shared__ uint32_t wave_total[BOUNCE_BLOCK / 64][4];
    __shared__ uint32_t block_base[4];
    uint32_t n;
    const uint32_t idx = seg_slot(segs, n);
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    bool alive = false;
    V3 no = mk(0, 0, 0), nd = mk(0, 0, 0), beta = mk(0, 0, 0);
    uint32_t pix = 0, smp = 0, slot = 0, born = 0, ctr = 0;
    // the three stretches below are entered by whole wavefronts: between them all 64 lanes are back together, which
    // the shared rejection loop in the middle needs
    bool hit = false;
    Rec rec;
    rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
    rt_material m;
    m.kind = RT_MAT_DIFFUSE_LIGHT; m.texture = 0; m.param = 0; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0;
    V3 wd = mk(0, 0, 0);
    // the wavefront walks the copy of the tree without everything its rays' signatures rule out
    const uint32_t first = S.n_variants > 1u ? tree_without(S, sig_skip_mask(idx < n ? q.sig[idx] : 0u)) : 0u;
    if (idx < n) {
        RngPos rp; rp.k0 = (uint32_t)seed; rp.k1 = (uint32_t)(seed >> 32);
        rp.pixel = q.pix[idx]; rp.sample = q.smp[idx]; rp.ctr = q.ctr[idx];
        RaySoA src;
        src.ox = q.ox; src.oy = q.oy; src.oz = q.oz; src.dx = q.dx; src.dy = q.dy; src.dz = q.dz; src.idx = idx;
        const Best b = closest_hit<false, true, MEDIA>(S, src, 0.001, RT_INF, L, rp, tally, first);
        // With full materials the rest of the path state is addressed through a copy of idx the compiler cannot see
        // through: otherwise it forms the addresses (and re-uses the ray loaded above) ahead of the walk and spills
        // them across it.  The plain-material kernel has the registers to spare and gains from the early loads.
        uint32_t at = idx;
        if (!PLAIN) asm volatile("" : "+v"(at));
        slot = q.slot[at];
        born = q.born[at];
        beta = mk(q.br[at], q.bg[at], q.bb[at]);
        pix = rp.pixel; smp = rp.sample; ctr = rp.ctr;
        if (b.ref == RT_NO_HIT) {
            lum_r[3u * (size_t)slot] = beta.x * bgr; lum_g[3u * (size_t)slot] = beta.y * bgg; lum_b[3u * (size_t)slot] = beta.z * bgb;
        } else {
            Ray w = ray_world(mk(q.ox[at], q.oy[at], q.oz[at]), mk(q.dx[at], q.dy[at], q.dz[at]));
            rec = finalize<false, PLAIN>(S, w, b);
            m = S.mats[rec.mat];
            wd = w.d;
            hit = true;
        }
    }
    const V3 ruv = wave_unit_vector(hit && scatter_wants_unit_vector<PLAIN>(m), (uint32_t)seed, (uint32_t)(seed >> 32),
                                    pix, smp, ctr);
    if (hit) {
        Rng rng = rng_make(seed, pix, smp, ctr);
        V3 att, emit;
        if (!scatter_given<PLAIN>(S, m, wd, rec, rng, ruv, nd, att, emit)) {
            lum_r[3u * (size_t)slot] = beta.x * emit.x; lum_g[3u * (size_t)slot] = beta.y * emit.y; lum_b[3u * (size_t)slot] = beta.z * emit.z;
        } else if (iter - born + 1u == max_depth) {
            lum_r[3u * (size_t)slot] = 0.0; lum_g[3u * (size_t)slot] = 0.0; lum_b[3u * (size_t)slot] = 0.0;  // camera.rs:350-352
        } else {
            beta = beta * att;
            no = rec.p;
            ctr = rng.ctr;
            alive = true;
        }
    }
    // compaction by signature class: ballot + prefix count per wave and class, one atomic per workgroup and class
    uint32_t sig = 0xFFFFFFFFu;
    if (alive) sig = ray_signature(*Gp, no, nd);
    const uint32_t cls = sig & (segs.nclass - 1u);
    const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    uint32_t prefix = 0;
    for (uint32_t c = 0; c < segs.nclass; ++c) {
        const unsigned long long mask = __ballot(alive && cls == c);
        if (cls == c) prefix = (uint32_t)__popcll(mask & ((1ull << lane) - 1ull));
        if (lane == 0) wave_total[wave][c] = (uint32_t)__popcll(mask);
    }
    __syncthreads();
    if (threadIdx.x < segs.nclass) {
        uint32_t tot = 0;
        for (int k = 0; k < BOUNCE_BLOCK / 64; ++k) { uint32_t c = wave_total[k][threadIdx.x]; wave_total[k][threadIdx.x] = tot; tot += c; }
        block_base[threadIdx.x] = tot ? atomicAdd(out_count + RT_COUNT_STRIDE * threadIdx.x, tot) : 0u;
    }
    __syncthreads();
    if (alive) {
        uint32_t dst = cls * segs.stride + block_base[cls] + wave_total[wave][cls] + prefix;
        out.ox[dst] = no.x; out.oy[dst] = no.y; out.oz[dst] = no.z;
        out.dx[dst] = nd.x; out.dy[dst] = nd.y; out.dz[dst] = nd.z;
        out.br[dst] = beta.x; out.bg[dst] = beta.y; out.bb[dst] = beta.z;
        out.pix[dst] = pix; out.smp[dst] = smp; out.slot[dst] = slot; out.born[dst] = born;
        out.ctr[dst] = ctr;
        out.sig[dst] = sig;
    }
}

// The tail of a render: once few rays are left and no batch remains to be generated, a launch (or two) per bounce costs
// its latency and little else -- dozens of launches of some tens of microseconds.  This kernel takes the rays that
// are left, one per lane, and follows each path to its end: the same material code and shared rejection loop as the
// bounce and shade kernels, bounce after bounce, no queue written; the walk is the lockstep one for trees without BVH
// entries; for trees with them each lane goes through the stretches of the top level as the traverse launches do, one
// after the other, a wide tree (rt_wide.h) by its own walk and everything else by the plain per-lane walk (with so few
// rays it is the length of the longest path that counts, not how well the lanes are used -- but the wide tree's
// boxes are tight where the reference's reach the origin, and a deep path through a large mesh is many times shorter
// there).  n_stretch = 0: the plain walk over the whole tree.  A lane whose path has ended takes the next ray of the
// queue (a wavefront reserves 64 queue positions per atomic on `work`), so the launch needs no more lanes than the device
// holds at once and its lanes stay in use while rays are left.  counters[0] += rays traced, counters[1] = max(iteration
// index after the last bounce of any path).
#define RT_FINISH_MAX (1u << 17)      // rays in flight (upper bound) from which the tail is taken in one launch
#define RT_FINISH_MAX_BVH (1u << 20)  // the same for trees with BVH entries
// LEAN (trees whose stretches are eight-wide trees of one kind of primitive, LEAN, and runs of quads at the top level, see
// rt_scene::lean): only that kind's test and the quads' are compiled in, and the runs of quads are a loop every lane takes
// together instead of the entry-by-entry walk -- the kernel then needs 150-180 registers instead of 256 and more, and a
// third wavefront fits a SIMD (the kernel spends its time waiting for the records of the tree, one dependent load after
// the other).  many_spheres, all of which is this launch, + 15 %; the tail of cornell_model - 1.4 ms.
// LIGHT (with LEAN; scenes whose textures are solid colours and checkers of them, see rt_scene::light_tex): no perlin
// arithmetic and no (u, v) -- only an image texture reads those, and a sphere's take an acos and an atan2 -- which
// is what keeps spheres with every material at 252 registers (164 without).
// wavefronts per SIMD (168 registers each at three; spheres with every material and texture would spill a hundred of them there)
#define RT_FINISH_WAVES(LEAN, PLAIN, LIGHT) (((LEAN) != 0 && ((PLAIN) || (LIGHT) || (LEAN) == RT_NODE_TRIANGLES)) ? 3 : 2)
template <bool LOCKSTEP, int MEDIA, bool PLAIN, int LEAN = 0, bool LIGHT = false>
__global__ void __launch_bounds__(BOUNCE_BLOCK, RT_FINISH_WAVES(LEAN, PLAIN, LIGHT)) k_finish(
    DScene S, uint64_t seed, Segs segs, Queue q, double* lum_r, double* lum_g, double* lum_b, double bgr, double bgg,
    double bgb, uint32_t iter, uint32_t max_depth, unsigned long long* counters, const Stretch* stretches,
    uint32_t n_stretch, const DWide* wides, uint32_t fast, uint32_t* work, uint32_t born_mask) {
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    const uint32_t lane = threadIdx.x & 63u;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t total = seg_total_all(segs);
    uint32_t pool_next = 0, pool_end = 0;
    bool exhausted = false;
    bool alive = false;
    V3 o = mk(0, 0, 0), d = mk(1, 1, 1), beta = mk(0, 0, 0);
    uint32_t pix = 0, smp = 0, slot = 0, born = 0, ctr = 0;
    uint32_t rays = 0, last = 0;
    uint32_t it = iter;  // the iteration this lane's path is in: every ray of the queue is in iteration `iter`
    for (;;) {
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && !exhausted) {
            if (pool_next == pool_end) {
                uint32_t base = 0;
                if (lane == 0u) base = atomicAdd(work, 64u);
                base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
                pool_next = base < total ? base : total;
                pool_end = base + 64u < total ? base + 64u : total;
                exhausted = base >= total;
            }
            if (!exhausted) {
                const uint32_t mine = pool_next + (uint32_t)__popcll(idle & lt_mask);
                if (!alive && mine < pool_end) {
                    const uint32_t idx = seg_nth_all(segs, mine);
                    o = mk(q.ox[idx], q.oy[idx], q.oz[idx]); d = mk(q.dx[idx], q.dy[idx], q.dz[idx]);
                    beta = mk(q.br[idx], q.bg[idx], q.bb[idx]);
                    pix = q.pix[idx]; smp = q.smp[idx]; slot = q.slot[idx]; born = q.born[idx] & born_mask; ctr = q.ctr[idx];
                    it = iter;
                    alive = true;
                }
                const uint32_t want = pool_next + (uint32_t)__popcll(idle);
                pool_next = want < pool_end ? want : pool_end;
            }
        }
        if (__ballot(alive) == 0ull) {
            if (exhausted) break;
            continue;
        }
        bool hit = false;
        Rec rec;
        rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
        rt_material m;
        m.kind = RT_MAT_DIFFUSE_LIGHT; m.texture = 0; m.param = 0; m.albedo[0] = m.albedo[1] = m.albedo[2] = 0;
        V3 lum = mk(0, 0, 0);
        bool ended = false;
        if (alive) {
            RayVal src; src.o = o; src.d = d;
            const Ray w = src.world();
            rays++;
            last = it + 1u;
            RngPos rp; rp.k0 = (uint32_t)seed; rp.k1 = (uint32_t)(seed >> 32); rp.pixel = pix; rp.sample = smp; rp.ctr = ctr;
            Best b;
            bool walked = false;
            if constexpr (LEAN != 0) {
                walked = true;
                b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0;
                for (uint32_t si = 0; si < n_stretch; ++si) {
                    const Stretch st = stretches[si];
                    if (st.kind == RT_STRETCH_WIDE) {
                        DWide W = wides[st.wide];
                        W.fast = fast;
                        WLane wl;
                        wide_begin(wl, S, W, w, 0.001, RT_INF, b);
                        // Spheres: the kind of step more lanes wait for, both kinds when both are many (as in
                        // k_traverse_wide; many_spheres - 5 % of the launch).  Triangles: both kinds every turn (an item is
                        // one triangle, so a lane is back on a node after every primitive step, and a turn it sits out
                        // is a turn lost: a large mesh + 15 % with the rule above).
                        for (;;) {
                            const int n_node = __popcll(__ballot(wl.state == RT_W_NODE));
                            const int n_prim = __popcll(__ballot(wl.state == RT_W_PRIM));
                            if (n_node + n_prim == 0) break;
                            const bool each = LEAN != RT_NODE_SPHERES;
                            if ((each || n_node >= RT_WIDE_STEP_THRESHOLD || n_node >= n_prim) && wl.state == RT_W_NODE) wide_node_step(W, wl);
                            else if ((each || n_prim >= RT_WIDE_STEP_THRESHOLD || n_prim > n_node) && wl.state == RT_W_PRIM) wide_prim_step<false, WideKeepsAll, LEAN>(S, W, wl, tally);
                        }
                        b = wl.best;
                    } else {
                        for (uint32_t i = st.first; i < st.end; ++i) {  // (the same entries for every lane: scalar loads)
                            const Node nu = load_node(S, i);
                            run_prims<false, true>(S, RT_NODE_QUADS, nu.first, nu.count, 0u, w, 0.001, fmin(RT_INF, b.t), b, tally);
                        }
                    }
                }
            } else if constexpr (!LOCKSTEP) {
                if (n_stretch != 0u) {
                    walked = true;
                    b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0;
                    for (uint32_t si = 0; si < n_stretch; ++si) {
                        const Stretch st = stretches[si];
                        if (st.kind == RT_STRETCH_WIDE) {
                            DWide W = wides[st.wide];
                            W.fast = fast;
                            WLane wl;
                            wide_begin(wl, S, W, w, 0.001, RT_INF, b);
                            while (wl.state != RT_W_DONE) {
                                if (wl.state == RT_W_NODE) wide_node_step(W, wl);
                                else wide_prim_step<false>(S, W, wl, tally);
                            }
                            b = wl.best;
                        } else {
                            b = closest_hit_range<false, false, MEDIA>(S, src, 0.001, RT_INF, L, rp, tally, st.first, st.end, b);
                        }
                    }
                }
            }
            if constexpr (LEAN == 0) { if (!walked) b = closest_hit<false, LOCKSTEP, MEDIA>(S, src, 0.001, RT_INF, L, rp, tally); }
            ctr = rp.ctr;
            if (b.ref == RT_NO_HIT) {
                lum = beta * mk(bgr, bgg, bgb);
                ended = true;
            } else {
                rec = finalize<false, PLAIN || LIGHT>(S, w, b);
                m = S.mats[rec.mat];
                hit = true;
            }
        }
        const V3 ruv = wave_unit_vector(hit && scatter_wants_unit_vector<PLAIN>(m), (uint32_t)seed, (uint32_t)(seed >> 32),
                                        pix, smp, ctr);
        if (hit) {
            Rng rng = rng_make(seed, pix, smp, ctr);
            V3 nd, att, emit;
            if (!scatter_given<PLAIN, !LIGHT>(S, m, d, rec, rng, ruv, nd, att, emit)) {
                lum = beta * emit;
                ended = true;
            } else if (it - born + 1u == max_depth) {
                ended = true;  // the scattered ray would be traced with depth 0: black (camera.rs:350-352)
            } else {
                beta = beta * att;
                o = rec.p;
                d = nd;
                ctr = rng.ctr;
            }
        }
        if (ended) { lum_r[3u * (size_t)slot] = lum.x; lum_g[3u * (size_t)slot] = lum.y; lum_b[3u * (size_t)slot] = lum.z; alive = false; }
        it++;
    }
    unsigned long long r = rays;
    uint32_t lt = last;
    for (int off = 32; off > 0; off >>= 1) {
        r += __shfl_down(r, off);
        const uint32_t other = __shfl_down(lt, off);
        lt = lt > other ? lt : other;
    }
    if ((threadIdx.x & 63u) == 0u && r != 0ull) { atomicAdd(&counters[0], r); atomicMax(&counters[1], (unsigned long long)lt); }
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
        r = r + lum_r[3u * k]; g = g + lum_g[3u * k]; b = b + lum_b[3u * k];
    }
    film[3u * pix] = r; film[3u * pix + 1u] = g; film[3u * pix + 2u] = b;
}
__global__ void k_film_add(double* dst, const double* src, size_t n) {  // one shard's film onto another's
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) dst[i] = dst[i] + src[i];
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

// get_ray + ray_color for chosen (pixel, sample) pairs, one whole path per lane (camera.rs:349-408 as a loop).
__global__ void __launch_bounds__(TRAV_BLOCK) k_trace_samples(DScene S, rt_camera cam, uint64_t seed,
                                                              uint32_t max_depth, double bgr, double bgg, double bgb,
                                                              uint64_t n, const uint32_t* pixel, const uint32_t* sample,
                                                              double* radiance3, uint32_t* nrays) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Levels L; L.base = g_levels + threadIdx.x; L.stride = blockDim.x;
    Tally tally = tally_zero();
    Rng rng = rng_make(seed, pixel[i], sample[i], 0u);
    V3 o, d;
    camera_ray(cam, pixel[i] % cam.width, pixel[i] / cam.width, rng, o, d);
    V3 beta = mk(1.0, 1.0, 1.0), lum = mk(0, 0, 0);
    uint32_t rays = 0;
    for (uint32_t depth = 0; depth < max_depth; ++depth) {
        RayVal src; src.o = o; src.d = d;
        Ray w = src.world();
        rays++;
        RngPos rp; rp.k0 = rng.k0; rp.k1 = rng.k1; rp.pixel = rng.pixel; rp.sample = rng.sample; rp.ctr = rng.ctr;
        Best b = closest_hit<false, false>(S, src, 0.001, RT_INF, L, rp, tally);
        rng.ctr = rp.ctr;
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

// ------------------------------------------------------------------------------------------------ static scratch
// Device memory of rt_intersect_batch_noalloc, the entry point for callers that cannot allocate: the closest hits
// that travel from one traverse launch to the next and on to the kernel that writes the records (8 + 4 + 4 bytes per
// ray), then the counter the per-lane kernels hand rays out with.  One copy per GPU, held until the process ends.
#define RT_SCRATCH_BYTES (RT_NOALLOC_MAX_RAYS * 16u + 256u)
static __device__ __align__(256) unsigned char scratch[RT_SCRATCH_BYTES];
// The array's address on each device (the name above is only a host-side stand-in for it), looked up once per device
// when a scene is first committed there.
#define RT_MAX_DEVICES 64
static unsigned char* g_scratch_ptr[RT_MAX_DEVICES] = {};
static std::mutex g_scratch_mutex;
static int scratch_lookup(int device) {  // with `device` current
    if (device < 0 || device >= RT_MAX_DEVICES) return RT_OK;  // (such a device simply has no allocation-free entry point)
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (g_scratch_ptr[device]) return RT_OK;
    void* ptr = nullptr;
    HIP_TRY(hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)));
    g_scratch_ptr[device] = static_cast<unsigned char*>(ptr);
    return RT_OK;
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
    std::vector<double> spheres, quads, tris, quadrics, boxes, cull_boxes;
    std::vector<uint32_t> sphere_mat, sphere_id, quad_mat, quad_id, tri_mat, tri_id, quadric_mat, quadric_id;
    std::vector<rt_medium> media;
    std::vector<rt_xform_op> ops;
    std::vector<rt_chain> chains;
    std::vector<rt_node> nodes;

    bool committed = false;
    int device = -1;
    uint32_t levels = 0;  // interval slots per lane the walk needs
    bool fast_bvh = false; // rt_scene_set_fast_bvh
    bool has_bvh = false; // selects the traverse kernel: lockstep (no BVH entries) or persistent
    // A tree with BVH entries is walked in stretches of whole top-level entries: those without a BVH entry below them
    // (every ray does the same work there) by the lockstep kernel, the others by the persistent one; the closest hit
    // so far travels from one launch to the next in the hit buffer.
    std::vector<Stretch> stretches;
    std::vector<WideBvh> wide;     // the eight-wide trees of the BVH entries that take one (rt_wide.h)
    std::vector<DWide> d_wide;     // on the device
    const Stretch* dev_stretches = nullptr;  // the two vectors above in device memory, for the tail kernel
    const DWide* dev_wides = nullptr;
    // Top level = one run (or several) of quads in the world's own frame, then nothing but eight-wide trees of triangles
    // or spheres: a render then walks the trees first and tests the quads inside the shade kernel (k_shade's WALLS
    // form), which saves the launch that would read every ray for those few quads.
    bool walls = false;
    // Every stretch of the top level is an eight-wide tree of this one kind of primitive (triangles or spheres) or a run
    // of quads in the world's own frame, and there is no medium: the tail kernel has a form for that (k_finish's LEAN).
    int lean = 0;
    bool light_tex = false;  // every texture is a solid colour or a checker (of such): nothing reads (u, v), no perlin noise
    GateInfo gate;                 // one gate per wide tree (the first RT_MAX_GATES of them): wide tree w has gate w
    const GateInfo* d_gate = nullptr;
    int media_mode = RT_MEDIA_NONE;  // RT_MEDIA_FLAT when every medium's boundary is plain primitives, else RT_MEDIA_ANY
    bool plain = false;   // only Lambertian / DiffuseLight / Isotropic over SolidColor: selects the lean shade kernel
    uint32_t persist_waves = 0;  // resident wavefronts of the persistent kernel on the whole device
    uint32_t path_waves = 0;     // the same for the whole-path kernel
    uint32_t wide_waves = 0;     // the same for the wide-tree kernel
    uint32_t n_cus = 1;          // compute units of the device
    std::vector<void*> allocs;
    DScene d;
    const uint4* nodes_nocull = nullptr;  // the tree with every cull box infinite (RT_FLAG_NO_CULL)
    SigInfo sig;                          // what ray signatures are made of (trees without BVH entries)
    const SigInfo* d_sig = nullptr;       // on the device
    hipStream_t stream = nullptr;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    hipEvent_t ring_ev[RT_RING][4] = {};  // per queued iteration: start, after traverse, end, counts copied
    uint32_t* d_ring = nullptr;           // RT_RING sets of class counters (RT_RING_WORDS words each)
    uint32_t* h_ring = nullptr;           // pinned copy
    // render workspace, grown on demand
    size_t ws_paths = 0, ws_lum = 0, ws_hits = 0;
    uint64_t ws_bytes = 0;  // device memory the workspace holds
    std::vector<void*> ws_allocs;
    Queue qa, qb;
    HitBuf hb, hb2;   // closest hits of the rays in qa / qb (hb2 only where work lists are in use)
    uint32_t* wl = nullptr;  // work lists, ws_paths slots each
    size_t ws_lists = 0;
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
        int prev = -1;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        (void)hipSetDevice(s->device);
        free_list(s->allocs);
        free_list(s->ws_allocs);
        if (s->h_count) (void)hipHostFree(s->h_count);
        for (auto& e : s->ev) if (e) (void)hipEventDestroy(e);
        for (auto& r : s->ring_ev) for (auto& e : r) if (e) (void)hipEventDestroy(e);
        if (s->h_ring) (void)hipHostFree(s->h_ring);
        if (s->stream) (void)hipStreamDestroy(s->stream);
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    delete s;
}

int rt_scene_set_fast_bvh(rt_scene* s, int on) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    s->fast_bvh = on != 0;
    return RT_OK;
}

#define NEED_UNCOMMITTED(s)                                                        \
    if (!(s)) return fail(RT_ERR_INVALID, "null scene");                           \
    if ((s)->committed) return fail(RT_ERR_STATE, "scene already committed")

static int rt_scene_set_materials_impl(rt_scene* s, const rt_material* m, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !m) return fail(RT_ERR_INVALID, "null materials");
    s->mats.assign(m, m + n);
    return RT_OK;
}
static int rt_scene_set_textures_impl(rt_scene* s, const rt_texture* t, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !t) return fail(RT_ERR_INVALID, "null textures");
    s->texs.assign(t, t + n);
    return RT_OK;
}
static int rt_scene_set_image_impl(rt_scene* s, uint32_t slot, const uint8_t* rgb8, uint32_t w, uint32_t h) {
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
static int rt_scene_set_perlin_impl(rt_scene* s, uint32_t slot, const double* vec256x3, const uint32_t* perm3x256) {
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
static int rt_scene_set_spheres_impl(rt_scene* s, const double* cr, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->spheres, s->sphere_mat, s->sphere_id, cr, 4, mat, id, n);
}
static int rt_scene_set_quads_impl(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->quads, s->quad_mat, s->quad_id, rec16, 16, mat, id, n);
}
static int rt_scene_set_triangles_impl(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->tris, s->tri_mat, s->tri_id, rec16, 16, mat, id, n);
}
static int rt_scene_set_quadrics_impl(rt_scene* s, const double* rec10, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    if (!s) return fail(RT_ERR_INVALID, "null scene");
    return set_prims(s, s->quadrics, s->quadric_mat, s->quadric_id, rec10, 10, mat, id, n);
}
static int rt_scene_set_media_impl(rt_scene* s, const rt_medium* m, uint32_t n) {
    NEED_UNCOMMITTED(s);
    if (n && !m) return fail(RT_ERR_INVALID, "null media");
    s->media.assign(m, m + n);
    return RT_OK;
}
static int rt_scene_set_xforms_impl(rt_scene* s, const rt_xform_op* ops, uint32_t n_ops, const rt_chain* chains,
                        uint32_t n_chains) {
    NEED_UNCOMMITTED(s);
    if ((n_ops && !ops) || (n_chains && !chains)) return fail(RT_ERR_INVALID, "null transform arrays");
    if (n_chains > 65535u) return fail(RT_ERR_LIMIT, "too many transform chains");
    s->ops.assign(ops, ops + n_ops);
    s->chains.assign(chains, chains + n_chains);
    return RT_OK;
}
static int rt_scene_set_nodes_impl(rt_scene* s, const rt_node* nodes, uint32_t n_nodes, const double* boxes6, uint32_t n_boxes) {
    NEED_UNCOMMITTED(s);
    if ((n_nodes && !nodes) || (n_boxes && !boxes6)) return fail(RT_ERR_INVALID, "null node arrays");
    s->nodes.assign(nodes, nodes + n_nodes);
    s->boxes.assign(boxes6, boxes6 + (size_t)6 * n_boxes);
    return RT_OK;
}
static int rt_scene_set_cull_boxes_impl(rt_scene* s, const double* tight6, uint32_t n_boxes) {
    NEED_UNCOMMITTED(s);
    if (n_boxes && !tight6) return fail(RT_ERR_INVALID, "null cull boxes");
    s->cull_boxes.assign(tight6, tight6 + (size_t)6 * n_boxes);
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
    if (!s->cull_boxes.empty() && s->cull_boxes.size() != s->boxes.size())
        return fail(RT_ERR_INVALID, "%zu cull boxes for %u boxes", s->cull_boxes.size() / 6, nbox);
    for (size_t i = 0; i < s->cull_boxes.size(); ++i)
        if (s->cull_boxes[i] != s->cull_boxes[i]) return fail(RT_ERR_INVALID, "cull box %zu: NaN bound", i / 6);
    struct Open { uint32_t skip; bool medium; bool levelled; };
    std::vector<Open> open;
    uint32_t max_level = 0;
    for (uint32_t i = 0; i < nn; ++i) {
        while (!open.empty() && open.back().skip == i) open.pop_back();
        const rt_node& n = s->nodes[i];
        uint32_t depth = 0;
        for (auto& o : open) depth += o.levelled;
        if (n.level != depth) return fail(RT_ERR_INVALID, "node %u: level %u but %u enclosing BVH/medium entries", i, n.level, depth);
        if (n.chain >= s->chains.size()) return fail(RT_ERR_INVALID, "node %u: chain out of range", i);
        max_level = std::max<uint32_t>(max_level, n.level);
        uint32_t limit = open.empty() ? nn : open.back().skip;
        bool in_medium = false;
        for (auto& o : open) in_medium |= o.medium;
        if (n.kind == RT_NODE_CULL) {
            if (n.skip <= i || n.skip > limit) return fail(RT_ERR_INVALID, "node %u: skip %u outside (%u, %u]", i, n.skip, i, limit);
            if (n.first >= nbox) return fail(RT_ERR_INVALID, "node %u: box out of range", i);
            Open o = {n.skip, false, false};
            open.push_back(o);
        } else if (n.kind == RT_NODE_BVH || n.kind == RT_NODE_MEDIUM) {
            if (n.skip <= i || n.skip > limit) return fail(RT_ERR_INVALID, "node %u: skip %u outside (%u, %u]", i, n.skip, i, limit);
            if (n.kind == RT_NODE_BVH) {
                if (n.first >= nbox) return fail(RT_ERR_INVALID, "node %u: box out of range", i);
            } else {
                if (n.first >= s->media.size()) return fail(RT_ERR_INVALID, "node %u: medium out of range", i);
                if (in_medium) return fail(RT_ERR_LIMIT, "node %u: a medium inside a medium's boundary is not supported", i);
            }
            if (n.level + 2u > RT_MAX_LEVELS) return fail(RT_ERR_LIMIT, "node %u: nesting deeper than %d levels", i, RT_MAX_LEVELS);
            Open o = {n.skip, n.kind == RT_NODE_MEDIUM, true};
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
    if (!open.empty()) return fail(RT_ERR_INVALID, "unterminated BVH/medium/cull entry");
    s->levels = max_level + 1u;
    s->has_bvh = false;
    for (uint32_t i = 0; i < nn; ++i) s->has_bvh |= s->nodes[i].kind == RT_NODE_BVH;
    s->plain = true;
    for (const rt_material& m : s->mats) {
        const bool kind_ok = m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC;
        s->plain = s->plain && kind_ok && s->texs[m.texture].kind == RT_TEX_SOLID;
    }
    if (s->levels > RT_MAX_LEVELS) return fail(RT_ERR_LIMIT, "scene needs %u levels, limit %d", s->levels, RT_MAX_LEVELS);
    s->wide.clear();
    s->stretches = build_stretches(s->nodes.data(), nn, s->has_bvh, [&](uint32_t e) -> int {
        WideBvh w;
        if (!build_wide(s->nodes.data(), nn, e, s->boxes.data(), s->spheres.data(), s->quads.data(), s->tris.data(), w)) return -1;
        s->wide.push_back(std::move(w));
        return (int)s->wide.size() - 1;
    });
    s->walls = false;
    if (s->has_bvh && s->stretches.size() >= 2 && s->stretches[0].kind == RT_STRETCH_LOCKSTEP && s->wide.size() <= RT_MAX_GATES) {
        bool ok = true;
        uint32_t quads = 0;
        for (uint32_t i = s->stretches[0].first; i < s->stretches[0].end; ++i) {
            const rt_node& n = s->nodes[i];
            ok = ok && n.kind == RT_NODE_QUADS && n.level == 0 && n.chain == 0;
            quads += n.count;
        }
        ok = ok && quads <= 64u;  // (a long list of quads is work worth a launch of its own)
        for (size_t k = 1; k < s->stretches.size() && ok; ++k) {
            const Stretch& st = s->stretches[k];
            ok = st.kind == RT_STRETCH_WIDE && st.wide >= 0;
            // primitives that test a half-open interval only (see k_shade)
            if (ok) ok = s->wide[(size_t)st.wide].only == RT_NODE_TRIANGLES || s->wide[(size_t)st.wide].only == RT_NODE_SPHERES;
        }
        s->walls = ok;
    }
    s->lean = 0;
    if (s->has_bvh && !s->stretches.empty()) {  // (a medium is neither kind of stretch)
        int only = 0;
        bool ok = true;
        for (const Stretch& st : s->stretches) {
            if (st.kind == RT_STRETCH_WIDE && st.wide >= 0) {
                const int o = (int)s->wide[(size_t)st.wide].only;
                ok = ok && (o == RT_NODE_TRIANGLES || o == RT_NODE_SPHERES) && (only == 0 || only == o);
                only = o;
            } else {
                ok = ok && st.kind == RT_STRETCH_LOCKSTEP;
                for (uint32_t i = st.first; i < st.end && ok; ++i) {
                    const rt_node& n = s->nodes[i];
                    ok = n.kind == RT_NODE_QUADS && n.level == 0 && n.chain == 0;
                }
            }
        }
        if (ok) s->lean = only;
    }
    s->light_tex = true;
    for (const rt_texture& t : s->texs) s->light_tex = s->light_tex && (t.kind == RT_TEX_SOLID || t.kind == RT_TEX_CHECKER);
    return RT_OK;
}

extern "C++" {
template <class T>
static int upload(rt_scene* s, const std::vector<T>& v, const T** out) {
    *out = nullptr;
    size_t bytes = std::max<size_t>(v.size() * sizeof(T), 16);  // never hand a kernel a null table
    void* p = nullptr;
    HIP_TRY(rt_hip_malloc(&p, bytes));
    s->allocs.push_back(p);
    HIP_TRY(hipMemset(p, 0, bytes));
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = (const T*)p;
    return RT_OK;
}
}  // extern "C++"

static int rt_scene_commit_impl(rt_scene* s, int device) {
    NEED_UNCOMMITTED(s);
    int rc = validate(s);
    if (rc) return rc;
    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) return fail(RT_ERR_DEVICE, "device %d not present (%d visible)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    if ((rc = scratch_lookup(device))) return rc;
    s->device = device;
    DScene& d = s->d;
    d.n_nodes = (uint32_t)s->nodes.size();
#define UP(vec, field) if ((rc = upload(s, vec, &d.field))) return rc
    // tree entries as the kernels read them (rt_records.h), and a second copy with every cull box switched off
    {
        const double* cb = s->cull_boxes.empty() ? nullptr : s->cull_boxes.data();
        // (signatures sort the queues of trees without BVH entries; with them the queue has one class, wavefronts
        // practically never agree on a signature, and the kernels neither compute nor store one)
        const std::vector<uint32_t> sig = s->has_bvh ? std::vector<uint32_t>() : signature_nodes(s->nodes.data(), s->nodes.size(), RT_MAX_SIG);
        {
            DScene h;  // host pointers, for ray_in_chain
            std::memset(&h, 0, sizeof h);
            h.chains = s->chains.data(); h.ops = s->ops.data();
            std::memset(&s->sig, 0, sizeof s->sig);
            s->sig.n = (uint32_t)sig.size();
            for (size_t b = 0; b < sig.size(); ++b)
                signature_entry(s->sig, h, (uint32_t)b, s->nodes[sig[b]], s->boxes.data() + 6 * (size_t)s->nodes[sig[b]].first);
            const std::vector<SigInfo> one(1, s->sig);
            if ((rc = upload(s, one, &s->d_sig))) return rc;
        }
        std::vector<uint32_t> rec = build_record_variants(s->nodes.data(), s->nodes.size(), s->boxes.data(), cb, true, sig);
        d.n_variants = 1u << sig.size();
        s->media_mode = RT_MEDIA_NONE;
        for (size_t i = 0; i < s->nodes.size(); ++i) {
            if (s->nodes[i].kind != RT_NODE_MEDIUM) continue;
            const bool flat = rec[32 * i + 3] == 1u;  // the record's count field
            if (!flat) s->media_mode = RT_MEDIA_ANY;
            else if (s->media_mode == RT_MEDIA_NONE) s->media_mode = RT_MEDIA_FLAT;
        }
        std::vector<float> leaf_boxes;
        if (s->has_bvh) build_leaf_boxes(s->nodes.data(), s->nodes.size(), s->tris.data(), s->quads.data(), rec, leaf_boxes);
        leaf_boxes.resize(leaf_boxes.size() + 48, 0.0f);
        if ((rc = upload(s, leaf_boxes, &d.leaf_sub))) return rc;
        std::vector<uint32_t> rec_nc = build_records(s->nodes.data(), s->nodes.size(), s->boxes.data(), cb, false, true);
        const uint32_t *d_rec = nullptr, *d_rec_nc = nullptr;
        if ((rc = upload(s, rec, &d_rec))) return rc;
        if ((rc = upload(s, rec_nc, &d_rec_nc))) return rc;
        d.nodes = reinterpret_cast<const uint4*>(d_rec);
        s->nodes_nocull = reinterpret_cast<const uint4*>(d_rec_nc);
    }
    s->d_wide.clear();
    for (const WideBvh& w : s->wide) {
        DWide dw;
        const uint32_t* dn = nullptr;
        const double* db = nullptr;
        if ((rc = upload(s, w.nodes, &dn))) return rc;
        if ((rc = upload(s, w.refbox, &db))) return rc;
        dw.nodes = reinterpret_cast<const uint4*>(dn);
        dw.refbox = reinterpret_cast<const double2*>(db);
        for (int ax = 0; ax < 3; ++ax) { dw.base[ax] = w.base[ax]; dw.step[ax] = w.step[ax]; }
        dw.K = w.K; dw.lambda = w.lambda; dw.chain = w.chain; dw.fast = 0u;
        for (int ax = 0; ax < 3; ++ax) dw.slack[ax] = w.slack[ax];
        s->d_wide.push_back(dw);
    }
    if ((rc = upload(s, s->stretches, &s->dev_stretches))) return rc;
    if ((rc = upload(s, s->d_wide, &s->dev_wides))) return rc;
    {
        DScene h;  // host pointers, for chain_map
        std::memset(&h, 0, sizeof h);
        h.chains = s->chains.data(); h.ops = s->ops.data();
        std::memset(&s->gate, 0, sizeof s->gate);
        s->gate.n = (uint32_t)std::min<size_t>(s->wide.size(), RT_MAX_GATES);
        for (uint32_t b = 0; b < s->gate.n; ++b) {
            std::memcpy(s->gate.box[b], s->wide[b].bounds, 48);
            chain_map(h, s->wide[b].chain, s->gate.xf[b]);
        }
        const std::vector<GateInfo> one(1, s->gate);
        if ((rc = upload(s, one, &s->d_gate))) return rc;
    }
    {
        std::vector<uint32_t> axis = quad_axis_flags(s->quads.data(), s->quad_mat.size());
        UP(axis, quad_axis);
    }
    UP(s->spheres, spheres); UP(s->quads, quads); UP(s->tris, tris);
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
    d.n_perlin = 0u;
    for (const rt_texture& t : s->texs) d.n_perlin += t.kind == RT_TEX_PERLIN ? 1u : 0u;
#undef UP
    HIP_TRY(hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking));
    for (auto& e : s->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& r : s->ring_ev) for (auto& e : r) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc((void**)&s->h_ring, RT_RING * RT_RING_WORDS * 4, hipHostMallocDefault));
    HIP_TRY(rt_hip_malloc((void**)&s->d_ring, RT_RING * RT_RING_WORDS * 4));
    s->allocs.push_back(s->d_ring);
    HIP_TRY(hipHostMalloc((void**)&s->h_count, 1024, hipHostMallocDefault));
    HIP_TRY(rt_hip_malloc((void**)&s->d_count, 1024));
    HIP_TRY(hipMemset(s->d_count, 0, 1024));
    s->allocs.push_back(s->d_count);
    HIP_TRY(rt_hip_malloc((void**)&s->d_tallies, 64));
    s->allocs.push_back(s->d_tallies);
    {
        hipDeviceProp_t prop;
        HIP_TRY(hipGetDeviceProperties(&prop, device));
        int per_cu = 0;
        const size_t lds_bytes = (size_t)s->levels * TRAV_BLOCK * sizeof(double2);
        if (s->media_mode == RT_MEDIA_NONE)
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_traverse_persistent<false, RT_MEDIA_NONE>, TRAV_BLOCK, lds_bytes));
        else if (s->media_mode == RT_MEDIA_FLAT)
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_traverse_persistent<false, RT_MEDIA_FLAT>, TRAV_BLOCK, lds_bytes));
        else
            HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_traverse_persistent<false, RT_MEDIA_ANY>, TRAV_BLOCK, lds_bytes));
        if (per_cu < 1) per_cu = 1;
        s->persist_waves = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        {
            const size_t lds = (size_t)s->levels * TRAV_BLOCK * sizeof(double2);
#define OCC(M, P) HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_paths_lockstep<false, M, P>, TRAV_BLOCK, lds))
            if (s->media_mode == RT_MEDIA_NONE) { if (s->plain) OCC(RT_MEDIA_NONE, true); else OCC(RT_MEDIA_NONE, false); }
            else if (s->media_mode == RT_MEDIA_FLAT) { if (s->plain) OCC(RT_MEDIA_FLAT, true); else OCC(RT_MEDIA_FLAT, false); }
            else { if (s->plain) OCC(RT_MEDIA_ANY, true); else OCC(RT_MEDIA_ANY, false); }
#undef OCC
        }
        if (per_cu < 1) per_cu = 1;
        s->path_waves = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        // (every instantiation of the wide walk holds four wavefronts per SIMD: the launch bound its registers are kept to)
        HIP_TRY(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k_traverse_wide<false, false, 0>, TRAV_BLOCK, 0));
        if (per_cu < 1) per_cu = 1;
        s->wide_waves = (uint32_t)prop.multiProcessorCount * (uint32_t)per_cu;
        s->n_cus = (uint32_t)prop.multiProcessorCount;
    }
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
    HIP_TRY(rt_hip_malloc(&p, std::max<size_t>(n * sizeof(T), 16)));
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
    HIP_TRY(rt_hip_malloc(&p, std::max<size_t>(n * sizeof(T), 16)));
    keep.push_back(p);
    *out = (T*)p;
    return RT_OK;
}
}  // extern "C++"

// One traverse+intersect launch over the rays of `a` on the scene's stream.
// `n_upper`: no fewer rays than the launch will find (it sizes the grid; the kernels read the true numbers themselves).
// k0: the first stretch to walk (1 when the quads of stretch 0 are left to the shade kernel, see rt_scene::walls)
static int launch_traverse(rt_scene* s, const TravArgs& a_, bool count, bool nocull, uint32_t n_upper, uint32_t* work = nullptr,
                           size_t k0 = 0, const double* list_share = nullptr) {
    if (!work) work = s->d_count + 4;  // the counter the per-lane kernels hand rays out with
    DScene d = s->d;
    if (nocull) { d.nodes = s->nodes_nocull; d.n_variants = 1u; }
    // every stretch of the queue starts a workgroup of its own
    const uint32_t waves = (n_upper + TRAV_BLOCK - 1) / TRAV_BLOCK + (a_.segs.in_count ? a_.segs.nclass + 1u : 0u);
    for (size_t k = k0; k < s->stretches.size(); ++k) {
        const Stretch& st = s->stretches[k];
        TravArgs a = a_;
        a.first = st.first; a.end = st.end; a.resume = (k > 0 || a_.resume) ? 1u : 0u;
        // interval slots for this stretch alone (a deep BVH elsewhere in the tree does not cost this launch its occupancy)
        const size_t lds = (size_t)std::min(st.levels, s->levels) * TRAV_BLOCK * sizeof(double2);
        a.wl = nullptr; a.wl_count = nullptr;
        if (st.kind == RT_STRETCH_WIDE && !nocull) {  // (a counting pass with the cull boxes off does the reference's tests)
            if (a_.wl && (uint32_t)st.wide < s->gate.n) {  // a_.wl / a_.wl_count: the lists' base and the counter set
                a.wl = a_.wl + (size_t)st.wide * s->ws_paths;
                a.wl_count = a_.wl_count + RT_COUNT_STRIDE * (4u + (uint32_t)st.wide);
            }
            HIP_TRY(hipMemsetAsync(work, 0, 4, s->stream));  // the work counter of this launch
#if defined(RT_DEV_KNOBS)  // development build (make DEV=1): grid divisor, thresholds and the cycle profile from the environment
            static const uint32_t wdiv = getenv("RT_WIDE_DIV") ? (uint32_t)std::max(1, atoi(getenv("RT_WIDE_DIV"))) : 1u;
            static const int tf = getenv("RT_WIDE_TF") ? atoi(getenv("RT_WIDE_TF")) : RT_WIDE_FETCH_THRESHOLD;
            static const int ts = getenv("RT_WIDE_TS") ? atoi(getenv("RT_WIDE_TS")) : RT_WIDE_STEP_THRESHOLD;
            static const bool prof = getenv("RT_WIDE_PROF") != nullptr;
#else
            const uint32_t wdiv = 1u;
            const int tf = RT_WIDE_FETCH_THRESHOLD, ts = RT_WIDE_STEP_THRESHOLD;
            const bool prof = false;
#endif
#if defined(RT_DEV_KNOBS)
            static const uint32_t per_lane = getenv("RT_WIDE_PER_LANE") ? (uint32_t)std::max(1, atoi(getenv("RT_WIDE_PER_LANE"))) : RT_WIDE_RAYS_PER_LANE;
#else
            const uint32_t per_lane = RT_WIDE_RAYS_PER_LANE;
#endif
            // A launch over few rays is as long as its longest walk, step after step, and a step takes the longer the more
            // wavefronts share the SIMD: below RT_WIDE_RAYS_PER_LANE rays of the tree's list per lane (the list's share of the
            // queue as the host last saw it, `list_share`) the grid shrinks with the queue, down to one wavefront per SIMD
            // (cornell_model at 8 to 32 spp, launches of 20 M rays and fewer: - 5 %).
            const double share = (list_share && a.wl) ? std::min(1.0, std::max(0.0, list_share[st.wide])) : 1.0;
            const uint32_t listed = (uint32_t)((double)n_upper * share);
            const uint32_t wanted = std::max(s->wide_waves / 4u, listed / (TRAV_BLOCK * per_lane));
            const uint32_t grid = std::min(waves, std::min(wanted, s->wide_waves / wdiv));
            a.chunk = std::max(64u, std::min(RT_CHUNK, (n_upper / std::max(1u, s->wide_waves) + 63u) / 64u * 64u));
            a.thr_fetch = tf; a.thr_step = ts;
            DWide dw = s->d_wide[(size_t)st.wide];
            dw.fast = (a_.fast || s->fast_bvh) ? 1u : 0u;
            if (count) hipLaunchKernelGGL((k_traverse_wide<false, true>), dim3(grid), dim3(TRAV_BLOCK), 0, s->stream, d, dw, a, work, nullptr, s->d_tallies);
            else if (prof) hipLaunchKernelGGL((k_traverse_wide<true>), dim3(grid), dim3(TRAV_BLOCK), 0, s->stream, d, dw, a, work, (unsigned long long*)(s->d_count + 64) + 16 * (st.wide & 3), nullptr);
            else {
                const uint32_t only = s->wide[(size_t)st.wide].only;
#define WIDE(K) hipLaunchKernelGGL((k_traverse_wide<false, false, K>), dim3(grid), dim3(TRAV_BLOCK), 0, s->stream, d, dw, a, work, nullptr, nullptr)
                if (only == RT_NODE_TRIANGLES) WIDE(RT_NODE_TRIANGLES);
                else if (only == RT_NODE_QUADS) WIDE(RT_NODE_QUADS);
                else if (only == RT_NODE_SPHERES) WIDE(RT_NODE_SPHERES);
                else WIDE(0);
#undef WIDE
            }
        } else if (st.kind == RT_STRETCH_LOCKSTEP) {
#define LOCKSTEP(C, M) hipLaunchKernelGGL((k_traverse_lockstep<C, M>), dim3(waves), dim3(TRAV_BLOCK), lds, s->stream, d, a, s->d_tallies)
            if (count) LOCKSTEP(true, RT_MEDIA_ANY);  // counting passes need no speed: one instantiation serves all scenes
            else if (s->media_mode == RT_MEDIA_NONE) LOCKSTEP(false, RT_MEDIA_NONE);
            else if (s->media_mode == RT_MEDIA_FLAT) LOCKSTEP(false, RT_MEDIA_FLAT);
            else LOCKSTEP(false, RT_MEDIA_ANY);
#undef LOCKSTEP
        } else {
            HIP_TRY(hipMemsetAsync(work, 0, 4, s->stream));  // the work counter of this launch
            const uint32_t grid = std::min(waves, s->persist_waves);
            // RT_CHUNK rays per hand-out when there is plenty of work; fewer (down to one wavefront's worth) when the
            // launch is small, so that the rays spread over the whole device instead of queueing up behind a few waves
            a.chunk = std::max(64u, std::min(RT_CHUNK, (n_upper / std::max(1u, s->persist_waves) + 63u) / 64u * 64u));
            DScene dp = d;
            dp.n_nodes = st.end;  // where the per-lane walk stops
#define PERSIST(C, M) hipLaunchKernelGGL((k_traverse_persistent<C, M>), dim3(grid), dim3(TRAV_BLOCK), lds, s->stream, dp, a, work, s->d_tallies)
            if (count) PERSIST(true, RT_MEDIA_ANY);
            else if (s->media_mode == RT_MEDIA_NONE) PERSIST(false, RT_MEDIA_NONE);
            else if (s->media_mode == RT_MEDIA_FLAT) PERSIST(false, RT_MEDIA_FLAT);
            else PERSIST(false, RT_MEDIA_ANY);
#undef PERSIST
        }
    }
    HIP_TRY(hipGetLastError());
    return RT_OK;
}

static int rt_intersect_batch_impl(const rt_scene* s_, uint64_t seed, const double* o3, const double* d3, const double* tmin,
                       const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim, double* t,
                       double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    if (n == 0) return RT_OK;
    if (!o3 || !d3 || !tmin || !tmax) return fail(RT_ERR_INVALID, "null ray arrays");
    if (n > 0x7FFFFFFFull) return fail(RT_ERR_LIMIT, "batch too large");
    HIP_TRY(hipSetDevice(s->device));
    // rays go through the same queue layout and the same traverse kernel a render uses
    std::vector<double> soa(6 * n);
    std::vector<uint32_t> pix(n), smp(n), ctr(n);
    for (uint64_t i = 0; i < n; ++i) {
        for (int c = 0; c < 3; ++c) { soa[c * n + i] = o3[3 * i + c]; soa[(3 + c) * n + i] = d3[3 * i + c]; }
        pix[i] = rng3 ? rng3[3 * i] : (uint32_t)i;
        smp[i] = rng3 ? rng3[3 * i + 1] : 0u;
        ctr[i] = rng3 ? rng3[3 * i + 2] : 0u;
    }
    std::vector<void*> tmp;
    int rc = RT_OK;
    double *d_soa, *d_tmin, *d_tmax, *d_t, *d_p, *d_n, *d_uv;
    uint4* d_hits;
    uint32_t *d_pix, *d_smp, *d_ctr, *d_prim, *d_cout;
    uint8_t *d_hit, *d_front;
    do {
        if ((rc = dev_copy_in(tmp, soa.data(), 6 * n, &d_soa))) break;
        if ((rc = dev_copy_in(tmp, tmin, n, &d_tmin))) break;
        if ((rc = dev_copy_in(tmp, tmax, n, &d_tmax))) break;
        if ((rc = dev_copy_in(tmp, pix.data(), n, &d_pix))) break;
        if ((rc = dev_copy_in(tmp, smp.data(), n, &d_smp))) break;
        if ((rc = dev_copy_in(tmp, ctr.data(), n, &d_ctr))) break;
        if ((rc = dev_alloc(tmp, true, n, &d_hits))) break;
        if ((rc = dev_alloc(tmp, hit != nullptr, n, &d_hit))) break;
        if ((rc = dev_alloc(tmp, prim != nullptr, n, &d_prim))) break;
        if ((rc = dev_alloc(tmp, t != nullptr, n, &d_t))) break;
        if ((rc = dev_alloc(tmp, p3 != nullptr, 3 * n, &d_p))) break;
        if ((rc = dev_alloc(tmp, n3 != nullptr, 3 * n, &d_n))) break;
        if ((rc = dev_alloc(tmp, uv2 != nullptr, 2 * n, &d_uv))) break;
        if ((rc = dev_alloc(tmp, front != nullptr, n, &d_front))) break;
        if ((rc = dev_alloc(tmp, ctr_out != nullptr, n, &d_cout))) break;
        TravArgs a;
        std::memset(&a, 0, sizeof a);
        a.q.ox = d_soa; a.q.oy = d_soa + n; a.q.oz = d_soa + 2 * n;
        a.q.dx = d_soa + 3 * n; a.q.dy = d_soa + 4 * n; a.q.dz = d_soa + 5 * n;
        a.q.pix = d_pix; a.q.smp = d_smp; a.q.ctr = d_ctr;
        a.hb.rec = d_hits;
        a.seed = seed;
        a.segs.n = (uint32_t)n;
        a.segs.nclass = 1u;
        a.tmin = d_tmin; a.tmax = d_tmax;
        if ((rc = launch_traverse(s, a, false, false, (uint32_t)n))) break;
        hipLaunchKernelGGL(k_finalize_batch, dim3((uint32_t)((n + 255) / 256)), dim3(256), 0, s->stream, s->d, a.q, a.hb,
                           (uint32_t)n, d_hit, d_prim, d_t, d_p, d_n, d_uv, d_front, d_cout);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
        if (e != hipSuccess) { rc = fail(RT_ERR_DEVICE, "intersect kernels: %s", hipGetErrorString(e)); break; }
#define BACK(hostp, devp, cnt) if (hostp && hipMemcpy(hostp, devp, (cnt) * sizeof(*hostp), hipMemcpyDeviceToHost) != hipSuccess) { rc = fail(RT_ERR_DEVICE, "copy back failed"); break; }
        BACK(hit, d_hit, n); BACK(prim, d_prim, n); BACK(t, d_t, n); BACK(p3, d_p, 3 * n); BACK(n3, d_n, 3 * n);
        BACK(uv2, d_uv, 2 * n); BACK(front, d_front, n); BACK(ctr_out, d_cout, n);
#undef BACK
    } while (0);
    free_list(tmp);
    return rc;
}

// rt_intersect_batch for callers that cannot allocate: rays and results stay in the caller's device memory, and the
// only device memory the kernels use beyond them is the static array above.
static int rt_intersect_batch_noalloc_impl(const rt_scene* s_, uint64_t seed, const double* od6, const double* tmin,
                                           const double* tmax, uint32_t* rng3, uint32_t n, uint8_t* hit, uint32_t* prim,
                                           double* t, double* p3, double* n3, double* uv2, uint8_t* front) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    if (n == 0) return RT_OK;
    if (n > RT_NOALLOC_MAX_RAYS)
        return fail(RT_ERR_LIMIT, "rt_intersect_batch_noalloc takes at most %u rays per call (%u given): its scratch memory is a static array",
                    RT_NOALLOC_MAX_RAYS, n);
    if (!od6 || !tmin || !tmax || !rng3) return fail(RT_ERR_INVALID, "null ray arrays");
    if (s->device < 0 || s->device >= RT_MAX_DEVICES || !g_scratch_ptr[s->device])
        return fail(RT_ERR_STATE, "no static scratch memory on device %d", s->device);
    HIP_TRY(hipSetDevice(s->device));
    unsigned char* const base = g_scratch_ptr[s->device];  // this device's copy of the array
    TravArgs a;
    std::memset(&a, 0, sizeof a);
    a.q.ox = const_cast<double*>(od6); a.q.oy = a.q.ox + n; a.q.oz = a.q.ox + 2 * (size_t)n;
    a.q.dx = a.q.ox + 3 * (size_t)n; a.q.dy = a.q.ox + 4 * (size_t)n; a.q.dz = a.q.ox + 5 * (size_t)n;
    a.q.pix = rng3; a.q.smp = rng3 + n; a.q.ctr = rng3 + 2 * (size_t)n;
    a.hb.rec = reinterpret_cast<uint4*>(base);
    uint32_t* const work = reinterpret_cast<uint32_t*>(base + 16 * (size_t)RT_NOALLOC_MAX_RAYS);
    a.seed = seed;
    a.segs.n = n;
    a.segs.nclass = 1u;
    a.tmin = tmin; a.tmax = tmax;
    int rc = launch_traverse(s, a, false, false, n, work);  // (clears the counter before every launch that uses it)
    if (rc) return rc;
    hipLaunchKernelGGL(k_finalize_batch, dim3((n + 255u) / 256u), dim3(256), 0, s->stream, s->d, a.q, a.hb, n, hit, prim, t, p3,
                       n3, uv2, front, (uint32_t*)nullptr);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(s->stream));
    return RT_OK;
}

static int ws_queue(rt_scene* s, Queue& q, size_t n) {
    double** f[9] = {&q.ox, &q.oy, &q.oz, &q.dx, &q.dy, &q.dz, &q.br, &q.bg, &q.bb};
    for (auto pp : f) {
        HIP_TRY(rt_hip_malloc((void**)pp, n * sizeof(double)));
        s->ws_allocs.push_back(*pp);
    }
    uint32_t** g[6] = {&q.pix, &q.smp, &q.slot, &q.born, &q.ctr, &q.sig};
    for (auto pp : g) {
        HIP_TRY(rt_hip_malloc((void**)pp, n * sizeof(uint32_t)));
        s->ws_allocs.push_back(*pp);
    }
    return RT_OK;
}
static int ensure_workspace(rt_scene* s, size_t paths, size_t lum, size_t hits, size_t lists = 0) {
    if (s->ws_paths >= paths && s->ws_lum >= lum && s->ws_hits >= hits && s->ws_lists >= lists) return RT_OK;
    free_list(s->ws_allocs);
    int rc;
    paths = std::max(paths, s->ws_paths);  // what is freed above is allocated again, never less
    lum = std::max(lum, s->ws_lum);
    hits = std::max(hits, s->ws_hits);
    lists = std::max(lists, s->ws_lists);
    s->ws_paths = s->ws_lum = s->ws_hits = s->ws_lists = 0;
    s->ws_bytes = 0;
    if (paths) {
        if ((rc = ws_queue(s, s->qa, paths))) return rc;
        if ((rc = ws_queue(s, s->qb, paths))) return rc;
    }
    if (hits) {
        HIP_TRY(rt_hip_malloc((void**)&s->hb.rec, hits * sizeof(uint4))); s->ws_allocs.push_back(s->hb.rec);
    }
    if (lists) {  // work lists are in use: a hit buffer for the second queue too, and the lists (one queue's length each)
        HIP_TRY(rt_hip_malloc((void**)&s->hb2.rec, hits * sizeof(uint4))); s->ws_allocs.push_back(s->hb2.rec);
        HIP_TRY(rt_hip_malloc((void**)&s->wl, lists * paths * sizeof(uint32_t))); s->ws_allocs.push_back(s->wl);
    }
    // the radiance of path slot k is three consecutive doubles at 3 k: the three stores of a finished path touch one
    // 64-byte sector (a second one for a quarter of the slots) instead of three
    HIP_TRY(rt_hip_malloc((void**)&s->lum_r, 3 * lum * sizeof(double))); s->ws_allocs.push_back(s->lum_r);
    s->lum_g = s->lum_r + 1;
    s->lum_b = s->lum_r + 2;
    s->ws_paths = paths;
    s->ws_lum = lum;
    s->ws_hits = hits;
    s->ws_lists = lists;
    s->ws_bytes = (uint64_t)paths * 192u + (uint64_t)hits * 16u + (lists ? (uint64_t)hits * 16u + (uint64_t)lists * paths * 4u : 0u) + (uint64_t)lum * 24u;
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

// Streaming wavefront schedule.  Samples are generated in batches of `spb` samples of every pixel.  One iteration =
// one traverse launch + one shade launch over ALL paths in flight, whatever their depth; whenever the queue has room
// the next batch is appended, so launches stay large while old paths die off (a launch over a nearly empty queue
// costs as much as its slowest ray).  A batch's radiances land in its own section of the luminance buffer and are
// added to the film, batch after batch and sample after sample, once the batch is `max_depth` iterations old or
// nothing is in flight: the film is the same in-order sum whatever the queue size.
static int rt_render_device_impl(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, void* film_dev,
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
    st.lockstep = s->has_bvh ? 0u : 1u;
    const bool count = (o->flags & RT_FLAG_COUNTERS) != 0;
    hipStream_t sm = s->stream;
    HIP_TRY(hipEventRecord(s->ev[0], sm));
    hipLaunchKernelGGL(k_fill, dim3((3u * npix + 255u) / 256u), dim3(256), 0, sm, film, (size_t)3 * npix,
                       (o->max_depth == 0 && o->spp_count > 0) ? 0.0 : -0.0);
    if (count) HIP_TRY(hipMemsetAsync(s->d_tallies, 0, 64, sm));
    HIP_TRY(hipMemsetAsync(s->d_count + 16, 0, 8, sm));  // work-list lengths, added up by the launches that use them

    const bool lean_paths = s->plain && s->media_mode == RT_MEDIA_NONE;
    const bool fused = !s->has_bvh && !(o->flags & RT_FLAG_QUEUES) && ((o->flags & RT_FLAG_WHOLE_PATHS) != 0 || lean_paths);
    if (o->max_depth > 0 && o->spp_count > 0 && fused) {
        // whole paths in one kernel, batch after batch; a batch is as many samples of every pixel as fit the budget
        const uint64_t budget = std::max<uint64_t>(o->batch_paths ? o->batch_paths : (32ull << 20), npix);
        const uint32_t spb = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, budget / npix), o->spp_count);
        const uint64_t bsize = (uint64_t)spb * npix;
        if (bsize > 0xFFFFFFFFull - RT_CHUNK * 0x10000ull) return fail(RT_ERR_LIMIT, "batch too large");
        if ((rc = ensure_workspace(s, 0, (size_t)bsize, 0))) return rc;
        const uint32_t nbatches = (o->spp_count + spb - 1) / spb;
        HIP_TRY(hipMemsetAsync(s->d_count + 8, 0, 16, sm));  // rays traced, deepest bounce
        const bool nocull = (o->flags & RT_FLAG_NO_CULL) != 0;
        DScene d = s->d;
        if (nocull) d.nodes = s->nodes_nocull;
        std::vector<hipEvent_t> evs(2 * (size_t)nbatches, nullptr);
        int rc2 = RT_OK;
        for (uint32_t b = 0; b < nbatches && rc2 == RT_OK; ++b) {
            const uint32_t ns = std::min(spb, o->spp_count - b * spb), nb = ns * npix;
            const uint32_t grid = std::min((nb + TRAV_BLOCK - 1) / TRAV_BLOCK, s->path_waves);
            hipError_t e = hipEventCreate(&evs[2 * b]);
            if (e == hipSuccess) e = hipEventCreate(&evs[2 * b + 1]);
            if (e == hipSuccess) e = hipMemsetAsync(s->d_count + 4, 0, 4, sm);
            if (e == hipSuccess) e = hipEventRecord(evs[2 * b], sm);
            if (e != hipSuccess) { rc2 = fail(RT_ERR_DEVICE, "path batch set-up: %s", hipGetErrorString(e)); break; }
#define PATH_ARGS d, *cam, o->seed, npix, o->spp_begin + b * spb, nb, o->max_depth, o->background[0], o->background[1], \
                  o->background[2], s->lum_r, s->lum_g, s->lum_b, s->d_count + 4, (unsigned long long*)(s->d_count + 8), s->d_tallies
#define PATHS(C, M, P) hipLaunchKernelGGL((k_paths_lockstep<C, M, P>), dim3(grid), dim3(TRAV_BLOCK), levels_lds(s, TRAV_BLOCK), sm, PATH_ARGS)
            if (count) PATHS(true, RT_MEDIA_ANY, false);  // counting passes need no speed: one instantiation serves all scenes
            else if (s->media_mode == RT_MEDIA_NONE) { if (s->plain) PATHS(false, RT_MEDIA_NONE, true); else PATHS(false, RT_MEDIA_NONE, false); }
            else if (s->media_mode == RT_MEDIA_FLAT) { if (s->plain) PATHS(false, RT_MEDIA_FLAT, true); else PATHS(false, RT_MEDIA_FLAT, false); }
            else { if (s->plain) PATHS(false, RT_MEDIA_ANY, true); else PATHS(false, RT_MEDIA_ANY, false); }
#undef PATHS
#undef PATH_ARGS
            e = hipGetLastError();
            if (e == hipSuccess) e = hipEventRecord(evs[2 * b + 1], sm);
            if (e != hipSuccess) { rc2 = fail(RT_ERR_DEVICE, "path kernel: %s", hipGetErrorString(e)); break; }
            hipLaunchKernelGGL(k_film_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, sm, s->lum_r, s->lum_g, s->lum_b,
                               npix, ns, film);
            st.paths += nb;
            st.path_launches++;
        }
        hipError_t e = hipStreamSynchronize(sm);
        if (rc2 == RT_OK && e != hipSuccess) rc2 = fail(RT_ERR_DEVICE, "path kernels: %s", hipGetErrorString(e));
        for (uint32_t b = 0; b < nbatches; ++b) {
            float ms = 0;
            if (rc2 == RT_OK && evs[2 * b + 1] && hipEventElapsedTime(&ms, evs[2 * b], evs[2 * b + 1]) == hipSuccess) st.ms_paths += ms;
            if (evs[2 * b]) (void)hipEventDestroy(evs[2 * b]);
            if (evs[2 * b + 1]) (void)hipEventDestroy(evs[2 * b + 1]);
        }
        if (rc2 != RT_OK) return rc2;
        unsigned long long h[2];
        HIP_TRY(hipMemcpy(h, s->d_count + 8, 16, hipMemcpyDeviceToHost));
        st.rays = h[0];
        st.bounces = (uint32_t)h[1] + 1u;
    } else if (o->max_depth > 0 && o->spp_count > 0) {
        // The queue is a row of regions of qcap slots: the rays that go on land in the region of their signature class
        // (trees without BVH entries; one class otherwise), new camera rays in a region of their own behind those.
        // Trees without BVH entries: one bounce is one kernel (walk, shade and compaction together) or, on request
        // and in counting passes (those kernels carry the counters), a traverse and a shade launch.
        const bool lockstep = !s->has_bvh;
        const bool bounce = lockstep && !count && !(o->flags & RT_FLAG_SPLIT_BOUNCE);
        const uint32_t nclass = lockstep ? 1u << s->sig.n : 1u;
        const bool nocull = (o->flags & RT_FLAG_NO_CULL) != 0;
        // work lists for the launches that walk wide trees: only where more than one launch makes up a bounce (a
        // lone tree's bounds are the scene's: every ray reaches them)
        const bool gated = !lockstep && !nocull && s->gate.n > 0 && s->stretches.size() > 1;
        // Paths in flight: 128 M by default, 256 M for a tree with BVH entries (whose launches hand rays to resident
        // wavefronts and end with a tail of few busy lanes: the larger the launch, the less that weighs) -- as long as the
        // workspace that takes (two queues of 96 bytes a slot, closest hits, work lists, a radiance slot per path of the
        // batches in flight) stays within two fifths of the device memory that is free.
        struct Plan { uint32_t spb, nbatches, nslots; uint64_t bsize, qcap, rows, bytes; };
        auto plan = [&](uint64_t want) {
            Plan p;
            p.spb = (uint32_t)std::max<uint64_t>(1, (want / 4) / npix);  // samples per batch
            p.spb = std::min(p.spb, o->spp_count);
            p.bsize = (uint64_t)p.spb * npix;                             // paths per full batch
            // no queue larger than the paths this call traces
            p.qcap = std::max<uint64_t>(std::min<uint64_t>(want, (uint64_t)npix * o->spp_count), p.bsize);
            p.nbatches = (o->spp_count + p.spb - 1) / p.spb;
            p.nslots = std::min<uint32_t>(o->max_depth + 1u, p.nbatches);
            p.rows = (uint64_t)nclass * p.qcap + p.bsize;
            p.bytes = p.rows * (192u + (bounce ? 0u : 16u) + (gated ? 16u + 4u * s->gate.n : 0u)) + p.bsize * p.nslots * 24u;
            return p;
        };
        uint64_t want = std::max<uint64_t>(o->batch_paths ? o->batch_paths : (s->has_bvh ? (256ull << 20) : (128ull << 20)), npix);
        if (!o->batch_paths) {
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) == hipSuccess) {
                // (what this scene's workspace holds already is freed before a larger one is allocated)
                const uint64_t budget = ((uint64_t)free_b + s->ws_bytes) / 5u * 2u;
                while (want > npix && want > (1ull << 20) && plan(want).bytes > std::max<uint64_t>(budget, s->ws_bytes)) want = want / 4u * 3u;
            }
        }
        const Plan pl = plan(want);
        const uint32_t spb = pl.spb, nbatches = pl.nbatches, nslots = pl.nslots;
        const uint64_t bsize = pl.bsize, qcap = pl.qcap, rows = pl.rows;
        if (rows > 0x7FFFFFFFull || bsize * nslots > 0xFFFFFFFFull) return fail(RT_ERR_LIMIT, "queue too large");
        if ((rc = ensure_workspace(s, (size_t)rows, (size_t)(bsize * nslots), bounce ? 0 : (size_t)rows, gated ? s->gate.n : 0))) return rc;
        Gates gates;
        std::memset(&gates, 0, sizeof gates);
        // a first stretch that goes to the lockstep kernel visits every ray: it starts from "no hit" itself, and the
        // kernels that make the rays need not write that
        const bool first_fresh = s->stretches[0].kind == RT_STRETCH_LOCKSTEP;
        // Two shade launches per bounce (k_shade's MODE) where some materials are plain and some are not -- as long as
        // the hits on the others stay few: the host sees how many there were a few iterations late, and goes back to
        // one launch for everything when more than a third of a bounce's rays were left to the second.
        bool any_plain = false;
        for (const rt_material& m : s->mats)
            any_plain = any_plain || ((m.kind == RT_MAT_LAMBERTIAN || m.kind == RT_MAT_DIFFUSE_LIGHT || m.kind == RT_MAT_ISOTROPIC) &&
                                      s->texs[m.texture].kind == RT_TEX_SOLID);
        bool two_shades = !lockstep && !s->plain && any_plain && !count && !(o->flags & RT_FLAG_SPLIT_BOUNCE);
        // the trees first, the quads of stretch 0 inside the shade kernel (counting passes keep the list's own order)
        const bool walls = gated && s->walls && !count && !(o->flags & RT_FLAG_SPLIT_BOUNCE);
        if (walls) for (uint32_t i = s->stretches[0].first; i < s->stretches[0].end; ++i) st.shade_quads += s->nodes[i].count;
        if (gated) {
            gates.info = s->d_gate; gates.lists = s->wl; gates.stride = (uint32_t)s->ws_paths; gates.n = s->gate.n;
            gates.init_hits = (first_fresh || walls) ? 0u : 1u;
            gates.tag = walls ? 1u : 0u;
        }
        HitBuf hb_in = s->hb, hb_out = gated ? s->hb2 : s->hb;
        auto batch_samples = [&](uint32_t b) { return std::min(spb, o->spp_count - b * spb); };
        // How many rays each launch leaves behind is counted on the device, and the next launch reads it there.  The
        // host reads those counts RT_LAG iterations late: until then it works with upper bounds (a ray count only
        // falls from one iteration to the next, new camera rays aside), which size the grids and decide when the next
        // batch fits; workgroups that find no rays leave at once.  So the device never waits for the host.
        struct Launched { uint32_t fresh; uint32_t upper_in; };
        std::vector<Launched> hist;            // per iteration queued
        std::vector<uint32_t> born_at(nbatches, 0);
        Queue qin = s->qa, qout = s->qb;
        uint32_t injected = 0, accumulated = 0, consumed = 0, last_busy = 0;
        uint32_t known_out = 0;                // exact number of rays iteration `consumed - 1` left behind
        double list_share[RT_MAX_GATES];       // of the rays an iteration left behind, the part in each tree's list (as last seen)
        for (double& v : list_share) v = 1.0;
        bool finished = false;                 // the tail went to k_finish
        HIP_TRY(hipMemsetAsync(s->d_ring, 0, RT_RING_WORDS * 4, sm));  // nothing waits before iteration 0
        auto consume = [&]() -> int {          // read what iteration `consumed` left behind
            const uint32_t j = consumed, r = j % RT_RING, rn = (j + 1u) % RT_RING;
            HIP_TRY(hipEventSynchronize(s->ring_ev[r][3]));
            const uint32_t n_in = known_out + hist[j].fresh;
            st.rays += n_in;
            if (n_in) last_busy = j + 1u;
            known_out = 0;
            for (uint32_t c = 0; c < nclass; ++c) known_out += s->h_ring[rn * RT_RING_WORDS + RT_COUNT_STRIDE * c];
            if (two_shades && n_in >= 4096u && (uint64_t)s->h_ring[rn * RT_RING_WORDS + RT_DEFER_WORD] * 3u > n_in) two_shades = false;
            if (gated && known_out >= 4096u)
                for (uint32_t g = 0; g < s->gate.n; ++g)
                    list_share[g] = (double)s->h_ring[rn * RT_RING_WORDS + RT_COUNT_STRIDE * (4u + g)] / (double)known_out;
            float ms = 0;
            if (bounce) {
                HIP_TRY(hipEventElapsedTime(&ms, s->ring_ev[r][0], s->ring_ev[r][2]));
                st.ms_bounce += ms;
                st.bounce_launches++;
            } else {
                HIP_TRY(hipEventElapsedTime(&ms, s->ring_ev[r][0], s->ring_ev[r][1]));
                st.ms_traverse += ms;
                HIP_TRY(hipEventElapsedTime(&ms, s->ring_ev[r][1], s->ring_ev[r][2]));
                st.ms_shade += ms;
                st.traverse_launches++;
            }
            consumed++;
            return RT_OK;
        };
        for (uint32_t iter = 0;; ++iter) {
            if (iter > RT_BORN_MASK) return fail(RT_ERR_LIMIT, "more than %u iterations", RT_BORN_MASK);
            if (iter >= RT_LAG && consumed + RT_LAG <= iter && (rc = consume())) return rc;
            // no more rays than this can be waiting: the last exact count plus the camera rays generated since
            uint64_t upper = known_out;
            for (uint32_t j = consumed; j < iter; ++j) upper += hist[j].fresh;
            // batches all of whose paths are finished go to the film, oldest first (a path lives max_depth iterations
            // at most; and once a launch is known to have left nothing behind, everything generated before it is done)
            while (accumulated < injected &&
                   (iter - born_at[accumulated] >= o->max_depth || (known_out == 0 && consumed > born_at[accumulated]))) {
                const uint32_t slot0 = (uint32_t)((accumulated % nslots) * bsize);
                hipLaunchKernelGGL(k_film_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, sm, s->lum_r + 3 * (size_t)slot0,
                                   s->lum_g + 3 * (size_t)slot0, s->lum_b + 3 * (size_t)slot0, npix, batch_samples(accumulated), film);
                accumulated++;
            }
            uint32_t fresh = 0;  // camera rays generated for this iteration
            if (injected < nbatches && upper + bsize <= qcap) {
                // a slot is reused only after its previous batch has been added to the film
                if (injected >= nslots && accumulated + nslots <= injected) return fail(RT_ERR_STATE, "luminance slot still in use");
                const uint32_t ns = batch_samples(injected), nb = ns * npix;
                const uint32_t slot0 = (uint32_t)((injected % nslots) * bsize);
                hipLaunchKernelGGL(k_raygen, dim3((nb + 255u) / 256u), dim3(256), 0, sm, s->d_sig, *cam, o->seed, npix,
                                   o->spp_begin + injected * spb, nb, nclass * (uint32_t)qcap, slot0, iter, qin, gates, hb_in,
                                   s->d_ring + (iter % RT_RING) * RT_RING_WORDS);
                born_at[injected] = iter;
                injected++;
                fresh = nb;
                st.paths += nb;
            }
            const uint32_t upper_in = (uint32_t)(upper + fresh);
            if (upper_in == 0) {
                if (injected == nbatches) break;  // nothing in flight and nothing left to generate
                return fail(RT_ERR_STATE, "queue empty with batches left");
            }
            const uint32_t r = iter % RT_RING, rn = (iter + 1u) % RT_RING;
#if defined(RT_DEV_KNOBS)
            static const uint32_t finish_bvh = getenv("RT_FINISH_BVH") ? (uint32_t)atoi(getenv("RT_FINISH_BVH")) : RT_FINISH_MAX_BVH;
#else
            const uint32_t finish_bvh = RT_FINISH_MAX_BVH;
#endif
            if ((bounce || (!lockstep && !count)) && injected == nbatches && upper_in <= (lockstep ? RT_FINISH_MAX : finish_bvh)) {
                // the tail: every path that is left is followed to its end in one launch
                DScene d = s->d;
                if (nocull) { d.nodes = s->nodes_nocull; d.n_variants = 1u; }
                Segs segs;
                std::memset(&segs, 0, sizeof segs);
                segs.in_count = s->d_ring + r * RT_RING_WORDS;
                segs.nclass = nclass;
                segs.stride = (uint32_t)qcap;
                segs.fresh = fresh;
                // (a deep tree's interval slots for 256 lanes would not fit a workgroup's LDS; the kernel has no
                // workgroup-wide step, so one wavefront per workgroup serves as well)
                const uint32_t fblock = levels_lds(s, BOUNCE_BLOCK) > 48u * 1024u ? 64u : (uint32_t)BOUNCE_BLOCK;
                // no more workgroups than the device holds at once (two of 256 lanes per CU): the lanes take the rays
                // from the queue as they get free
#if defined(RT_DEV_KNOBS)
                static const bool no_lean = getenv("RT_FINISH_GENERIC") != nullptr;
#else
                const bool no_lean = false;
#endif
                const int lean = (lockstep || nocull || no_lean || fblock != (uint32_t)BOUNCE_BLOCK) ? 0 : s->lean;
#if defined(RT_DEV_KNOBS)
                static const bool no_light = getenv("RT_FINISH_NOLIGHT") != nullptr;
#else
                const bool no_light = false;
#endif
                const bool light_tex = s->light_tex && !no_light;
                const uint32_t resident = s->n_cus * (fblock == 64u ? 8u : 2u) * (uint32_t)RT_FINISH_WAVES(lean, s->plain, light_tex) / 2u;
                const uint32_t blocks = std::min((upper_in + fblock - 1) / fblock, resident);
                HIP_TRY(hipMemsetAsync(s->d_count + 8, 0, 16, sm));
                HIP_TRY(hipMemsetAsync(s->d_count + 4, 0, 4, sm));  // the counter the rays are handed out with
                HIP_TRY(hipEventRecord(s->ev[1], sm));
#define FINISH_ARGS d, o->seed, segs, qin, s->lum_r, s->lum_g, s->lum_b, o->background[0], o->background[1], \
                    o->background[2], iter, o->max_depth, (unsigned long long*)(s->d_count + 8), s->dev_stretches, \
                    (uint32_t)((lockstep || nocull) ? 0u : s->stretches.size()), s->dev_wides, \
                    (uint32_t)(((o->flags & RT_FLAG_FAST_BVH) || s->fast_bvh) ? 1u : 0u), s->d_count + 4, \
                    (gates.tag ? RT_BORN_MASK : 0xFFFFFFFFu)
#define FINISH(LS, M, P) hipLaunchKernelGGL((k_finish<LS, M, P>), dim3(blocks), dim3(fblock), \
                                            levels_lds(s, fblock), sm, FINISH_ARGS)
#define LEAN(P, K) hipLaunchKernelGGL((k_finish<false, RT_MEDIA_NONE, P, K>), dim3(blocks), dim3(fblock), levels_lds(s, fblock), sm, FINISH_ARGS)
#define LEAN_LIGHT(K) hipLaunchKernelGGL((k_finish<false, RT_MEDIA_NONE, false, K, true>), dim3(blocks), dim3(fblock), levels_lds(s, fblock), sm, FINISH_ARGS)
                if (lean == RT_NODE_TRIANGLES) { if (s->plain) LEAN(true, RT_NODE_TRIANGLES); else if (light_tex) LEAN_LIGHT(RT_NODE_TRIANGLES); else LEAN(false, RT_NODE_TRIANGLES); }
                else if (lean == RT_NODE_SPHERES) { if (s->plain) LEAN(true, RT_NODE_SPHERES); else if (light_tex) LEAN_LIGHT(RT_NODE_SPHERES); else LEAN(false, RT_NODE_SPHERES); }
                else if (!lockstep && s->media_mode == RT_MEDIA_NONE) { if (s->plain) FINISH(false, RT_MEDIA_NONE, true); else FINISH(false, RT_MEDIA_NONE, false); }
                else if (!lockstep) { if (s->plain) FINISH(false, RT_MEDIA_ANY, true); else FINISH(false, RT_MEDIA_ANY, false); }
                else if (s->media_mode == RT_MEDIA_NONE) { if (s->plain) FINISH(true, RT_MEDIA_NONE, true); else FINISH(true, RT_MEDIA_NONE, false); }
                else if (s->media_mode == RT_MEDIA_FLAT) { if (s->plain) FINISH(true, RT_MEDIA_FLAT, true); else FINISH(true, RT_MEDIA_FLAT, false); }
                else { if (s->plain) FINISH(true, RT_MEDIA_ANY, true); else FINISH(true, RT_MEDIA_ANY, false); }
#undef LEAN_LIGHT
#undef LEAN
#undef FINISH
#undef FINISH_ARGS
                HIP_TRY(hipEventRecord(s->ev[2], sm));
                HIP_TRY(hipGetLastError());
                finished = true;
                break;
            }
            hist.push_back({fresh, upper_in});
            uint32_t* const out_count = s->d_ring + rn * RT_RING_WORDS;
            Segs segs;
            std::memset(&segs, 0, sizeof segs);
            segs.in_count = s->d_ring + r * RT_RING_WORDS;
            segs.nclass = nclass;
            segs.stride = (uint32_t)qcap;
            segs.fresh = fresh;
            HIP_TRY(hipMemsetAsync(out_count, 0, RT_RING_WORDS * 4, sm));
            HIP_TRY(hipEventRecord(s->ring_ev[r][0], sm));
            if (bounce) {
                DScene d = s->d;
                if (nocull) { d.nodes = s->nodes_nocull; d.n_variants = 1u; }
                const uint32_t blocks = (upper_in + BOUNCE_BLOCK - 1) / BOUNCE_BLOCK + nclass + 1u;
#define BOUNCE_ARGS d, s->d_sig, o->seed, segs, qin, qout, out_count, s->lum_r, s->lum_g, s->lum_b, o->background[0], \
                    o->background[1], o->background[2], iter, o->max_depth
#define BOUNCE(M, P) hipLaunchKernelGGL((k_bounce_lockstep<M, P>), dim3(blocks), dim3(BOUNCE_BLOCK), \
                                        levels_lds(s, BOUNCE_BLOCK), sm, BOUNCE_ARGS)
                if (s->media_mode == RT_MEDIA_NONE) { if (s->plain) BOUNCE(RT_MEDIA_NONE, true); else BOUNCE(RT_MEDIA_NONE, false); }
                else if (s->media_mode == RT_MEDIA_FLAT) { if (s->plain) BOUNCE(RT_MEDIA_FLAT, true); else BOUNCE(RT_MEDIA_FLAT, false); }
                else { if (s->plain) BOUNCE(RT_MEDIA_ANY, true); else BOUNCE(RT_MEDIA_ANY, false); }
#undef BOUNCE
#undef BOUNCE_ARGS
            } else {
                TravArgs ta;
                std::memset(&ta, 0, sizeof ta);
                ta.q = qin; ta.hb = hb_in; ta.seed = o->seed; ta.segs = segs;
                if (gated) {
                    ta.resume = first_fresh ? 0u : 1u; ta.wl = s->wl; ta.wl_count = segs.in_count;
                    ta.list_total = (unsigned long long*)(s->d_count + 16);
                }
                ta.fast = (o->flags & RT_FLAG_FAST_BVH) ? 1u : 0u;
                if (walls) ta.resume = 1u;  // (from t = infinity, written where the ray was made)
                if ((rc = launch_traverse(s, ta, count, nocull, upper_in, nullptr, walls ? 1 : 0, gated ? list_share : nullptr))) return rc;
                HIP_TRY(hipEventRecord(s->ring_ev[r][1], sm));
                const uint32_t sblocks = (upper_in + SHADE_BLOCK - 1) / SHADE_BLOCK + nclass + 1u;
#define SHADE_ARGS s->d, s->d_sig, o->seed, segs, qin, hb_in, qout, out_count, s->lum_r, s->lum_g, s->lum_b, o->background[0], \
                   o->background[1], o->background[2], iter, o->max_depth, gates, hb_out, wl, qin.sig, out_count + RT_DEFER_WORD
#define SHADE(P, W, M, G) hipLaunchKernelGGL((k_shade<P, W, M>), dim3(G), dim3(SHADE_BLOCK), 0, sm, SHADE_ARGS)
                Walls wl;
                wl.first = s->stretches[0].first; wl.end = s->stretches[0].end;
                if (s->plain) {
                    if (walls) SHADE(true, true, RT_SHADE_ALL, sblocks); else SHADE(true, false, RT_SHADE_ALL, sblocks);
                } else if (two_shades) {
                    // plain materials first, the hits on the others noted in a list (the input queue's signature array:
                    // trees with BVH entries keep no signatures) for the full code
                    const uint32_t dblocks = sblocks;
                    if (walls) { SHADE(true, true, RT_SHADE_PLAIN_DEFER, sblocks); SHADE(false, true, RT_SHADE_DEFERRED, dblocks); }
                    else { SHADE(true, false, RT_SHADE_PLAIN_DEFER, sblocks); SHADE(false, false, RT_SHADE_DEFERRED, dblocks); }
                } else {
                    if (walls) SHADE(false, true, RT_SHADE_ALL, sblocks); else SHADE(false, false, RT_SHADE_ALL, sblocks);
                }
#undef SHADE
#undef SHADE_ARGS
            }
            HIP_TRY(hipEventRecord(s->ring_ev[r][2], sm));
            HIP_TRY(hipMemcpyAsync(s->h_ring + rn * RT_RING_WORDS, out_count, RT_RING_WORDS * 4, hipMemcpyDeviceToHost, sm));
            HIP_TRY(hipEventRecord(s->ring_ev[r][3], sm));
            HIP_TRY(hipGetLastError());
            std::swap(qin, qout);
            std::swap(hb_in, hb_out);
        }
        while (consumed < hist.size()) if ((rc = consume())) return rc;
        if (finished) {
            unsigned long long h[2];
            HIP_TRY(hipMemcpyAsync(h, s->d_count + 8, 16, hipMemcpyDeviceToHost, sm));
            HIP_TRY(hipStreamSynchronize(sm));
            st.rays += h[0];
            last_busy = std::max(last_busy, (uint32_t)h[1]);
            float ms = 0;
            HIP_TRY(hipEventElapsedTime(&ms, s->ev[1], s->ev[2]));
            st.ms_finish += ms;
            st.finish_launches++;
        } else if (known_out != 0) {
            return fail(RT_ERR_STATE, "rays left in the queue");
        }
        while (accumulated < injected) {  // (a render of depth 0 generates nothing; everything else was added above)
            const uint32_t slot0 = (uint32_t)((accumulated % nslots) * bsize);
            hipLaunchKernelGGL(k_film_accumulate, dim3((npix + 255u) / 256u), dim3(256), 0, sm, s->lum_r + 3 * (size_t)slot0,
                               s->lum_g + 3 * (size_t)slot0, s->lum_b + 3 * (size_t)slot0, npix, batch_samples(accumulated), film);
            accumulated++;
        }
        st.bounces = std::min(last_busy, o->max_depth);
    }
    HIP_TRY(hipEventRecord(s->ev[3], sm));
    HIP_TRY(hipStreamSynchronize(sm));
    HIP_TRY(hipGetLastError());
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, s->ev[0], s->ev[3]));
    st.ms_total = ms;
    if (s->has_bvh) {
        unsigned long long lr = 0;
        HIP_TRY(hipMemcpy(&lr, s->d_count + 16, 8, hipMemcpyDeviceToHost));
        st.list_rays = lr;
    }
    if (count) {
        unsigned long long h[8];
        HIP_TRY(hipMemcpy(h, s->d_tallies, 64, hipMemcpyDeviceToHost));
        st.node_tests = h[0];
        st.prim_tests = h[1];
        st.prim_bytes = h[2];
        st.cull_tests = h[3];
        st.prim_flops = h[4];
        st.wide_tests = h[5];
    }
#if defined(RT_DEV_KNOBS)
    if (getenv("RT_WIDE_PROF") != nullptr) {
        unsigned long long h[64];
        HIP_TRY(hipMemcpy(h, s->d_count + 64, sizeof h, hipMemcpyDeviceToHost));
        for (size_t w = 0; w < std::min<size_t>(s->wide.size(), 4); ++w) {
            const unsigned long long* p = h + 16 * w;
            if (!p[10]) continue;
            fprintf(stderr, "wide tree %zu: waves %llu, cycles/wave %.0f | hand-out: %.1f%% of cycles, %llu rounds, %.1f lanes, %.0f cycles each | "
                    "node: %.1f%%, %llu steps, %.1f lanes, %.0f cycles | prim: %.1f%%, %llu steps, %.1f lanes, %.0f cycles\n", w, p[10],
                    (double)p[9] / p[10], 100.0 * p[0] / p[9], p[3], (double)p[6] / std::max(1ull, p[3]), (double)p[0] / std::max(1ull, p[3]),
                    100.0 * p[1] / p[9], p[4], (double)p[7] / std::max(1ull, p[4]), (double)p[1] / std::max(1ull, p[4]),
                    100.0 * p[2] / p[9], p[5], (double)p[8] / std::max(1ull, p[5]), (double)p[2] / std::max(1ull, p[5]));
        }
        HIP_TRY(hipMemset(s->d_count + 64, 0, sizeof h));
    }
#endif
    if (stats) *stats = st;
    return RT_OK;
}

static int rt_render_impl(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, double* film_rgb, rt_stats* stats) {
    rt_scene* s = const_cast<rt_scene*>(s_);
    NEED_COMMITTED(s);
    int rc = check_render_args(s, cam, o);
    if (rc) return rc;
    if (!film_rgb) return fail(RT_ERR_INVALID, "null film");
    HIP_TRY(hipSetDevice(s->device));
    size_t bytes = (size_t)3 * cam->width * cam->height * sizeof(double);
    void* d_film = nullptr;
    HIP_TRY(rt_hip_malloc(&d_film, bytes));
    rc = rt_render_device_impl(s, cam, o, d_film, stats);
    if (rc == RT_OK && hipMemcpy(film_rgb, d_film, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "film copy back failed");
    (void)hipFree(d_film);
    return rc;
}

// RCCL for the film reduction of rt_render_multi, loaded when first needed: a host that renders on one GPU need not
// have the library, and a process that already holds a copy (PyTorch brings its own) gets that one.
extern "C++" {
static std::atomic<uint64_t> g_rccl_reductions{0};
namespace {
struct Rccl {
    void* lib = nullptr;
    ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*Reduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    const char* (*GetErrorString)(ncclResult_t) = nullptr;
    bool ok = false;
    std::mutex lock;
    std::map<std::vector<int>, std::vector<ncclComm_t>> comms;  // one communicator set per list of devices, kept
};
Rccl& rccl() {
    static Rccl r;
    static std::once_flag once;
    std::call_once(once, []() {
        const char* off = getenv("RT_MULTI_RCCL");
        if (off && off[0] == '0') return;
        for (const char* name : {"librccl.so.1", "librccl.so"}) {
            r.lib = dlopen(name, RTLD_NOW | RTLD_LOCAL);
            if (r.lib) break;
        }
        if (!r.lib) return;
#define RCCL_SYM(field, sym) r.field = reinterpret_cast<decltype(r.field)>(dlsym(r.lib, sym))
        RCCL_SYM(CommInitAll, "ncclCommInitAll"); RCCL_SYM(CommDestroy, "ncclCommDestroy"); RCCL_SYM(Reduce, "ncclReduce");
        RCCL_SYM(GroupStart, "ncclGroupStart"); RCCL_SYM(GroupEnd, "ncclGroupEnd"); RCCL_SYM(GetErrorString, "ncclGetErrorString");
#undef RCCL_SYM
        r.ok = r.CommInitAll && r.CommDestroy && r.Reduce && r.GroupStart && r.GroupEnd && r.GetErrorString;
    });
    return r;
}
// Sums the n device films into films[0] with one ncclReduce per device (root 0) in one group; the films live on the n
// distinct devices devs[].  RT_OK, or an error with the reason set; 1 when RCCL is not to be had (the caller then
// gathers the shards itself).
int rccl_reduce_films(const std::vector<int>& devs, const std::vector<void*>& films, size_t count) {
    Rccl& r = rccl();
    if (!r.ok) return 1;
    std::lock_guard<std::mutex> hold(r.lock);
    auto it = r.comms.find(devs);
    if (it == r.comms.end()) {
        std::vector<ncclComm_t> c(devs.size());
        const ncclResult_t e = r.CommInitAll(c.data(), (int)devs.size(), devs.data());
        if (e != ncclSuccess) return fail(RT_ERR_DEVICE, "ncclCommInitAll: %s", r.GetErrorString(e));
        it = r.comms.emplace(devs, c).first;
    }
    ncclResult_t e = r.GroupStart();
    for (size_t k = 0; k < devs.size() && e == ncclSuccess; ++k) {
        if (hipSetDevice(devs[k]) != hipSuccess) { (void)r.GroupEnd(); return fail(RT_ERR_DEVICE, "cannot select device %d", devs[k]); }
        e = r.Reduce(films[k], films[0], count, ncclDouble, ncclSum, 0, it->second[k], (hipStream_t)0);
    }
    const ncclResult_t e2 = r.GroupEnd();
    if (e == ncclSuccess) e = e2;
    if (e != ncclSuccess) return fail(RT_ERR_DEVICE, "ncclReduce: %s", r.GetErrorString(e));
    for (size_t k = 0; k < devs.size(); ++k) {
        if (hipSetDevice(devs[k]) != hipSuccess || hipStreamSynchronize((hipStream_t)0) != hipSuccess)
            return fail(RT_ERR_DEVICE, "waiting for the film reduction on device %d", devs[k]);
    }
    return RT_OK;
}
}  // namespace
}  // extern "C++"

static int rt_render_multi_impl(const rt_scene* const* scenes, int n, const rt_camera* cam, const rt_render_opts* o,
                    double* film_rgb, rt_stats* stats) {
    if (!scenes || n < 1 || n > 64) return fail(RT_ERR_INVALID, "bad scene list");
    for (int r = 0; r < n; ++r) {
        if (!scenes[r]) return fail(RT_ERR_INVALID, "null scene");
        if (!scenes[r]->committed) return fail(RT_ERR_STATE, "scene not committed");
        for (int q = 0; q < r; ++q)
            if (scenes[q] == scenes[r]) return fail(RT_ERR_INVALID, "the same scene handle twice (a handle is not re-entrant)");
    }
    int rc = check_render_args(scenes[0], cam, o);
    if (rc) return rc;
    if (!film_rgb) return fail(RT_ERR_INVALID, "null film");
    const size_t npix = (size_t)cam->width * cam->height, bytes = 3 * npix * sizeof(double);
    struct Shard { void* film = nullptr; rt_stats st; int rc = RT_OK; std::string err; };
    std::vector<Shard> sh((size_t)n);
    std::vector<std::thread> workers;
    const uint32_t base = o->spp_count / (uint32_t)n, extra = o->spp_count % (uint32_t)n;
    for (int r = 0; r < n; ++r) {
        workers.emplace_back([&, r]() {
            Shard& me = sh[(size_t)r];
            std::memset(&me.st, 0, sizeof me.st);
            rt_render_opts mine = *o;
            mine.spp_count = base + ((uint32_t)r < extra ? 1u : 0u);
            mine.spp_begin = o->spp_begin + (uint32_t)r * base + std::min<uint32_t>((uint32_t)r, extra);
            hipError_t e = hipSetDevice(scenes[r]->device);
            if (e == hipSuccess) e = rt_hip_malloc(&me.film, bytes);
            if (e != hipSuccess) { me.rc = RT_ERR_DEVICE; me.err = hipGetErrorString(e); return; }
            me.rc = rt_render_device_impl(scenes[r], cam, &mine, me.film, &me.st);
            if (me.rc != RT_OK) me.err = rt_last_error();  // the error text is per thread
        });
    }
    for (auto& w : workers) w.join();
    rc = RT_OK;
    for (int r = 0; r < n && rc == RT_OK; ++r)
        if (sh[(size_t)r].rc != RT_OK) rc = fail(sh[(size_t)r].rc, "shard %d: %s", r, sh[(size_t)r].err.c_str());
    // The films are summed on the first device: by RCCL (one reduce, root 0) when every scene has a device of its own,
    // else -- several scenes on one device, or no RCCL to load -- gathered there by peer copy and added in rank order.
    // (RT_MULTI_RCCL=0 keeps RCCL out; RT_MULTI_RCCL=force uses it for a single scene too.)
    const int dev0 = scenes[0]->device;
    bool reduced = false;
    if (rc == RT_OK) {
        std::vector<int> devs;
        std::vector<void*> films;
        bool distinct = true;
        for (int r = 0; r < n; ++r) {
            for (int q = 0; q < r; ++q) distinct = distinct && scenes[q]->device != scenes[r]->device;
            devs.push_back(scenes[r]->device);
            films.push_back(sh[(size_t)r].film);
        }
        const char* mode = getenv("RT_MULTI_RCCL");
        if (distinct && (n > 1 || (mode && std::strcmp(mode, "force") == 0))) {
            const int e = rccl_reduce_films(devs, films, 3 * npix);
            if (e == RT_OK) { reduced = true; g_rccl_reductions.fetch_add(1); }
            else if (e < 0) rc = e;
        }
    }
    void* tmp = nullptr;
    std::vector<double> staging;
    if (rc == RT_OK && hipSetDevice(dev0) != hipSuccess) rc = fail(RT_ERR_DEVICE, "cannot select device %d", dev0);
    if (rc == RT_OK && n > 1 && !reduced && rt_hip_malloc(&tmp, bytes) != hipSuccess) rc = fail(RT_ERR_DEVICE, "out of device memory");
    for (int r = 1; r < n && rc == RT_OK && !reduced; ++r) {
        const int dev = scenes[r]->device;
        hipError_t e = dev == dev0 ? hipMemcpy(tmp, sh[(size_t)r].film, bytes, hipMemcpyDeviceToDevice)
                                   : hipMemcpyPeer(tmp, dev0, sh[(size_t)r].film, dev, bytes);
        if (e != hipSuccess) {  // no peer path between the two devices: through the host
            (void)hipGetLastError();
            staging.resize(3 * npix);
            e = hipSetDevice(dev);
            if (e == hipSuccess) e = hipMemcpy(staging.data(), sh[(size_t)r].film, bytes, hipMemcpyDeviceToHost);
            if (e == hipSuccess) e = hipSetDevice(dev0);
            if (e == hipSuccess) e = hipMemcpy(tmp, staging.data(), bytes, hipMemcpyHostToDevice);
        }
        if (e == hipSuccess) {
            hipLaunchKernelGGL(k_film_add, dim3((uint32_t)((3 * npix + 255) / 256)), dim3(256), 0, 0, (double*)sh[0].film,
                               (const double*)tmp, 3 * npix);
            e = hipGetLastError();
            if (e == hipSuccess) e = hipDeviceSynchronize();
        }
        if (e != hipSuccess) rc = fail(RT_ERR_DEVICE, "gathering shard %d: %s", r, hipGetErrorString(e));
    }
    if (rc == RT_OK && hipMemcpy(film_rgb, sh[0].film, bytes, hipMemcpyDeviceToHost) != hipSuccess)
        rc = fail(RT_ERR_DEVICE, "film copy back failed");
    if (tmp) (void)hipFree(tmp);
    for (int r = 0; r < n; ++r)
        if (sh[(size_t)r].film) { (void)hipSetDevice(scenes[r]->device); (void)hipFree(sh[(size_t)r].film); }
    if (rc == RT_OK && stats) {
        rt_stats t;
        std::memset(&t, 0, sizeof t);
        for (int r = 0; r < n; ++r) {
            const rt_stats& a = sh[(size_t)r].st;
            t.paths += a.paths; t.rays += a.rays; t.node_tests += a.node_tests; t.prim_tests += a.prim_tests;
            t.cull_tests += a.cull_tests; t.prim_bytes += a.prim_bytes; t.prim_flops += a.prim_flops; t.wide_tests += a.wide_tests; t.list_rays += a.list_rays;
            t.ms_total = std::max(t.ms_total, a.ms_total); t.ms_traverse = std::max(t.ms_traverse, a.ms_traverse);
            t.ms_shade = std::max(t.ms_shade, a.ms_shade); t.ms_paths = std::max(t.ms_paths, a.ms_paths);
            t.ms_bounce = std::max(t.ms_bounce, a.ms_bounce); t.ms_finish = std::max(t.ms_finish, a.ms_finish);
            t.finish_launches += a.finish_launches;
            t.traverse_launches += a.traverse_launches; t.path_launches += a.path_launches;
            t.bounce_launches += a.bounce_launches;
            t.bounces = std::max(t.bounces, a.bounces); t.lockstep = a.lockstep; t.shade_quads = a.shade_quads;
        }
        *stats = t;
    }
    return rc;
}

static int rt_trace_samples_impl(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, uint64_t n,
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

static int rt_tonemap_rgb8_impl(const rt_scene* s_, const double* film_rgb, uint32_t width, uint32_t height, uint32_t spp,
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

static int rt_debug_arith_impl(int device, const double* a, const double* b, uint64_t n, double* quot, double* root, double* recip,
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


// ---- the exported entry points: no exception crosses the boundary, and the caller's current HIP device is the same
// after a call as before it
extern "C++" {
struct DeviceGuard {
    int prev = -1;
    DeviceGuard() { if (hipGetDevice(&prev) != hipSuccess) { prev = -1; (void)hipGetLastError(); } }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
template <class F>
static int guarded(F&& f) {
    DeviceGuard keep;
    try {
        return f();
    } catch (const std::bad_alloc&) {
        return fail(RT_ERR_LIMIT, "out of host memory");
    } catch (const std::exception& e) {
        return fail(RT_ERR_INVALID, "unexpected failure: %s", e.what());
    } catch (...) {
        return fail(RT_ERR_INVALID, "unexpected failure");
    }
}
}  // extern "C++"
int rt_scene_set_materials(rt_scene* s, const rt_material* m, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_materials_impl(s, m, n); });
}
int rt_scene_set_textures(rt_scene* s, const rt_texture* t, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_textures_impl(s, t, n); });
}
int rt_scene_set_image(rt_scene* s, uint32_t slot, const uint8_t* rgb8, uint32_t w, uint32_t h) {
    return guarded([&]() -> int { return rt_scene_set_image_impl(s, slot, rgb8, w, h); });
}
int rt_scene_set_perlin(rt_scene* s, uint32_t slot, const double* vec256x3, const uint32_t* perm3x256) {
    return guarded([&]() -> int { return rt_scene_set_perlin_impl(s, slot, vec256x3, perm3x256); });
}
int rt_scene_set_spheres(rt_scene* s, const double* cr, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_spheres_impl(s, cr, mat, id, n); });
}
int rt_scene_set_quads(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_quads_impl(s, rec16, mat, id, n); });
}
int rt_scene_set_triangles(rt_scene* s, const double* rec16, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_triangles_impl(s, rec16, mat, id, n); });
}
int rt_scene_set_quadrics(rt_scene* s, const double* rec10, const uint32_t* mat, const uint32_t* id, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_quadrics_impl(s, rec10, mat, id, n); });
}
int rt_scene_set_media(rt_scene* s, const rt_medium* m, uint32_t n) {
    return guarded([&]() -> int { return rt_scene_set_media_impl(s, m, n); });
}
int rt_scene_set_xforms(rt_scene* s, const rt_xform_op* ops, uint32_t n_ops, const rt_chain* chains, uint32_t n_chains) {
    return guarded([&]() -> int { return rt_scene_set_xforms_impl(s, ops, n_ops, chains, n_chains); });
}
int rt_scene_set_nodes(rt_scene* s, const rt_node* nodes, uint32_t n_nodes, const double* boxes6, uint32_t n_boxes) {
    return guarded([&]() -> int { return rt_scene_set_nodes_impl(s, nodes, n_nodes, boxes6, n_boxes); });
}
int rt_scene_set_cull_boxes(rt_scene* s, const double* tight6, uint32_t n_boxes) {
    return guarded([&]() -> int { return rt_scene_set_cull_boxes_impl(s, tight6, n_boxes); });
}
int rt_scene_commit(rt_scene* s, int device) {
    return guarded([&]() -> int { return rt_scene_commit_impl(s, device); });
}
int rt_intersect_batch(const rt_scene* s_, uint64_t seed, const double* o3, const double* d3, const double* tmin, const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim, double* t, double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out) {
    return guarded([&]() -> int { return rt_intersect_batch_impl(s_, seed, o3, d3, tmin, tmax, rng3, n, hit, prim, t, p3, n3, uv2, front, ctr_out); });
}
int rt_render_device(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, void* film_dev, rt_stats* stats) {
    return guarded([&]() -> int { return rt_render_device_impl(s_, cam, o, film_dev, stats); });
}
int rt_render(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, double* film_rgb, rt_stats* stats) {
    return guarded([&]() -> int { return rt_render_impl(s_, cam, o, film_rgb, stats); });
}
int rt_render_multi(const rt_scene* const* scenes, int n, const rt_camera* cam, const rt_render_opts* o, double* film_rgb, rt_stats* stats) {
    return guarded([&]() -> int { return rt_render_multi_impl(scenes, n, cam, o, film_rgb, stats); });
}
uint64_t rt_multi_rccl_reductions(void) { return g_rccl_reductions.load(); }
void rt_multi_shutdown(void) {
    Rccl& r = rccl();
    std::lock_guard<std::mutex> hold(r.lock);
    if (r.ok)
        for (auto& kv : r.comms)
            for (ncclComm_t c : kv.second) (void)r.CommDestroy(c);
    r.comms.clear();
}
uint64_t rt_debug_device_allocs(void) { return g_device_allocs.load(); }
int rt_trace_samples(const rt_scene* s_, const rt_camera* cam, const rt_render_opts* o, uint64_t n, const uint32_t* pixel, const uint32_t* sample, double* radiance3, uint32_t* nrays) {
    return guarded([&]() -> int { return rt_trace_samples_impl(s_, cam, o, n, pixel, sample, radiance3, nrays); });
}
int rt_tonemap_rgb8(const rt_scene* s_, const double* film_rgb, uint32_t width, uint32_t height, uint32_t spp, uint8_t* rgb8) {
    return guarded([&]() -> int { return rt_tonemap_rgb8_impl(s_, film_rgb, width, height, spp, rgb8); });
}
int rt_debug_arith(int device, const double* a, const double* b, uint64_t n, double* quot, double* root, double* recip, double* logv) {
    return guarded([&]() -> int { return rt_debug_arith_impl(device, a, b, n, quot, root, recip, logv); });
}

int rt_intersect_batch_noalloc(const rt_scene* s, uint64_t seed, const double* od6_dev, const double* tmin_dev,
                               const double* tmax_dev, uint32_t* rng3_dev, uint32_t n, uint8_t* hit_dev, uint32_t* prim_dev,
                               double* t_dev, double* p3_dev, double* n3_dev, double* uv2_dev, uint8_t* front_dev) {
    // (no exception guard: nothing in this path allocates, on the host or on the device)
    DeviceGuard keep;
    return rt_intersect_batch_noalloc_impl(s, seed, od6_dev, tmin_dev, tmax_dev, rng3_dev, n, hit_dev, prim_dev, t_dev, p3_dev,
                                           n3_dev, uv2_dev, front_dev);
}

}  // extern "C"
