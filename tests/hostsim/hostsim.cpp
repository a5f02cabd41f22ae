// hostsim.cpp -- TEST-ONLY: runs the per-lane device functions of raytracer_amd/csrc/rt_core.h on the CPU, one ray
// or path at a time, so that the flattened-tree walk can be compared with the oracle in a container without a GPU.
// It is never linked into librt_amd.so or loaded by the raytracer_amd package.
#include <cstdint>
#include <cstring>
#include <vector>

#include "../../raytracer_amd/csrc/rt_core.h"
#include "../../raytracer_amd/csrc/rt_records.h"
#include "../../raytracer_amd/csrc/rt_wide.h"

using namespace rt;

struct sim_scene {  // host pointers to the arrays of a flattened scene
    const rt_node* nodes; uint64_t n_nodes;
    const double* boxes; const double* spheres; const double* quads; const double* tris; const double* quadrics;
    const uint32_t* sphere_mat; const uint32_t* sphere_id; const uint32_t* quad_mat; const uint32_t* quad_id;
    const uint32_t* tri_mat; const uint32_t* tri_id; const uint32_t* quadric_mat; const uint32_t* quadric_id;
    const rt_medium* media; const rt_xform_op* ops; const rt_chain* chains; const rt_material* mats;
    const rt_texture* texs; const uint8_t* texels; const uint32_t* image_desc; const double* perlin_vec;
    const uint32_t* perlin_perm;
    const double* cull_boxes;  // one per box, or null
    uint64_t n_quads;
};

static std::vector<uint32_t> g_records, g_quad_axis;
static std::vector<float> g_leaf_boxes;
static int use_leaf_boxes = 1;  // 0: BVH leaves without cull boxes per pair of primitives
extern "C" void sim_set_leaf_boxes(int on) { use_leaf_boxes = on; }
static int use_cull = 1;      // 0: every cull box switched off, as under RT_FLAG_NO_CULL
static int use_one_pass = 1;  // 0: media always take their two boundary hits in two passes
static SigInfo g_sig;
static DScene to_dscene(const sim_scene* s) {
    DScene d;
    const std::vector<uint32_t> sig = signature_nodes(s->nodes, (size_t)s->n_nodes, RT_MAX_SIG);
    if (use_cull) {
        g_records = build_record_variants(s->nodes, (size_t)s->n_nodes, s->boxes, s->cull_boxes, use_one_pass != 0, sig);
        d.n_variants = 1u << sig.size();
    } else {
        g_records = build_records(s->nodes, (size_t)s->n_nodes, s->boxes, s->cull_boxes, false, use_one_pass != 0);
        d.n_variants = 1u;
    }
    std::memset(&g_sig, 0, sizeof g_sig);
    g_sig.n = (uint32_t)sig.size();
    g_leaf_boxes.clear();
    if (use_cull && use_leaf_boxes) build_leaf_boxes(s->nodes, (size_t)s->n_nodes, s->tris, s->quads, g_records, g_leaf_boxes);
    g_leaf_boxes.resize(g_leaf_boxes.size() + 48, 0.0f);
    d.leaf_sub = g_leaf_boxes.data();
    d.nodes = reinterpret_cast<const uint4*>(g_records.data()); d.n_nodes = (uint32_t)s->n_nodes; d.spheres = s->spheres; d.quads = s->quads;
    d.tris = s->tris; d.quadrics = s->quadrics; d.sphere_mat = s->sphere_mat; d.sphere_id = s->sphere_id;
    d.quad_mat = s->quad_mat; d.quad_id = s->quad_id; d.tri_mat = s->tri_mat; d.tri_id = s->tri_id;
    d.quadric_mat = s->quadric_mat; d.quadric_id = s->quadric_id; d.media = s->media; d.ops = s->ops; d.chains = s->chains;
    d.mats = s->mats; d.texs = s->texs; d.texels = s->texels; d.image_desc = s->image_desc; d.perlin_vec = s->perlin_vec;
    d.perlin_perm = s->perlin_perm;
    g_quad_axis = quad_axis_flags(s->quads, (size_t)s->n_quads);
    g_quad_axis.push_back(0u);
    d.quad_axis = g_quad_axis.data();
    for (size_t b = 0; b < sig.size(); ++b) signature_entry(g_sig, d, (uint32_t)b, s->nodes[sig[b]], s->boxes + 6 * (size_t)s->nodes[sig[b]].first);
    return d;
}

static int mode = 0;  // 0 generic walk, 1 lockstep walk, 2 per-lane state machine, 3 stretches (lockstep / wide tree / state machine)
static unsigned long long g_wide_stretches = 0, g_wide_steps = 0;
extern "C" unsigned long long sim_wide_stretches(void) { return g_wide_stretches; }
extern "C" unsigned long long sim_wide_steps(void) { return g_wide_steps; }
extern "C" void sim_set_uniform_mode(int on) { host_uniform_mode() = on != 0; }
extern "C" void sim_set_walk_mode(int m) { mode = m; }
extern "C" void sim_set_cull(int on) { use_cull = on; }
extern "C" void sim_set_one_pass(int on) { use_one_pass = on; }
static int use_fast = 0;  // wide trees: children nearest first (RT_FLAG_FAST_BVH)
extern "C" void sim_set_fast_bvh(int on) { use_fast = on; }
static int use_sig = 1;  // lockstep walk: 0 leaves every ray's signature at all ones (nothing is passed over)
extern "C" void sim_set_signatures(int on) { use_sig = on; }
extern "C" unsigned long long sim_fallbacks(void) { return host_fallback_count(); }

extern "C" int sim_intersect(const sim_scene* sc, uint64_t seed, const double* o3, const double* d3, const double* tmin,
                             const double* tmax, const uint32_t* rng3, uint64_t n, uint8_t* hit, uint32_t* prim, double* t,
                             double* p3, double* n3, double* uv2, uint8_t* front, uint32_t* ctr_out, uint64_t* tallies3) {
    DScene S = to_dscene(sc);
    double2 slots[RT_MAX_LEVELS];
    Levels L; L.base = slots; L.stride = 1;
    uint64_t nt = 0, pt = 0, ct = 0;
    std::vector<WideBvh> wide;
    std::vector<Stretch> stretches;
    if (mode == 3) {
        bool has_bvh = false;
        for (uint64_t i = 0; i < sc->n_nodes; ++i) has_bvh |= sc->nodes[i].kind == RT_NODE_BVH;
        stretches = build_stretches(sc->nodes, (uint32_t)sc->n_nodes, has_bvh, [&](uint32_t e) -> int {
            WideBvh w;
            if (!build_wide(sc->nodes, (size_t)sc->n_nodes, e, sc->boxes, sc->spheres, sc->quads, sc->tris, w)) return -1;
            wide.push_back(std::move(w));
            return (int)wide.size() - 1;
        });
        g_wide_stretches = wide.size();
        g_wide_steps = 0;
    }
    for (uint64_t i = 0; i < n; ++i) {
        Tally tally = tally_zero();
        RngPos rng;
        rng.k0 = (uint32_t)seed; rng.k1 = (uint32_t)(seed >> 32);
        rng.pixel = rng3 ? rng3[3 * i] : (uint32_t)i; rng.sample = rng3 ? rng3[3 * i + 1] : 0u; rng.ctr = rng3 ? rng3[3 * i + 2] : 0u;
        RayAoS src; src.o3 = o3; src.d3 = d3; src.i = i;
        Ray w = src.world();
        uint32_t first = 0;  // a "wavefront" of one lane walks the tree without everything its signature rules out
        if (mode == 1 && use_sig) first = tree_without(S, sig_skip_mask(ray_signature(g_sig, w.o, w.d)));
        Best b;
        if (mode == 3) {  // the launches of a tree with BVH entries, one after the other
            b.t = RT_INF; b.ref = RT_NO_HIT; b.chain = 0;
            for (const Stretch& st : stretches) {
                if (st.kind == RT_STRETCH_LOCKSTEP) {
                    b = closest_hit_range<true, true>(S, src, tmin[i], tmax[i], L, rng, tally, st.first, st.end, b);
                } else if (st.kind == RT_STRETCH_WIDE) {
                    const WideBvh& wb = wide[(size_t)st.wide];
                    DWide W;
                    W.nodes = reinterpret_cast<const uint4*>(wb.nodes.data());
                    W.refbox = reinterpret_cast<const double2*>(wb.refbox.data());
                    for (int ax = 0; ax < 3; ++ax) { W.base[ax] = wb.base[ax]; W.step[ax] = wb.step[ax]; }
                    W.K = wb.K; W.lambda = wb.lambda; W.chain = wb.chain; W.fast = (uint32_t)use_fast;
                    for (int ax = 0; ax < 3; ++ax) W.slack[ax] = wb.slack[ax];
                    WLane wl;
                    wide_begin(wl, S, W, w, tmin[i], tmax[i], b);
                    while (wl.state != RT_W_DONE) {
                        g_wide_steps++;
                        if (wl.state == RT_W_NODE) wide_node_step(W, wl);
                        else wide_prim_step<true>(S, W, wl, tally);
                    }
                    b = wl.best;
                } else {
                    DScene R = S;
                    R.n_nodes = st.end;
                    Lane ln;
                    lane_begin(ln, src, tmin[i], tmax[i], L, st.first);
                    ln.best = b;
                    while (ln.state != RT_ST_DONE) {
                        if (ln.state == RT_ST_NODE) lane_node_step<true>(R, src, ln, L, rng, tally);
                        if (ln.state == RT_ST_RUN) lane_prim_step<true>(R, ln, tally);
                        if (ln.state == RT_ST_MEDIUM) lane_medium_step<true>(R, src, ln, L, rng, tally);
                        if (ln.state == RT_ST_CHAIN) lane_chain_step(R, src, ln);
                        if (ln.state == RT_ST_SUB) lane_sub_step<true>(R, ln, tally);
                    }
                    b = ln.best;
                }
            }
        } else if (mode == 2) {  // the per-lane state machine of the persistent kernel
            Lane ln;
            lane_begin(ln, src, tmin[i], tmax[i], L);
            while (ln.state != RT_ST_DONE) {
                if (ln.state == RT_ST_NODE) lane_node_step<true>(S, src, ln, L, rng, tally);
                if (ln.state == RT_ST_RUN) lane_prim_step<true>(S, ln, tally);
                if (ln.state == RT_ST_MEDIUM) lane_medium_step<true>(S, src, ln, L, rng, tally);
                if (ln.state == RT_ST_CHAIN) lane_chain_step(S, src, ln);
                if (ln.state == RT_ST_SUB) lane_sub_step<true>(S, ln, tally);
            }
            b = ln.best;
        } else if (mode == 1) {
            b = closest_hit<true, true>(S, src, tmin[i], tmax[i], L, rng, tally, first);
        } else {
            b = closest_hit<true, false>(S, src, tmin[i], tmax[i], L, rng, tally);
        }
        nt += tally.node_tests; pt += tally.prim_tests; ct += tally.cull_tests;
        bool h = b.ref != RT_NO_HIT;
        Rec rec;
        rec.p = mk(0, 0, 0); rec.n = mk(0, 0, 0); rec.u = 0; rec.v = 0; rec.front = false; rec.prim_id = RT_NO_HIT; rec.mat = 0;
        if (h) rec = finalize<true>(S, w, b);
        hit[i] = h; prim[i] = rec.prim_id; t[i] = h ? b.t : 0.0;
        p3[3 * i] = rec.p.x; p3[3 * i + 1] = rec.p.y; p3[3 * i + 2] = rec.p.z;
        n3[3 * i] = rec.n.x; n3[3 * i + 1] = rec.n.y; n3[3 * i + 2] = rec.n.z;
        uv2[2 * i] = rec.u; uv2[2 * i + 1] = rec.v; front[i] = rec.front; ctr_out[i] = rng.ctr;
    }
    tallies3[0] = nt; tallies3[1] = pt; tallies3[2] = ct;
    return 0;
}

extern "C" int sim_trace_samples(const sim_scene* sc, const rt_camera* cam, uint64_t seed, uint32_t max_depth,
                                 const double* bg, uint64_t n, const uint32_t* pixel, const uint32_t* sample,
                                 double* radiance3, uint32_t* nrays) {
    DScene S = to_dscene(sc);
    double2 slots[RT_MAX_LEVELS];
    Levels L; L.base = slots; L.stride = 1;
    for (uint64_t i = 0; i < n; ++i) {
        Tally tally = tally_zero();
        Rng rng = rng_make(seed, pixel[i], sample[i], 0u);
        V3 o, d;
        camera_ray(*cam, pixel[i] % cam->width, pixel[i] / cam->width, rng, o, d);
        V3 beta = mk(1.0, 1.0, 1.0), lum = mk(0, 0, 0);
        uint32_t rays = 0;
        for (uint32_t depth = 0; depth < max_depth; ++depth) {
            RayVal src; src.o = o; src.d = d;
            Ray w = src.world();
            rays++;
            RngPos rp; rp.k0 = rng.k0; rp.k1 = rng.k1; rp.pixel = rng.pixel; rp.sample = rng.sample; rp.ctr = rng.ctr;
            Best b = closest_hit<false, false>(S, src, 0.001, RT_INF, L, rp, tally);
            rng.ctr = rp.ctr;
            if (b.ref == RT_NO_HIT) { lum = beta * mk(bg[0], bg[1], bg[2]); break; }
            Rec rec = finalize<true>(S, w, b);
            V3 nd, att, emit;
            const rt_material m = S.mats[rec.mat];
            if (!scatter(S, m, w.d, rec, rng, nd, att, emit)) { lum = beta * emit; break; }
            beta = beta * att;
            o = rec.p;
            d = nd;
        }
        radiance3[3 * i] = lum.x; radiance3[3 * i + 1] = lum.y; radiance3[3 * i + 2] = lum.z;
        nrays[i] = rays;
    }
    return 0;
}
