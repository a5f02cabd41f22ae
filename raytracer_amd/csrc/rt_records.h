// rt_records.h -- host side: the 128-byte tree records the kernels read, built from the ABI's rt_node array.
//
//   bytes   0..15   rt_node header; a MEDIUM entry's `count` becomes 1 when its boundary is nothing but spheres,
//                   quads and triangles (the walk then finds both boundary hits in one pass)
//                   a CULL entry's `count` is 1 when it is a signature entry (see signature_nodes) in `pruned`: the
//                   lockstep walk passes it over; a medium with nothing but such an entry becomes one itself
//   bytes  16..63   BVH entry: the reference's box (3 x (start, end)); CULL entry: its box
//   bytes  64..95   CULL entry followed, after its subtree, by another CULL entry of the same level and chain: that
//                   entry's box as six f32 rounded outwards, its `skip`, and 1; zeros otherwise
//   bytes  88..91   BVH entry that is such a leaf and has cull boxes per pair of primitives (build_leaf_boxes): 1 + its
//                   index among those leaves
//   bytes  64..87   BVH entry: conservative cull box as six f32 (x, y, z: start, end), rounded outwards (infinite when
//                   the caller supplied none).  A BVH entry's header carries, in place of the box index the device has
//                   no use for, the run of primitives that is its whole subtree when it is a leaf in the same chain
//                   (the usual case), so that the walk goes from the box test straight to the primitives:
//                   `first` = first primitive, `count` = kind << 24 | count; count 0 otherwise.  The walk reads the first
//                   96 bytes of such a record (six 16-byte loads).
//
// `cull` = false gives the records that switch every cull box off: infinite boxes, and a CULL entry's skip
// pointing at the next entry, because a ray whose interval is already empty fails even an infinite box.
#pragma once

#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "rt_amd.h"

namespace rt {

// Cull boxes inside the leaves of a BVH (no counterpart in the reference; results unchanged).  The reference puts up
// to 16 primitives in a leaf and tests all of them once the leaf's box is hit.  For every BVH leaf that is one run
// of 3..16 triangles or quads in its own chain, eight boxes go to `out` (8 x 6 f32, rounded outwards; empty past the
// end of the run): box j bounds the corners of primitives 2j and 2j + 1 of the run, widened by
// 1e-5 (1 + |coordinate|) like the cull boxes the host layer supplies.  The leaf's record (in `rec`, as built by
// build_records with cull boxes on) gets 1 + the index of its eight boxes in bytes 88..91.
inline void outward_f32(const double* box6, float* out6);
inline void build_leaf_boxes(const rt_node* nodes, size_t n_nodes, const double* tris, const double* quads,
                             std::vector<uint32_t>& rec, std::vector<float>& out) {
    const double inf = std::numeric_limits<double>::infinity();
    for (size_t i = 0; i + 1 < n_nodes; ++i) {
        const rt_node& b = nodes[i];
        const rt_node& n = nodes[i + 1];
        if (b.kind != RT_NODE_BVH || b.skip != i + 2 || n.chain != b.chain) continue;
        if ((n.kind != RT_NODE_TRIANGLES && n.kind != RT_NODE_QUADS) || n.count < 3 || n.count > 16) continue;
        const double* recs = n.kind == RT_NODE_TRIANGLES ? tris : quads;
        const uint32_t leaf = (uint32_t)(out.size() / 48);
        for (uint32_t j = 0; j < 8; ++j) {
            double box[6] = {inf, -inf, inf, -inf, inf, -inf};  // empty: fails every test
            for (uint32_t k = 2 * j; k < std::min(2 * j + 2, n.count); ++k) {
                const double* r = recs + 16 * (size_t)(n.first + k);  // n(3) d q(3) u(3) v(3) w(3)
                for (int c = 0; c < 3; ++c) {
                    const double q = r[4 + c], u = r[7 + c], v = r[10 + c];
                    const double corner[4] = {q, q + u, q + v, n.kind == RT_NODE_QUADS ? q + u + v : q};
                    for (double x : corner) {
                        if (!(x == x)) { box[2 * c] = -inf; box[2 * c + 1] = inf; continue; }  // NaN: no culling
                        box[2 * c] = std::min(box[2 * c], x - 1e-5 * (1.0 + std::fabs(x)));
                        box[2 * c + 1] = std::max(box[2 * c + 1], x + 1e-5 * (1.0 + std::fabs(x)));
                    }
                }
            }
            float f[6];
            outward_f32(box, f);
            out.insert(out.end(), f, f + 6);
        }
        rec[32 * i + 22] = leaf + 1u;
    }
}

// A box (x, y, z: start, end) as f32, every bound rounded away from the inside.
inline void outward_f32(const double* box6, float* out6) {
    for (int k = 0; k < 6; ++k) {
        float f = (float)box6[k];
        if (!(k & 1) && (double)f > box6[k]) f = std::nextafterf(f, -std::numeric_limits<float>::infinity());
        if ((k & 1) && (double)f < box6[k]) f = std::nextafterf(f, std::numeric_limits<float>::infinity());
        out6[k] = f;
    }
}

inline std::vector<uint32_t> build_records(const rt_node* nodes, size_t n_nodes, const double* boxes,
                                           const double* cull_boxes, bool cull, bool one_pass_media,
                                           const std::vector<uint32_t>& sig_nodes = {}, uint32_t pruned = 0u,
                                           uint32_t index_base = 0u) {
    const double inf = std::numeric_limits<double>::infinity();
    const double everything[6] = {-inf, inf, -inf, inf, -inf, inf};
    std::vector<uint32_t> rec(32 * n_nodes, 0u);
    for (size_t i = 0; i < n_nodes; ++i) {
        rt_node n = nodes[i];
        uint32_t* r = &rec[32 * i];
        bool boundary_pruned = false;  // a medium whose whole boundary lies under a pruned entry goes with it
        if (n.kind == RT_NODE_MEDIUM && cull && i + 1 < n_nodes && nodes[i + 1].kind == RT_NODE_CULL && nodes[i + 1].skip == n.skip)
            for (size_t b = 0; b < sig_nodes.size(); ++b)
                if (sig_nodes[b] == i + 1 && ((pruned >> b) & 1u)) boundary_pruned = true;
        if (boundary_pruned) {
            n.kind = RT_NODE_CULL;
            n.count = 1u;
        } else if (n.kind == RT_NODE_MEDIUM) {
            bool flat = one_pass_media;
            for (uint32_t j = (uint32_t)i + 1u; j < n.skip; ++j) {
                const uint8_t k = nodes[j].kind;
                flat = flat && (k == RT_NODE_SPHERES || k == RT_NODE_QUADS || k == RT_NODE_TRIANGLES || k == RT_NODE_CULL);
            }
            n.count = flat ? 1u : 0u;
        } else if (n.kind == RT_NODE_BVH) {
            std::memcpy(r + 4, boxes + 6 * (size_t)n.first, 48);
            float cf[6];
            outward_f32((cull && cull_boxes) ? cull_boxes + 6 * (size_t)n.first : everything, cf);
            std::memcpy(r + 16, cf, 24);
            n.first = 0u;
            n.count = 0u;
            if (n.skip == i + 2) {
                const rt_node& c = nodes[i + 1];
                if (c.kind >= RT_NODE_SPHERES && c.kind <= RT_NODE_QUADRICS && c.count > 0 && c.count < (1u << 24) &&
                    c.chain == n.chain) {
                    n.first = c.first;
                    n.count = (uint32_t)c.kind << 24 | c.count;
                }
            }
        } else if (n.kind == RT_NODE_CULL) {
            std::memcpy(r + 4, cull ? boxes + 6 * (size_t)n.first : everything, 48);
            // the CULL entry that follows this one's subtree at the same level (the next box of a row of boxes), so
            // that a lane failing this box tests the next in the same step
            if (cull && n.skip < n_nodes) {
                const rt_node& nx = nodes[n.skip];
                if (nx.kind == RT_NODE_CULL && nx.chain == n.chain && nx.level == n.level) {
                    float cf[6];
                    outward_f32(boxes + 6 * (size_t)nx.first, cf);
                    std::memcpy(r + 16, cf, 24);
                    r[22] = nx.skip + index_base;
                    r[23] = 1u;
                }
            }
            n.count = 0u;
            for (size_t b = 0; cull && b < sig_nodes.size(); ++b)
                if (sig_nodes[b] == i && ((pruned >> b) & 1u)) n.count = 1u;  // the lockstep walk passes it over
            if (!cull) n.skip = (uint32_t)i + 1u;  // (index_base is added below)
        }
        n.skip += index_base;  // every index in the record counts from where this copy of the tree starts
        std::memcpy(r, &n, sizeof(rt_node));
    }
    return rec;
}

// The CULL entries a ray's signature covers, for a tree without BVH entries (the lockstep walk): the first `max_sig`
// in tree order of those the walk itself meets (not inside another one, not inside a medium's boundary) and those
// that hold a medium's whole boundary.  Empty for any other tree.
inline std::vector<uint32_t> signature_nodes(const rt_node* nodes, size_t n_nodes, size_t max_sig) {
    std::vector<uint32_t> out;
    for (size_t i = 0; i < n_nodes; ++i)
        if (nodes[i].kind == RT_NODE_BVH) return out;
    for (size_t i = 0; i < n_nodes && out.size() < max_sig;) {
        if (nodes[i].kind == RT_NODE_CULL) {
            out.push_back((uint32_t)i);
            i = nodes[i].skip;
        } else if (nodes[i].kind == RT_NODE_MEDIUM) {
            if (i + 1 < n_nodes && nodes[i + 1].kind == RT_NODE_CULL && nodes[i + 1].skip == nodes[i].skip)
                out.push_back((uint32_t)i + 1u);
            i = nodes[i].skip;
        } else {
            ++i;
        }
    }
    return out;
}

// Every copy of the tree the lockstep walk chooses from, one after the other: copy m lacks the signature entries in m.
inline std::vector<uint32_t> build_record_variants(const rt_node* nodes, size_t n_nodes, const double* boxes,
                                                   const double* cull_boxes, bool one_pass_media,
                                                   const std::vector<uint32_t>& sig_nodes) {
    std::vector<uint32_t> all;
    for (uint32_t m = 0; m < (1u << sig_nodes.size()); ++m) {
        const std::vector<uint32_t> v = build_records(nodes, n_nodes, boxes, cull_boxes, true, one_pass_media, sig_nodes, m,
                                                      m * (uint32_t)n_nodes);
        all.insert(all.end(), v.begin(), v.end());
    }
    return all;
}

// Per quad (16 f64: n d q u v w): 1 + the coordinate axis its normal lies along when n and w are exactly zero in the
// other two components and every value is of a magnitude that cannot overflow in the intersection arithmetic; else 0.
inline std::vector<uint32_t> quad_axis_flags(const double* quads, size_t n_quads) {
    std::vector<uint32_t> out(n_quads, 0u);
    for (size_t k = 0; k < n_quads; ++k) {
        const double* q = quads + 16 * k;
        bool tame = true;
        for (int j = 0; j < 16; ++j) tame = tame && std::fabs(q[j]) < 1e100;  // false for NaN too
        if (!tame) continue;
        const double* nrm = q;
        const double* w = q + 13;
        for (int a = 0; a < 3; ++a) {
            const int b = (a + 1) % 3, c = (a + 2) % 3;
            if (nrm[a] != 0.0 && nrm[b] == 0.0 && nrm[c] == 0.0 && w[b] == 0.0 && w[c] == 0.0) out[k] = 1u + (uint32_t)a;
        }
    }
    return out;
}

}  // namespace rt
