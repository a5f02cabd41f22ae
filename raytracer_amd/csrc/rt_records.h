// rt_records.h -- host side: the 128-byte tree records the kernels read, built from the ABI's rt_node array.
//
//   bytes   0..15   rt_node header; a MEDIUM entry's `count` becomes 1 when its boundary is nothing but spheres,
//                   quads and triangles (the walk then finds both boundary hits in one pass)
//   bytes  16..63   BVH entry: the reference's box (3 x (start, end)); CULL entry: its box
//   bytes  64..111  BVH entry: conservative cull box (infinite when the caller supplied none)
//   bytes 112..127  BVH entry whose subtree is exactly one run of primitives in the same chain (a leaf, the usual
//                   case): kind, first, count of that run, so that the walk goes from the box test straight to the
//                   primitives; kind 0 otherwise
//
// `cull` = false gives the records that switch every cull box off: infinite boxes, and a CULL entry's skip
// pointing at the next entry, because a ray whose interval is already empty fails even an infinite box.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "rt_amd.h"

namespace rt {

inline std::vector<uint32_t> build_records(const rt_node* nodes, size_t n_nodes, const double* boxes,
                                           const double* cull_boxes, bool cull, bool one_pass_media) {
    const double inf = std::numeric_limits<double>::infinity();
    const double everything[6] = {-inf, inf, -inf, inf, -inf, inf};
    std::vector<uint32_t> rec(32 * n_nodes, 0u);
    for (size_t i = 0; i < n_nodes; ++i) {
        rt_node n = nodes[i];
        uint32_t* r = &rec[32 * i];
        if (n.kind == RT_NODE_MEDIUM) {
            bool flat = one_pass_media;
            for (uint32_t j = (uint32_t)i + 1u; j < n.skip; ++j) {
                const uint8_t k = nodes[j].kind;
                flat = flat && (k == RT_NODE_SPHERES || k == RT_NODE_QUADS || k == RT_NODE_TRIANGLES || k == RT_NODE_CULL);
            }
            n.count = flat ? 1u : 0u;
        } else if (n.kind == RT_NODE_BVH) {
            std::memcpy(r + 4, boxes + 6 * (size_t)n.first, 48);
            std::memcpy(r + 16, (cull && cull_boxes) ? cull_boxes + 6 * (size_t)n.first : everything, 48);
            if (n.skip == i + 2) {
                const rt_node& c = nodes[i + 1];
                if (c.kind >= RT_NODE_SPHERES && c.kind <= RT_NODE_QUADRICS && c.count > 0 && c.chain == n.chain) {
                    r[28] = c.kind; r[29] = c.first; r[30] = c.count;
                }
            }
        } else if (n.kind == RT_NODE_CULL) {
            std::memcpy(r + 4, cull ? boxes + 6 * (size_t)n.first : everything, 48);
            if (!cull) n.skip = (uint32_t)i + 1u;
        }
        std::memcpy(r, &n, sizeof(rt_node));
    }
    return rec;
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
