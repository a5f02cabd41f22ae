// rt_wide.h -- an eight-wide acceleration tree over the LEAVES of a reference BVH, and the per-lane walk that uses it.
//
// Why the reference's inner nodes can be replaced.  BVHNode::hit (bvh.rs:29-60) narrows the ray interval by its box
// and hands the result to its children.  Every box is the union of the boxes of the objects below it (bvh.rs:88-92),
// so a child's box lies inside its parent's, and so does the slab interval a ray cuts out of it, bound by bound
// ((lo - o) * inv is monotone in lo).  Narrowing by parent and child therefore gives what narrowing by the child
// alone gives, and a parent that misses implies a child that misses: what a leaf's primitives are tested against is
// (tmin, min(tmax, closest so far)) narrowed by the LEAF's box alone, and a leaf is entered exactly when that test
// passes.  The inner nodes decide nothing but which leaves are looked at -- in the reference's order, upper half
// first (bvh.rs:50-57,127-130).  build_wide checks the nesting box by box (and that no box is degenerate) and refuses
// the tree otherwise; such a tree keeps the entry-by-entry walk of rt_core.h.
//
// The tree built here: the leaves in visiting order, each cut into items (a triangle, a pair of quads or of spheres, each
// with the box around it; one box-shaped list of quads, ...) that take consecutive item slots;
// a leaf of several items starts on a multiple of eight slots, and so does a change of kind inside it.  Above the
// items sits a complete 8-ary tree in heap order: node (k, g) covers slots [g * 8^k, (g + 1) * 8^k) and holds the
// boxes of its eight children on a 16-bit grid, rounded outwards (none of them narrows anything: a subtree is passed
// over only when the ray's interval cannot touch the geometry below it).  A lane walks it with a cursor that only
// moves forward, as the reference does, and remembers which children of the nodes above it are still to be visited
// in eight bits per level of one 64-bit register.  The reference's narrowing by the leaf's own f64 box happens in the
// first primitive step below each level-1 node of the leaf (lambda = 1; below every one of them anew: the narrowed
// interval is a function of the box and of the closest hit so far, so doing it again gives what carrying it along
// would) or, where every leaf is a single item and eight leaves share a level-1 node (lambda = 0), in each item's.
#pragma once

#include "rt_core.h"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <vector>

namespace rt {

#define RT_WIDE_LOG 3u                    // eight children per node
#define RT_WIDE (1u << RT_WIDE_LOG)
#define RT_WIDE_MAX_K 7u                  // eight pending bits per level (levels 1..7) in one 64-bit register

struct DWide {               // as the kernels see it
    const uint4* nodes;      // 128 bytes per node: the eight child boxes, 6 x u16 each (x, y, z: start, end) on the grid
                             // base + q * step of the tree's bounds, starts rounded down and ends up (96 bytes); then
                             // four words: level-1 nodes: the first primitive of the node's first item | bits 0..7 the
                             // items' primitive kind, bits 8..10 the axis along which children 0..3 lie beyond (bit 10:
                             // before) children 4..7, bits 16..23 which items are the six sides of an axis-aligned box in
                             // create_box's order | the eight items' primitive counts, a byte each (two words).  The
                             // items of a level-1 node are consecutive runs of one array of primitives.
    const double2* refbox;   // per level-lambda node: the reference's box of its leaf, 3 x (start, end)
    double base[3], step[3]; // the grid
    uint32_t K;              // the root is node (K, 0)
    uint32_t lambda;         // 1: the level-1 nodes belong to one leaf each and narrow by its box; 0: the items do
    uint32_t chain;          // the instance chain the whole tree lives in
    uint32_t fast;           // 1: children are visited nearest first (see wide_node_step); results then differ from the
                             // reference's where two primitives are hit at exactly the same parameter
    float slack[3];          // per axis, in grid steps: how far an item's box on the grid can reach beyond the geometry
                             // (the widening of prim_box, the rounding onto the grid, the extra step of either side and
                             // the f32 evaluation of wide_box); read where a box's sides are chosen (wide_box_pairs)
};
RT_HD uint32_t wide_node_index(uint32_t K, uint32_t k, uint32_t g) {
    return (0x49249249u & ((1u << (RT_WIDE_LOG * (K - k))) - 1u)) + g;  // (8^(K-k) - 1) / 7 nodes lie above level k
}

// ---------------------------------------------------------------------------------------------------- host side
struct WideBvh {
    uint32_t entry = 0, end = 0, chain = 0, K = 0, lambda = 0, leaves = 0;
    std::vector<uint32_t> nodes;  // 32 words per node
    std::vector<double> refbox;   // 6 per level-lambda node
    double base[3] = {0, 0, 0}, step[3] = {1, 1, 1};
    double bounds[6] = {0, 0, 0, 0, 0, 0};  // of everything below the entry (x, y, z: start, end), in its chain
    uint32_t only = 0;            // the kind of every item's primitives when they are all of one kind, else 0
    float slack[3] = {0, 0, 0};   // see DWide
};
namespace wide_detail {
struct Item { uint32_t kind, first, count; double box[6]; };
struct Leaf { uint32_t box; std::vector<Item> items; };
struct Ctx {
    const rt_node* nodes; size_t n; const double* boxes; const double* spheres; const double* quads; const double* tris;
    uint32_t chain; std::vector<Leaf> leaves;
};
inline void grow(double* box, double lo, double hi, int c) {
    const double inf = std::numeric_limits<double>::infinity();
    if (!(lo == lo) || !(hi == hi)) { box[2 * c] = -inf; box[2 * c + 1] = inf; return; }  // NaN: no culling
    const double a = std::isfinite(lo) ? lo - 1e-5 * (1.0 + std::fabs(lo)) : lo;
    const double b = std::isfinite(hi) ? hi + 1e-5 * (1.0 + std::fabs(hi)) : hi;
    box[2 * c] = std::min(box[2 * c], a);
    box[2 * c + 1] = std::max(box[2 * c + 1], b);
}
// Bounds of every point primitives [first, first + count) of one kind can report a hit at, widened by
// 1e-5 (1 + |coordinate|) like the cull boxes of the host layer.
inline void prim_box(const Ctx& c, uint32_t kind, uint32_t first, uint32_t count, double* box) {
    const double inf = std::numeric_limits<double>::infinity();
    for (int a = 0; a < 3; ++a) { box[2 * a] = inf; box[2 * a + 1] = -inf; }
    for (uint32_t k = first; k < first + count; ++k) {
        for (int a = 0; a < 3; ++a) {
            if (kind == RT_NODE_SPHERES) {
                const double* s = c.spheres + 4 * (size_t)k;
                grow(box, s[a] - std::fabs(s[3]), s[a] + std::fabs(s[3]), a);
            } else if (kind == RT_NODE_QUADS || kind == RT_NODE_TRIANGLES) {
                const double* r = (kind == RT_NODE_QUADS ? c.quads : c.tris) + 16 * (size_t)k;  // n d q u v w
                const double q = r[4 + a], u = r[7 + a], v = r[10 + a];
                const double corner[4] = {q, q + u, q + v, kind == RT_NODE_QUADS ? q + u + v : q};
                for (double x : corner) grow(box, x, x, a);
            } else {
                grow(box, -inf, inf, a);  // a quadric is unbounded
            }
        }
    }
}
inline bool is_run(uint8_t kind) { return kind >= RT_NODE_SPHERES && kind <= RT_NODE_QUADRICS; }
// Whether quads [first, first + 6) are the six sides of an axis-aligned box in create_box's order
// (parallelogram.rs:146-195): z = max, x = max, z = min, x = min, y = max, y = min, each side a rectangle with edges
// along the axes that spans the whole face.  (The walk may then look at the two or four sides a ray can meet first and
// at the others only if those are missed: wide_box_pairs.)
inline bool is_reference_box(const double* quads, uint32_t first) {
    const double inf = std::numeric_limits<double>::infinity();
    double lo[3] = {inf, inf, inf}, hi[3] = {-inf, -inf, -inf};
    for (uint32_t i = 0; i < 6u; ++i) {
        const double* r = quads + 16 * (size_t)(first + i);
        for (int j = 0; j < 16; ++j) if (!std::isfinite(r[j])) return false;
        for (int a = 0; a < 3; ++a) {
            const double q = r[4 + a], u = r[7 + a], v = r[10 + a];
            for (double x : {q, q + u, q + v, q + u + v}) { lo[a] = std::min(lo[a], x); hi[a] = std::max(hi[a], x); }
        }
    }
    static const int axis[6] = {2, 0, 2, 0, 1, 1};
    static const bool upper[6] = {true, true, false, false, true, false};
    for (uint32_t i = 0; i < 6u; ++i) {
        const double* r = quads + 16 * (size_t)(first + i);
        const double *n = r, *q = r + 4, *u = r + 7, *v = r + 10;
        const int a = axis[i], b = (a + 1) % 3, c = (a + 2) % 3;
        if (!(lo[a] < hi[a])) return false;
        if (!(n[a] != 0.0 && n[b] == 0.0 && n[c] == 0.0 && u[a] == 0.0 && v[a] == 0.0)) return false;
        if (q[a] != (upper[i] ? hi[a] : lo[a])) return false;
        // u along one of the two other axes, v along the other, and together they reach from corner to corner
        const bool ub = u[b] != 0.0 && u[c] == 0.0 && v[c] != 0.0 && v[b] == 0.0;
        const bool uc = u[c] != 0.0 && u[b] == 0.0 && v[b] != 0.0 && v[c] == 0.0;
        if (!ub && !uc) return false;
        for (int x : {b, c}) {
            const double e0 = q[x], e1 = q[x] + u[x] + v[x];
            if (std::min(e0, e1) != lo[x] || std::max(e0, e1) != hi[x]) return false;
        }
    }
    return true;
}
inline bool collect(Ctx& c, uint32_t e, int parent, int depth) {
    if (depth > 40) return false;
    const rt_node& n = c.nodes[e];
    if (n.kind != RT_NODE_BVH || n.chain != c.chain || n.skip <= e || n.skip > c.n) return false;
    const double* b = c.boxes + 6 * (size_t)n.first;
    for (int a = 0; a < 3; ++a) {
        if (!(b[2 * a] < b[2 * a + 1])) return false;  // degenerate (or NaN): the nesting argument needs lo < hi
        if (parent >= 0) {
            const double* pb = c.boxes + 6 * (size_t)c.nodes[parent].first;
            if (!(b[2 * a] >= pb[2 * a] && b[2 * a + 1] <= pb[2 * a + 1])) return false;
        }
    }
    if (e + 1 < n.skip && c.nodes[e + 1].kind == RT_NODE_BVH) {
        for (uint32_t ch = e + 1; ch < n.skip; ch = c.nodes[ch].skip) {
            if (c.nodes[ch].kind != RT_NODE_BVH || c.nodes[ch].level != n.level + 1u) return false;
            if (!collect(c, ch, (int)e, depth + 1)) return false;
        }
        return true;
    }
    Leaf leaf;
    leaf.box = n.first;
    for (uint32_t j = e + 1; j < n.skip;) {
        const rt_node& x = c.nodes[j];
        if (x.chain != c.chain || x.level != n.level + 1u) return false;
        const rt_node* run = nullptr;
        bool bare = false;
        if (is_run(x.kind)) { run = &x; bare = true; j = j + 1; }
        else if (x.kind == RT_NODE_CULL && x.skip == j + 2 && x.skip <= n.skip && is_run(c.nodes[j + 1].kind) &&
                 c.nodes[j + 1].chain == c.chain && c.nodes[j + 1].level == n.level + 1u) { run = &c.nodes[j + 1]; j = x.skip; }
        else return false;
        if (run->count == 0 || run->count >= (1u << 24)) { if (run->count == 0) continue; return false; }
        const bool planar = run->kind == RT_NODE_QUADS || run->kind == RT_NODE_TRIANGLES;
        if (bare && (planar || run->kind == RT_NODE_SPHERES) && run->count >= 3) {  // each with the box around its corners (around its spheres)
            // one triangle to an item (a mesh's triangles are small against their leaf: a box each tells most of them
            // apart, and a test passed over saves the eight loads of the record -- cornell_model - 5 %, bust - 20 % against
            // pairs); quads and spheres in pairs (singly: no gain on final_scene and many_spheres, spherical_mirror + 2 %)
            const uint32_t per = run->kind == RT_NODE_TRIANGLES ? 1u : 2u;
            for (uint32_t k = 0; k < run->count; k += per) {
                Item it; it.kind = run->kind; it.first = run->first + k; it.count = std::min(per, run->count - k);
                prim_box(c, it.kind, it.first, it.count, it.box);
                leaf.items.push_back(it);
            }
        } else {
            Item it; it.kind = run->kind; it.first = run->first; it.count = run->count;
            prim_box(c, it.kind, it.first, it.count, it.box);
            leaf.items.push_back(it);
        }
    }
    c.leaves.push_back(std::move(leaf));
    return true;
}
}  // namespace wide_detail

namespace wide_detail {
// The tree over the collected leaves for one choice of lambda; false when a level-1 node would hold items of different
// kinds or of runs that are not consecutive (possible for lambda = 0 only), or when the tree would be too deep.
inline bool build_wide_slots(const Ctx& c, const rt_node* nodes, uint32_t root, const double* boxes, uint32_t lambda, WideBvh& out) {
    std::vector<std::vector<uint32_t>> slot_of(c.leaves.size());
    std::vector<uint32_t> box_of;  // per level-lambda node: the leaf whose box narrows the interval there
    uint64_t slots = 0;
    for (size_t l = 0; l < c.leaves.size(); ++l) {
        const Leaf& leaf = c.leaves[l];
        slot_of[l].resize(leaf.items.size());
        if (lambda == 0u) {
            if (leaf.items.size() != 1) return false;
            slot_of[l][0] = (uint32_t)slots++;
            box_of.push_back((uint32_t)l);
            continue;
        }
        for (size_t i = 0; i < leaf.items.size(); ++i) {
            const Item& it = leaf.items[i];
            if (i > 0) {
                const Item& before = leaf.items[i - 1];
                if (it.kind != before.kind || it.first != before.first + before.count) slots = (slots + RT_WIDE - 1) / RT_WIDE * RT_WIDE;
            }
            slot_of[l][i] = (uint32_t)slots++;
        }
        slots = (slots + RT_WIDE - 1) / RT_WIDE * RT_WIDE;
        while (box_of.size() < slots / RT_WIDE) box_of.push_back((uint32_t)l);
    }
    uint32_t K = 1;
    while ((1ull << (RT_WIDE_LOG * K)) < slots) ++K;
    if (K > RT_WIDE_MAX_K) return false;  // 2 M item slots
    out.entry = root; out.end = nodes[root].skip; out.chain = c.chain; out.K = K; out.lambda = lambda;
    out.leaves = (uint32_t)c.leaves.size();
    out.only = 0u;
    {
        bool first = true, same = true;
        for (const Leaf& l : c.leaves)
            for (const Item& it : l.items) { if (first) { out.only = it.kind; first = false; } same = same && it.kind == out.only; }
        if (!same) out.only = 0u;
    }
    const double inf = std::numeric_limits<double>::infinity();
    const size_t width = (size_t)1 << (RT_WIDE_LOG * K);
    // every level's boxes first (the grid needs the bounds of the whole tree), then the records
    std::vector<std::vector<double>> lv(K + 1);
    lv[0].assign(6 * width, 0.0);
    for (size_t i = 0; i < width; ++i) for (int a = 0; a < 3; ++a) { lv[0][6 * i + 2 * a] = inf; lv[0][6 * i + 2 * a + 1] = -inf; }
    std::vector<uint32_t> desc(2 * width, 0u);
    out.refbox.resize(6 * box_of.size());
    for (size_t g = 0; g < box_of.size(); ++g) std::memcpy(&out.refbox[6 * g], boxes + 6 * (size_t)c.leaves[box_of[g]].box, 48);
    for (size_t l = 0; l < c.leaves.size(); ++l) {
        for (size_t i = 0; i < c.leaves[l].items.size(); ++i) {
            const Item& it = c.leaves[l].items[i];
            const size_t slot = slot_of[l][i];
            std::memcpy(&lv[0][6 * slot], it.box, 48);
            desc[2 * slot] = it.first;
            desc[2 * slot + 1] = it.kind << 24 | it.count;
        }
    }
    for (uint32_t k = 1; k <= K; ++k) {
        const size_t count = lv[k - 1].size() / (6 * RT_WIDE);
        lv[k].assign(6 * count, 0.0);
        for (size_t g = 0; g < count; ++g) {
            double u[6] = {inf, -inf, inf, -inf, inf, -inf};
            for (uint32_t j = 0; j < RT_WIDE; ++j) {
                const double* b = &lv[k - 1][6 * (RT_WIDE * g + j)];
                if (b[0] <= b[1]) for (int a = 0; a < 3; ++a) { u[2 * a] = std::min(u[2 * a], b[2 * a]); u[2 * a + 1] = std::max(u[2 * a + 1], b[2 * a + 1]); }
            }
            std::memcpy(&lv[k][6 * g], u, 48);
        }
    }
    std::memcpy(out.bounds, lv[K].data(), 48);
    for (int a = 0; a < 3; ++a) {
        const double lo = out.bounds[2 * a], hi = out.bounds[2 * a + 1];
        if (!(lo <= hi)) { out.base[a] = 0.0; out.step[a] = 1.0; continue; }  // nothing below the entry
        if (!std::isfinite(lo) || !std::isfinite(hi)) return false;           // unbounded geometry: no grid
        out.step[a] = std::max((hi - lo) / 65530.0, 1e-300);
        out.base[a] = lo - 2.0 * out.step[a];
        if (!std::isfinite(out.step[a]) || !std::isfinite(out.base[a])) return false;
    }
    for (int a = 0; a < 3; ++a) {
        // prim_box widens by 1e-5 (1 + |coordinate|); onto the grid: less than one step, one more step either side, and
        // wide_box's f32 arithmetic is good for that extra step (see wide_begin); half a step on top
        const double reach = std::max(std::fabs(out.bounds[2 * a]), std::fabs(out.bounds[2 * a + 1]));
        const double widen = std::isfinite(reach) ? 1e-5 * (1.0 + reach) / out.step[a] : 0.0;
        out.slack[a] = (float)std::min(widen + 3.5, 1e30);
    }
    const size_t total = (width - 1) / (RT_WIDE - 1);
    out.nodes.assign(32 * total, 0u);
    for (uint32_t k = 1; k <= K; ++k) {
        const size_t count = lv[k].size() / 6;
        for (size_t g = 0; g < count; ++g) {
            uint32_t* words = &out.nodes[32 * (size_t)wide_node_index(K, k, (uint32_t)g)];
            uint16_t* rec = reinterpret_cast<uint16_t*>(words);
            for (uint32_t j = 0; j < RT_WIDE; ++j) {
                const double* b = &lv[k - 1][6 * (RT_WIDE * g + j)];
                for (int a = 0; a < 3; ++a) {
                    uint16_t q0 = 65535, q1 = 65535;  // no child: a box without extent, which no interval can touch
                    if (b[0] <= b[1]) {
                        // largest grid point <= start, smallest >= end, checked against the values the walk computes
                        double f0 = std::floor((b[2 * a] - out.base[a]) / out.step[a]), f1 = std::ceil((b[2 * a + 1] - out.base[a]) / out.step[a]);
                        f0 = std::min(std::max(f0, 0.0), 65535.0); f1 = std::min(std::max(f1, 0.0), 65535.0);
                        while (f0 > 0.0 && out.base[a] + f0 * out.step[a] > b[2 * a]) f0 -= 1.0;
                        while (f1 < 65535.0 && out.base[a] + f1 * out.step[a] < b[2 * a + 1]) f1 += 1.0;
                        if (out.base[a] + f0 * out.step[a] > b[2 * a] || out.base[a] + f1 * out.step[a] < b[2 * a + 1]) return false;
                        // one more grid step either way: the walk evaluates these bounds in f32 (wide_box)
                        q0 = (uint16_t)std::max(f0 - 1.0, 0.0); q1 = (uint16_t)std::min(f1 + 1.0, 65535.0);
                    }
                    rec[6 * j + 2 * a] = q0; rec[6 * j + 2 * a + 1] = q1;
                }
            }
            uint32_t axis = 0;
            {   // The reference splits a node along its longest axis and visits the upper half first (bvh.rs:118-130),
                // so the first half of a node's children usually lies beyond the second along one axis: the one along
                // which the centres of the two halves are farthest apart.  Only the nearest-first order of the fast
                // mode reads it.
                double ctr[2][3] = {{0, 0, 0}, {0, 0, 0}};
                int have[2] = {0, 0};
                for (uint32_t j = 0; j < RT_WIDE; ++j) {
                    const double* b = &lv[k - 1][6 * (RT_WIDE * g + j)];
                    if (!(b[0] <= b[1])) continue;
                    const int h = j < RT_WIDE / 2 ? 0 : 1;
                    for (int a = 0; a < 3; ++a) ctr[h][a] += 0.5 * (b[2 * a] + b[2 * a + 1]);
                    have[h]++;
                }
                if (have[0] && have[1]) {
                    double best = -1.0;
                    for (int a = 0; a < 3; ++a) {
                        const double sep = std::fabs(ctr[0][a] / have[0] - ctr[1][a] / have[1]);
                        if (sep > best) { best = sep; axis = (uint32_t)a; }
                    }
                    if (ctr[0][axis] / have[0] < ctr[1][axis] / have[1]) axis |= 4u;  // the first half is the LOWER one here
                }
            }
            uint32_t kind = 0, first = 0, boxes = 0;
            unsigned long long counts = 0ull;
            if (k == 1) {  // the items: consecutive runs of one array of primitives, the used slots first
                bool ended = false;
                uint32_t next = 0;
                for (uint32_t j = 0; j < RT_WIDE; ++j) {
                    const uint32_t f = desc[2 * (RT_WIDE * g + j)], kc = desc[2 * (RT_WIDE * g + j) + 1];
                    const uint32_t cnt = kc & 0xFFFFFFu;
                    if (cnt == 0) { ended = true; continue; }
                    if (ended || cnt > 255u) return false;
                    if (j == 0) { kind = kc >> 24; first = f; next = f; }
                    if ((kc >> 24) != kind || f != next) return false;
                    next = f + cnt;
                    counts |= (unsigned long long)cnt << (8u * j);
                    if (kind == RT_NODE_QUADS && cnt == 6u && is_reference_box(c.quads, f)) boxes |= 1u << j;
                }
            }
            words[24] = first;
            words[25] = kind | axis << 8 | boxes << 16;
            words[26] = (uint32_t)counts;
            words[27] = (uint32_t)(counts >> 32);
        }
    }
    return true;
}
}  // namespace wide_detail

// Builds the wide tree for the top-level BVH entry `root` (level 0).  False when the subtree is not of the shape
// described above; the caller then walks it entry by entry.
inline bool build_wide(const rt_node* nodes, size_t n_nodes, uint32_t root, const double* boxes, const double* spheres,
                       const double* quads, const double* tris, WideBvh& out) {
    using namespace wide_detail;
    if (root >= n_nodes || nodes[root].kind != RT_NODE_BVH || nodes[root].level != 0) return false;
    Ctx c; c.nodes = nodes; c.n = n_nodes; c.boxes = boxes; c.spheres = spheres; c.quads = quads; c.tris = tris;
    c.chain = nodes[root].chain;
    if (!collect(c, root, -1, 0) || c.leaves.empty()) return false;
    size_t most = 0;
    for (const Leaf& l : c.leaves) most = std::max(most, l.items.size());
    // lambda = 0: every leaf is one item, eight leaves to a level-1 node.  lambda = 1: a leaf is one or more level-1 nodes
    // of its own (each of them does the leaf's narrowing: the interval depends on the box and the closest hit so far
    // alone); a new one starts wherever the items change kind or stop being consecutive runs of their array.
    if (most <= 1 && build_wide_slots(c, nodes, root, boxes, 0u, out)) return true;
    return build_wide_slots(c, nodes, root, boxes, 1u, out);
}


// The stretches of top-level entries a tree with BVH entries is walked in, one launch each, the closest hit so far
// travelling from one to the next: entries with no BVH entry below them (every ray does the same work there) together
// by the lockstep walk; a BVH entry build_wide accepts (wide[k] >= 0 for top-level entry k... given per entry index)
// by the wide walk, one stretch each; the rest by the entry-by-entry per-lane walk.
#define RT_STRETCH_LOCKSTEP 0
#define RT_STRETCH_LANES 1
#define RT_STRETCH_WIDE 2
struct Stretch { uint32_t first, end; int kind; int wide; uint32_t levels; };  // levels: interval slots per lane its walk needs
template <class Qualifies>  // int qualifies(uint32_t entry): index of the entry's wide tree, or -1
inline std::vector<Stretch> build_stretches(const rt_node* nodes, uint32_t nn, bool has_bvh, Qualifies qualifies) {
    std::vector<Stretch> out;
    for (uint32_t i = 0; i < nn;) {
        const rt_node& n = nodes[i];
        const uint32_t next = (n.kind == RT_NODE_BVH || n.kind == RT_NODE_MEDIUM || n.kind == RT_NODE_CULL) ? n.skip : i + 1u;
        bool uniform = true;
        for (uint32_t j = i; j < next; ++j) uniform = uniform && nodes[j].kind != RT_NODE_BVH;
        if (!has_bvh) uniform = true;
        int kind = uniform ? RT_STRETCH_LOCKSTEP : RT_STRETCH_LANES, w = -1;
        if (!uniform && n.kind == RT_NODE_BVH) { w = qualifies(i); if (w >= 0) kind = RT_STRETCH_WIDE; }
        if (!out.empty() && out.back().kind == kind && kind != RT_STRETCH_WIDE) out.back().end = next;
        else out.push_back({i, next, kind, w, 1u});
        i = next;
    }
    if (out.empty()) out.push_back({0u, 0u, RT_STRETCH_LOCKSTEP, -1, 1u});
    for (Stretch& st : out)  // an entry reads the interval of its level; a BVH or MEDIUM entry writes the next one
        for (uint32_t j = st.first; j < st.end; ++j)
            st.levels = std::max<uint32_t>(st.levels, nodes[j].level + ((nodes[j].kind == RT_NODE_BVH || nodes[j].kind == RT_NODE_MEDIUM) ? 2u : 1u));
    return out;
}

// ---------------------------------------------------------------------------------------------------- the walk
#define RT_W_NODE 0u
#define RT_W_PRIM 1u
#define RT_W_DONE 2u
struct WLane {
    Ray cur;                  // in the tree's chain, reciprocal direction ready
    float ax, ay, az, bx, by, bz;  // grid value q of a child box -> ray parameter q * a + b, per axis (f32, see wide_box)
    uint32_t cur_chain;
    double tmin, tmax;        // the ray's interval at the top level
    double s, e;              // inside a leaf: the interval its box narrowed that to, e shortened by every hit since
    Best best;
    unsigned long long pending;  // bits 8k .. 8k + 7: children of the level-k node above the cursor still to be visited
    uint32_t rev;                // fast mode: bit k set = the level-k node's children are taken from 7 down to 0;
                                 // bit 31: the leaf's box has yet to narrow the interval below the level-1 node above (lambda = 1)
    uint32_t slot, level;     // the node to look at next is (level, slot >> 3 level)
    uint32_t k, kend, kind;   // run in progress; kind bit 9: the leaf's box has yet to narrow the interval (lambda = 0);
                              // bit 10: the item is a box whose sides have yet to be chosen (wide_box_pairs); bit 11: the
                              // chosen sides are being tested, bits 12..14 which pairs those are; bit 8 and bits 16..:
                              // lane_skip_pairs' mask of the pairs still to test
    uint32_t item_first, item_kind;  // the items of the level-1 node above the cursor: first primitive, kind,
    unsigned long long item_counts;  // primitives per item (a byte each)
    uint32_t state;
};
// Whether the ray's interval (s, e) may touch a child box, given as three words of two u16 grid values (x, y, z:
// start | end << 16).  Nothing depends on the exact outcome -- a box is passed over only when this says it cannot be
// touched -- so the test is written for speed: f32, one fused multiply-add per plane, no branches; the boxes are a grid
// step wider than the geometry on every side for it (see wide_begin), and the caller rounds s down and e up.  A NaN
// (0 * inf: a ray in a box's plane) drops out of min / max, which leaves that axis undecided: the box counts as touched.
RT_HD bool wide_box(uint32_t wx, uint32_t wy, uint32_t wz, const WLane& ln, float s, float e, float& entry) {
    const float x0 = fmaf((float)(wx & 0xFFFFu), ln.ax, ln.bx), x1 = fmaf((float)(wx >> 16), ln.ax, ln.bx);
    const float y0 = fmaf((float)(wy & 0xFFFFu), ln.ay, ln.by), y1 = fmaf((float)(wy >> 16), ln.ay, ln.by);
    const float z0 = fmaf((float)(wz & 0xFFFFu), ln.az, ln.bz), z1 = fmaf((float)(wz >> 16), ln.az, ln.bz);
    const float tn = fmaxf(fmaxf(fminf(x0, x1), fminf(y0, y1)), fmaxf(fminf(z0, z1), s));
    const float tf = fminf(fminf(fmaxf(x0, x1), fmaxf(y0, y1)), fminf(fmaxf(z0, z1), e));
    entry = tn;
    return !(tf <= tn);
}
RT_HD void wide_begin(WLane& ln, const DScene& S, const DWide& W, const Ray& world, double tmin, double tmax, Best best) {
    ln.cur = ray_in_chain(S, world, W.chain);
    ray_need_inv(ln.cur);
    ln.ax = (float)(W.step[0] * ln.cur.inv.x); ln.ay = (float)(W.step[1] * ln.cur.inv.y); ln.az = (float)(W.step[2] * ln.cur.inv.z);
    ln.bx = (float)((W.base[0] - ln.cur.o.x) * ln.cur.inv.x); ln.by = (float)((W.base[1] - ln.cur.o.y) * ln.cur.inv.y);
    ln.bz = (float)((W.base[2] - ln.cur.o.z) * ln.cur.inv.z);
    // The f32 evaluation is off by at most 2^-23 (|grid offset| + |base - origin|) in space, which the extra grid step
    // around every box (2^-16 of the tree's extent) covers as long as the origin is within ~100 extents of the tree;
    // a ray from farther away leaves every axis undecided (NaN) and visits every node.
    const double far = 100.0 * 65530.0;
    if (!(fabs(W.base[0] - ln.cur.o.x) < far * W.step[0] && fabs(W.base[1] - ln.cur.o.y) < far * W.step[1] &&
          fabs(W.base[2] - ln.cur.o.z) < far * W.step[2])) {
        ln.bx = ln.by = ln.bz = __builtin_nanf("");
    }
    ln.cur_chain = W.chain;
    ln.tmin = tmin; ln.tmax = tmax; ln.s = tmin; ln.e = tmax;
    ln.best = best;
    ln.pending = 0ull;
    ln.rev = 0u;
    ln.slot = 0u; ln.level = W.K;
    ln.k = ln.kend = 0u; ln.kind = 0u;
    ln.item_first = 0u; ln.item_kind = 0u; ln.item_counts = 0ull;
    ln.state = RT_W_NODE;
}
// Moves the cursor to child j of the level-k node above it: a node of level k - 1, or (k = 1) an item, whose run of
// primitives starts.  Written without branches and with every field assigned on every path: fields stored in one
// branch only end up addressed through a selected pointer, and the whole lane state then lives in scratch memory.
RT_HD void wide_goto(WLane& ln, const DWide& W, uint32_t k, uint32_t j) {
    ln.slot = ((ln.slot >> (RT_WIDE_LOG * k)) << (RT_WIDE_LOG * k)) | (j << (RT_WIDE_LOG * (k - 1u)));
    // item j's run: the primitives after those of items 0 .. j - 1 (the bytes below byte j summed by one multiply)
    const unsigned long long below = ln.item_counts & ((1ull << (8u * j)) - 1ull);
    const uint32_t first = ln.item_first + (uint32_t)((below * 0x0101010101010101ull) >> 56);
    const uint32_t count = (uint32_t)(ln.item_counts >> (8u * j)) & 0xFFu;
    const bool item = k == 1u;
    ln.kind = item ? ((ln.item_kind & 0xFFu) | ((W.lambda == 0u || (ln.rev >> 31) != 0u) ? 0x200u : 0u) | (((ln.item_kind >> (16u + j)) & 1u) != 0u ? 0x400u : 0u)) : ln.kind;
    ln.k = item ? first : ln.k;
    ln.kend = item ? first + count : ln.kend;
    ln.level = k - 1u;
    ln.state = item ? RT_W_PRIM : RT_W_NODE;
}
RT_HD uint32_t ctz64(unsigned long long x) {
#if defined(__HIP_DEVICE_COMPILE__)
    return (uint32_t)(__ffsll((long long)x) - 1);
#else
    return (uint32_t)__builtin_ctzll(x);
#endif
}
RT_HD uint32_t top_bit8(uint32_t g) {  // index of the highest set bit of g in 1..255
#if defined(__HIP_DEVICE_COMPILE__)
    return 31u - (uint32_t)__clz((int)g);
#else
    return 31u - (uint32_t)__builtin_clz(g);
#endif
}
// On to the nearest child still to be visited of the nodes above the cursor, or done.
RT_HD void wide_ascend(WLane& ln, const DWide& W) {
    const bool none = ln.pending == 0ull;
    const uint32_t p = none ? 8u : ctz64(ln.pending);
    const uint32_t k = p >> 3;
    uint32_t j = p & 7u;
    if (W.fast) {  // the highest-numbered child first where the node's order is reversed
        const uint32_t g = (uint32_t)(ln.pending >> (8u * k)) & 0xFFu;
        j = (((ln.rev >> k) & 1u) != 0u && !none) ? top_bit8(g | 1u) : j;
    }
    ln.pending &= ~((none ? 0ull : 1ull) << (8u * k + j));
    wide_goto(ln, W, k, j);
    ln.state = none ? RT_W_DONE : ln.state;
}
// One node for a lane in RT_W_NODE: its eight child boxes.
struct WNodeRec { uint4 q0, q1, q2, q3, q4, q5, tail; };  // the first 112 bytes of a node
RT_HD uint32_t wide_cursor_node(const DWide& W, const WLane& ln) { return wide_node_index(W.K, ln.level, ln.slot >> (RT_WIDE_LOG * ln.level)); }
// `tally` (counting passes only): node_tests counts the reference's box test of a leaf, wide_tests the child boxes.
RT_HD void wide_node_step(const DWide& W, WLane& ln, const WNodeRec& rec, Tally* tally = nullptr) {
    const uint32_t k = ln.level;
    const uint4 q0 = rec.q0, q1 = rec.q1, q2 = rec.q2, q3 = rec.q3, q4 = rec.q4, q5 = rec.q5;
    uint32_t axis = 0u;
    if (W.fast || k == 1u) {
        const uint4 tail = rec.tail;
        axis = (tail.y >> 8) & 7u;
        // (assigned on every path, see wide_goto)
        ln.item_first = k == 1u ? tail.x : ln.item_first;
        ln.item_kind = k == 1u ? tail.y : ln.item_kind;
        ln.item_counts = k == 1u ? ((unsigned long long)tail.w << 32 | tail.z) : ln.item_counts;
    }
    // A leaf's node (lambda = 1) leaves BVHNode::hit's box test for the leaf (bvh.rs:31-37) to the first of its items
    // the walk goes to (wide_prim_step): most rays that get here touch none of the items' boxes, and the test's three
    // loads and f64 arithmetic would be paid by the whole wavefront in every step any lane stands on such a node.
    ln.rev = (W.lambda != 0u && k == W.lambda) ? (ln.rev | 0x80000000u) : ln.rev;
    if (tally) tally->wide_tests += RT_WIDE;
    const double ds = ln.tmin, de = fmin(ln.tmax, ln.best.t);
    float cs = (float)ds, ce = (float)de;
    cs = fmaf(-fabsf(cs), 1.1920929e-7f, cs);  // not above ds, not below de (an infinity stays one)
    ce = fmaf(fabsf(ce), 1.1920929e-7f, ce);
    uint32_t m = 0u, nearest = 0u;
    float t, least = __builtin_inff();
    // (the child whose box the ray enters first, for the fast mode: a running minimum, the lowest child on ties)
#define RT_WIDE_CHILD(J, WX, WY, WZ) \
    if (wide_box(WX, WY, WZ, ln, cs, ce, t)) { m |= 1u << J; const bool nearer = t < least; nearest = nearer ? J : nearest; least = nearer ? t : least; }
    RT_WIDE_CHILD(0u, q0.x, q0.y, q0.z)
    RT_WIDE_CHILD(1u, q0.w, q1.x, q1.y)
    RT_WIDE_CHILD(2u, q1.z, q1.w, q2.x)
    RT_WIDE_CHILD(3u, q2.y, q2.z, q2.w)
    RT_WIDE_CHILD(4u, q3.x, q3.y, q3.z)
    RT_WIDE_CHILD(5u, q3.w, q4.x, q4.y)
    RT_WIDE_CHILD(6u, q4.z, q4.w, q5.x)
    RT_WIDE_CHILD(7u, q5.y, q5.z, q5.w)
#undef RT_WIDE_CHILD
    if (!W.fast) {
        // The children to visit join those of the nodes above, and the walk goes on with the nearest of them all: the
        // lowest set bit is this node's first such child when it has one (no bit of a lower level is set while the
        // cursor stands here), else the next sibling of an ancestor -- one path for both, without a branch.
        ln.pending |= (unsigned long long)m << (8u * k);
        wide_ascend(ln, W);
        return;
    }
    if (m == 0u) { wide_ascend(ln, W); return; }
#if defined(__HIP_DEVICE_COMPILE__)
    uint32_t j = (uint32_t)(__ffs((int)m) - 1);
#else
    uint32_t j = (uint32_t)__builtin_ctz(m);
#endif
    if (W.fast) {
        // Nearest first: the child whose box the ray enters first now; the others later from the near end of the node,
        // judged by one axis: children 0..3 lie beyond (axis bit 2: before) children 4..7 along `axis`, so a ray that
        // runs up that axis meets 4..7 first.  Whatever the order, every child whose box the interval can touch is
        // visited unless a closer hit has shortened the interval meanwhile; only hits at exactly equal parameters can
        // come out differently from the reference's order.
        // (comparisons first, then the choice between their results: a choice between the fields themselves would be
        // turned into a load through a selected address, and the lane state would live in scratch memory)
        const bool px = ln.cur.d.x > 0.0, py = ln.cur.d.y > 0.0, pz = ln.cur.d.z > 0.0;
        const uint32_t a = axis & 3u;
        const bool up = (a == 0u && px) || (a == 1u && py) || (a == 2u && pz);
        const bool reversed = ((axis & 4u) != 0u) != up;
        j = least < __builtin_inff() ? nearest : (reversed ? top_bit8(m) : j);  // (every entry a NaN: any order)
        ln.rev = (ln.rev & ~(1u << k)) | ((reversed ? 1u : 0u) << k);
    }
    m &= ~(1u << j);
    ln.pending |= (unsigned long long)m << (8u * k);
    wide_goto(ln, W, k, j);
}
// The same with the node read by the lane itself.
RT_HD void wide_node_step(const DWide& W, WLane& ln, Tally* tally = nullptr) {
    const uint4* rp = W.nodes + 8u * wide_cursor_node(W, ln);
    WNodeRec rec;
    rec.q0 = rp[0]; rec.q1 = rp[1]; rec.q2 = rp[2]; rec.q3 = rp[3]; rec.q4 = rp[4]; rec.q5 = rp[5]; rec.tail = rp[6];
    wide_node_step(W, ln, rec, tally);
}
// The next primitives of the run in progress for a lane in RT_W_PRIM (lane_prim_tests: two planar primitives or four
// spheres per step).
// `Away` says where the parts of the lane state live that the primitive tests do not touch: the kernel keeps them in
// LDS while those run (its registers are what decides how many wavefronts fit a SIMD) and brings them back here, where
// they are needed again.  The default keeps everything in the lane.
struct WideKeepsAll {
    RT_HD void fetch_ray(WLane&) const {}      // cur.inv, tmin, tmax
    RT_HD void fetch_cursor(WLane&) const {}   // pending, rev, slot, level, item_*
    RT_HD void fetch_grid(WLane&) const {}     // ax .. bz
    RT_HD void stash_cursor(const WLane&) const {}
};
// Which pairs of a box's six sides (create_box's order: z = max, x = max | z = min, x = min | y = max, y = min) are worth
// testing first, for a lane that stands on such an item with its cursor and grid coefficients at hand.  The reference tests
// all six and keeps the closest hit; a ray meets a box's surface where it enters -- the side of the last axis to be
// entered -- or, when that lies before its interval, where it leaves: the side of the first axis to be left.  Those
// axes are read off the item's box on the grid (child j of the level-1 node above the cursor), whose planes lie within
// slack[a] grid steps outside the box's own: side (a, entering) is a candidate when a's plane can be the last entered and
// that can be inside the interval, side (a, leaving) when a's plane can be the first left and the entry can be before the
// interval -- or at the exit itself, for a ray that only touches a corner or an edge.  Anything not finite (a ray parallel to a pair of sides, an origin too far away for the grid) makes every
// pair a candidate.  0: the box lies wholly beyond the interval.  The caller tests the other pairs as well when the chosen ones give no hit, so the choice cannot
// lose a hit; it cannot change one either: a side that is not a candidate is separated from the entry (or exit) point by
// more than the slack, far beyond what the rounding of a quad's own test can bridge.
RT_HD const uint32_t* wide_item_box(const DWide& W, const WLane& ln) {  // the item's box on the grid: three words, as wide_box takes them
    return reinterpret_cast<const uint32_t*>(W.nodes + 8u * wide_node_index(W.K, 1u, ln.slot >> RT_WIDE_LOG)) + 3u * (ln.slot & 7u);
}
RT_HD uint32_t wide_box_pairs(const DWide& W, const WLane& ln, uint32_t wx, uint32_t wy, uint32_t wz) {
    const float x0 = fmaf((float)(wx & 0xFFFFu), ln.ax, ln.bx), x1 = fmaf((float)(wx >> 16), ln.ax, ln.bx);
    const float y0 = fmaf((float)(wy & 0xFFFFu), ln.ay, ln.by), y1 = fmaf((float)(wy >> 16), ln.ay, ln.by);
    const float z0 = fmaf((float)(wz & 0xFFFFu), ln.az, ln.bz), z1 = fmaf((float)(wz >> 16), ln.az, ln.bz);
    const float nx = fminf(x0, x1), fx = fmaxf(x0, x1), ny = fminf(y0, y1), fy = fmaxf(y0, y1), nz = fminf(z0, z1), fz = fmaxf(z0, z1);
    const float mx = W.slack[0] * fabsf(ln.ax), my = W.slack[1] * fabsf(ln.ay), mz = W.slack[2] * fabsf(ln.az);
    const bool finite = fabsf(x0) + fabsf(x1) + fabsf(y0) + fabsf(y1) + fabsf(z0) + fabsf(z1) + mx + my + mz < 1e30f;
    const float tn = fmaxf(fmaxf(nx, ny), nz), tf = fminf(fminf(fx, fy), fz);
    const float s = (float)ln.s, sm = fabsf(s) * 4.8e-7f;
    const bool entry = fmaxf(fmaxf(nx + mx, ny + my), nz + mz) >= s - sm;  // the entry point may lie inside the interval
    // ... or before it; or the ray may only graze the box, entering where it leaves (a corner, an edge): then the sides
    // it leaves by are hit at the same parameter as those it enters by, and the reference's order decides between them
    const bool leave = tn <= s + sm || tf - (mx + my + mz) <= tn + (mx + my + mz);
    // pairs: 0 = {z max, x max}, 1 = {z min, x min}, 2 = {y max, y min}; a ray running up an axis enters by its min side
    const uint32_t px_in = ln.cur.d.x > 0.0 ? 2u : 1u, pz_in = ln.cur.d.z > 0.0 ? 2u : 1u;
    const uint32_t px_out = 3u - px_in, pz_out = 3u - pz_in;
    uint32_t mask = 0u;
    mask |= (entry && nx + mx >= tn) ? px_in : 0u;
    mask |= (entry && nz + mz >= tn) ? pz_in : 0u;
    mask |= (entry && ny + my >= tn) ? 4u : 0u;
    mask |= (leave && fx - mx <= tf) ? px_out : 0u;
    mask |= (leave && fz - mz <= tf) ? pz_out : 0u;
    mask |= (leave && fy - my <= tf) ? 4u : 0u;
    // Nothing at all when the box begins beyond the interval's end as it stands now (the node above looked at it with
    // the end of that moment; closer hits in the items before this one may have shortened it since)
    const float e = (float)ln.e, em = fabsf(e) * 4.8e-7f;
    if (finite && tn > e + em) return 0u;
    return (finite && mask != 0u) ? mask : 7u;
}
template <bool COUNT, class Away = WideKeepsAll, int ONLY = 0>
RT_HD void wide_prim_step(const DScene& S, const DWide& W, WLane& ln, Tally& tally, const Away& away = Away()) {
    // What an item's first step may have to do before its primitives: the leaf's box narrows the interval (the item is a
    // whole leaf, lambda = 0, or the first one the walk has come to below a leaf's node, lambda = 1), and a box has its
    // sides chosen.  Both read a few words whose addresses the cursor gives: the loads go out together, ahead of either use.
    const bool narrow = (ln.kind & 0x200u) != 0u;
    const bool boxed = (ONLY == 0 || ONLY == RT_NODE_QUADS) && (ln.kind & 0x400u) != 0u;
    if (narrow || boxed) {
        away.fetch_cursor(ln);
        double2 b0, b1, b2;
        b0.x = b0.y = b1.x = b1.y = b2.x = b2.y = 0.0;
        uint32_t wx = 0u, wy = 0u, wz = 0u;
        if (narrow) { const double2* rb = W.refbox + 3u * (ln.slot >> (RT_WIDE_LOG * W.lambda)); b0 = rb[0]; b1 = rb[1]; b2 = rb[2]; }
        if (boxed) { const uint32_t* words = wide_item_box(W, ln); wx = words[0]; wy = words[1]; wz = words[2]; }
#if defined(__HIP_DEVICE_COMPILE__)
        // (also keeps the compiler from sinking the loads into the slab test's branches, where each would wait on its own)
        asm volatile("" ::"v"(b0.x), "v"(b1.x), "v"(b2.x), "v"(wx), "v"(wy), "v"(wz));
#endif
        if (narrow) {
            ln.kind &= ~0x200u;
            away.fetch_ray(ln);
            double s = ln.tmin, e = fmin(ln.tmax, ln.best.t);
            if (COUNT) tally.node_tests++;
            // (no early exit: a leaf that is missed has its run ended here and moves on with the lanes whose runs are over)
            const bool ok = slab3_flat(b0, b1, b2, ln.cur, s, e);
            ln.s = ok ? s : ln.s; ln.e = ok ? e : ln.e;
            ln.k = ok ? ln.k : ln.kend;
            if (W.lambda != 0u) {
                // done for this node's other items too; a leaf that is missed has none of them visited
                ln.rev &= 0x7FFFFFFFu;
                ln.pending = ok ? ln.pending : (ln.pending & ~(0xFFull << 8));
                away.stash_cursor(ln);
            }
        }
        if (boxed && ln.k < ln.kend) {  // a box: the sides the ray can meet first, the others only if those are missed
            away.fetch_grid(ln);
            const uint32_t mask = wide_box_pairs(W, ln, wx, wy, wz);
            ln.kind = (ln.kind & 0x3FFu) | 0x100u | (mask != 0u ? 0x800u : 0u) | mask << 12;
            lane_skip_pairs(ln, mask);  // (no pair at all: straight to the item's end)
        }
    }
    if (ln.k < ln.kend) lane_prim_tests<COUNT, ONLY>(S, ln, tally);
    if (ln.k >= ln.kend) {
        if constexpr (ONLY == 0 || ONLY == RT_NODE_QUADS) {
            if ((ln.kind & 0x800u) != 0u) {
                const uint32_t first = ln.kend - 6u, rest = ~(ln.kind >> 12) & 7u;
                const bool hit_here = (ln.best.ref >> RT_REF_KIND_SHIFT) == RT_NODE_QUADS && (ln.best.ref & RT_REF_INDEX_MASK) - first < 6u;
                ln.kind &= ~0x800u;
                if (!hit_here && rest != 0u) { ln.k = first; lane_skip_pairs(ln, rest); return; }
            }
        }
        away.fetch_cursor(ln); wide_ascend(ln, W); away.stash_cursor(ln);
    }
}

}  // namespace rt
