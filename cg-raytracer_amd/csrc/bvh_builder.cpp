// bvh_builder.cpp -- the reference's BVH construction (src/bounding_volume_hierarchy.cpp:42-76,
// :88-134, :168-207, :235-372) on a flat layout.
//
// The reference deep-copies every vertex of a mesh into every node (bvh.cpp:205-206) and so cannot
// build dragon-sized scenes.  Here one permutation array `order` holds the primitive
// ids; every node owns a contiguous range of it, subdivided into "runs" (one per reference Mesh held
// by the node).  What is kept exactly, because traversal results depend on it:
//   * split axis rule (:289), median split at size()/2 with the middle element going right (:201-202),
//   * libstdc++ std::sort with the same comparison outcomes => same order among equal keys,
//   * mesh-list split for multi-mesh nodes, keyed by each mesh's median-triangle centroid (:88-110),
//   * child boxes over triangle-referenced vertices only (:235-268),
//   * leaf rule: level 11, or one mesh with one triangle (:320-322), breadth-first numbering (:343-372).
#include "bvh_builder.h"

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <thread>
#include <utility>

#include "cgrt_math.h"

namespace cgrt {
namespace {

struct Run {
    int mesh;
    uint32_t first, count;  // sub-range of `order`
};

struct Builder {
    const HostScene& sc;
    std::vector<uint32_t>& order;
    std::vector<float> cen[3];  // per-prim centroid coordinate per axis (bvh.cpp:126)

    Builder(const HostScene& s, std::vector<uint32_t>& ord) : sc(s), order(ord) {}

    const float* vpos(uint32_t v) const { return &sc.pos_nrm[6 * (size_t)v]; }

    void centroids() {
        for (int a = 0; a < 3; a++) cen[a].resize(sc.ntris);
        for (uint32_t p = 0; p < sc.ntris; p++) {
            const float *A = vpos(sc.tri[3 * p]), *B = vpos(sc.tri[3 * p + 1]), *C = vpos(sc.tri[3 * p + 2]);
            for (int a = 0; a < 3; a++) cen[a][p] = ((A[a] + B[a]) + C[a]) / 3.0f;
        }
    }

    // std::sort of order[first, first+count) by centroid coordinate (bvh.cpp:122-134).  Sorting
    // (key, id) pairs with a key-only comparator makes libstdc++'s introsort take the same
    // decisions, hence the same permutation, as sorting the ids with the reference's comparator.
    void sort_range(uint32_t first, uint32_t count, int axis, std::vector<std::pair<float, uint32_t>>& tmp) {
        tmp.resize(count);
        const std::vector<float>& key = cen[axis];
        for (uint32_t i = 0; i < count; i++) tmp[i] = std::make_pair(key[order[first + i]], order[first + i]);
        std::sort(tmp.begin(), tmp.end(),
                  [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) { return a.first < b.first; });
        for (uint32_t i = 0; i < count; i++) order[first + i] = tmp[i].second;
    }

    Box6 bounds(uint32_t first, uint32_t count) const {  // bvh.cpp:235-268
        // :238 routes the first vertex index through a float
        float fi = (float)sc.tri[3 * (size_t)order[first]];
        const float* f = vpos((uint32_t)(size_t)fi);
        float mn[3] = {f[0], f[1], f[2]}, mx[3] = {f[0], f[1], f[2]};
        for (uint32_t i = 0; i < count; i++) {
            const uint32_t p = order[first + i];
            for (int k = 0; k < 3; k++) {
                const float* v = vpos(sc.tri[3 * (size_t)p + k]);
                for (int a = 0; a < 3; a++) {
                    mn[a] = (v[a] < mn[a]) ? v[a] : mn[a];
                    mx[a] = (v[a] > mx[a]) ? v[a] : mx[a];
                }
            }
        }
        Box6 b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = mn[a];
            b.hi[a] = mx[a];
        }
        return b;
    }
};


// ------------------------------------------------------------------------------------------------
// In-leaf accelerator (DESIGN.md "In-leaf accelerator").
//
// The reference caps its tree at 12 levels, so big meshes end in fat leaves (~390 triangles at 800 K)
// that intersectLeaf scans linearly (bvh.cpp:535-553).  The scan's outcome is a pure function of the
// leaf's triangle SET plus each triangle's scan position: the smallest accepted t wins, equal t goes to
// the earlier scan position, and an origin-on-plane acceptance (t = 0, ray_tracing.cpp:43-47) goes to
// the LAST such triangle.  So the kernel may test the triangles in any order -- and skip any triangle
// that provably cannot be accepted -- as long as it applies that rule.  This builds, per reference
// leaf, a 4-wide BVH (surface-area heuristic, full sweep on every axis) used only to skip triangles
// whose padded box the ray misses or enters beyond the current best t.  The reference topology, its
// visit order and its culling decisions above the leaves are untouched.
//
// Depth is bounded (SUB_MAX_DEPTH) because the per-ray stack is a fixed LDS slice: a split is only
// eligible if both sides still fit under the remaining depth.
struct SubBuilder {
    std::vector<TriRecord>& tris;
    std::vector<TriNormals>& nrms;
    std::vector<SubNode>& nodes;
    int leaf_tris;
    // per-leaf scratch
    std::vector<Box6> tb;        // triangle boxes (leaf-local)
    std::vector<float> cen[3];
    std::vector<uint32_t> idx;   // permutation being built (leaf-local indices)
    std::vector<float> rarea;

    static Box6 empty_box() {
        Box6 b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::numeric_limits<float>::infinity();
            b.hi[a] = -std::numeric_limits<float>::infinity();
        }
        return b;
    }
    static void grow(Box6& b, const Box6& o) {
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::min(b.lo[a], o.lo[a]);
            b.hi[a] = std::max(b.hi[a], o.hi[a]);
        }
    }
    static float half_area(const Box6& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
    Box6 range_box(uint32_t b, uint32_t e) const {
        Box6 r = empty_box();
        for (uint32_t i = b; i < e; i++) grow(r, tb[idx[i]]);
        return r;
    }
    // triangles a subtree with depth_left levels can hold
    static uint64_t capacity(int depth_left, int leaf_tris) {
        uint64_t c = (uint64_t)leaf_tris;
        for (int i = 0; i < depth_left; i++) c *= (uint64_t)SUB_WIDTH;
        return c;
    }

    // Splits idx[b, e) in two (SAH, full sweep on every axis) and returns the position of the split; each side
    // may hold at most `cap` triangles (what still fits under the remaining depth).
    uint32_t split(uint32_t b, uint32_t e, uint64_t cap) {
        const uint32_t n = e - b;
        const uint32_t kmin = (uint32_t)std::max<int64_t>(1, (int64_t)n - (int64_t)cap);
        const uint32_t kmax = (uint32_t)std::min<uint64_t>(n - 1, cap);
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1;
        uint32_t best_k = n / 2;
        std::vector<uint32_t> ord[3];
        rarea.resize(n);
        for (int a = 0; a < 3; a++) {
            ord[a].assign(idx.begin() + b, idx.begin() + e);
            std::stable_sort(ord[a].begin(), ord[a].end(), [&](uint32_t p, uint32_t q) { return cen[a][p] < cen[a][q]; });
            Box6 acc = empty_box();
            for (uint32_t i = n; i-- > 1;) {
                grow(acc, tb[ord[a][i]]);
                rarea[i] = half_area(acc);
            }
            acc = empty_box();
            for (uint32_t k = 1; k < n; k++) {
                grow(acc, tb[ord[a][k - 1]]);
                if (k < kmin || k > kmax) continue;
                const float cost = half_area(acc) * (float)k + rarea[k] * (float)(n - k);  // surface-area heuristic
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_k = k;
                }
            }
        }
        if (best_axis < 0)  // NaN geometry: any split inside [kmin, kmax] keeps the depth bound
            best_k = std::min(std::max(n / 2, kmin), kmax);
        else
            std::copy(ord[best_axis].begin(), ord[best_axis].end(), idx.begin() + b);
        return b + best_k;
    }

    uint32_t child_ref(uint32_t base, uint32_t b, uint32_t e, int depth_left) {
        const uint32_t cn = e - b;
        // a child becomes a run of records when it is small enough or no depth is left
        if (cn <= (uint32_t)leaf_tris || depth_left <= 0) return REF_LEAF | ((cn - 1) << 26) | (base + b);
        return build4(base, b, e, depth_left);
    }

    // 4-wide node over idx[b, e) (more than leaf_tris triangles): split in two, then each half again; the node is two
    // consecutive SubNode records {children 0,1} {children 2,3}; a missing child has an empty box (never hit).
    uint32_t build4(uint32_t base, uint32_t b, uint32_t e, int depth_left) {
        const uint32_t me = (uint32_t)nodes.size();
        nodes.push_back(SubNode());
        nodes.push_back(SubNode());
        const uint64_t cap1 = capacity(depth_left - 1, leaf_tris);
        const uint32_t m = split(b, e, 2 * cap1);
        uint32_t cb[4], ce[4];
        int nc = 0;
        const uint32_t hb[2] = {b, m}, he[2] = {m, e};
        for (int h = 0; h < 2; h++) {
            if (he[h] - hb[h] > (uint32_t)leaf_tris) {
                const uint32_t q = split(hb[h], he[h], cap1);
                cb[nc] = hb[h], ce[nc++] = q;
                cb[nc] = q, ce[nc++] = he[h];
            } else {
                cb[nc] = hb[h], ce[nc++] = he[h];
            }
        }
        Box6 box[4];
        uint32_t ref[4];
        for (int c = 0; c < 4; c++) {
            if (c < nc) {
                box[c] = range_box(cb[c], ce[c]);
                ref[c] = child_ref(base, cb[c], ce[c], depth_left - 1);
            } else {
                box[c] = empty_box();
                ref[c] = REF_NONE;
            }
        }
        for (int r = 0; r < 2; r++) {
            SubNode& N = nodes[me + r];
            sub_box_store(N.box0, box[2 * r]);
            sub_box_store(N.box1, box[2 * r + 1]);
            N.ref0 = ref[2 * r];
            N.ref1 = ref[2 * r + 1];
            N.pad[0] = N.pad[1] = 0;
        }
        return me;
    }

    void run(LeafRec& L) {
        const uint32_t n = L.count;
        if (n <= (uint32_t)leaf_tris) return;  // small leaf: all candidates are tested
        int lt = leaf_tris;
        while (capacity(SUB_MAX_DEPTH, lt) < n) lt *= 2;  // giant leaf: coarser sub-leaves, same depth bound
        if ((uint32_t)lt > SUB_RUN_MAX || (uint64_t)L.first + n > SUB_MAX_RECORDS) return;  // beyond the encoding: scanned linearly
        const int saved = leaf_tris;
        leaf_tris = lt;
        tb.resize(n);
        idx.resize(n);
        for (int a = 0; a < 3; a++) cen[a].resize(n);
        for (uint32_t i = 0; i < n; i++) {
            const TriRecord& T = tris[L.first + i];
            for (int a = 0; a < 3; a++) {
                const float lo = std::min(T.v0[a], std::min(T.v1[a], T.v2[a]));
                const float hi = std::max(T.v0[a], std::max(T.v1[a], T.v2[a]));
                tb[i].lo[a] = lo;
                tb[i].hi[a] = hi;
                cen[a][i] = 0.5f * lo + 0.5f * hi;
            }
            idx[i] = i;
        }
        L.sub_root = build4(L.first, 0, n, SUB_MAX_DEPTH);
        leaf_tris = saved;
        // apply the permutation to the records (scan_k keeps every triangle's reference scan position)
        std::vector<TriRecord> t2(n);
        std::vector<TriNormals> n2(n);
        for (uint32_t i = 0; i < n; i++) {
            t2[i] = tris[L.first + idx[i]];
            n2[i] = nrms[L.first + idx[i]];
        }
        std::copy(t2.begin(), t2.end(), tris.begin() + L.first);
        std::copy(n2.begin(), n2.end(), nrms.begin() + L.first);
    }
};

// Can the reference accept this triangle only at a point of its own box (up to rounding)?  That is what lets the kernels
// skip a triangle whose widened box the ray misses.  It holds whenever the stored plane normal n = normalize(cross(e1, e2))
// (ray_tracing.cpp:74-82, float) is a finite non-zero vector along the true normal.  It fails when the float arithmetic
// degenerates: |cross|^2 overflows -> 1/sqrt(inf) = 0 -> n = (0, 0, 0), D = 0, and then `dot(o, n) == D` holds for EVERY ray
// (the origin-on-plane rule, :43-47, accepts it at t = 0 wherever the ray is); |cross|^2 underflows -> n = +-inf / NaN with
// inf == inf comparisons; a cross product so small that its components are subnormal points anywhere.  Such "wild" triangles
// must be tested by every ray that reaches their leaf: a leaf that holds one is scanned linearly, and the scene gets no fast
// tree (tests/test_adversarial_gpu.py scales a scene until this happens).
static bool plane_is_tame(const float* a, const float* b, const float* c, const F3 n) {
    // a NaN component (zero-area triangle: 0 * inf) makes every dot product with n NaN: no comparison succeeds, the triangle
    // is never accepted -- inert, and skipping it is always right
    if (n.x != n.x || n.y != n.y || n.z != n.z) return true;
    if (!(std::fabs(n.x) <= 4.0f && std::fabs(n.y) <= 4.0f && std::fabs(n.z) <= 4.0f)) return false;  // +-inf
    const double e1[3] = {(double)b[0] - a[0], (double)b[1] - a[1], (double)b[2] - a[2]};
    const double e2[3] = {(double)c[0] - a[0], (double)c[1] - a[1], (double)c[2] - a[2]};
    const double cx = e1[1] * e2[2] - e2[1] * e1[2], cy = e1[2] * e2[0] - e2[2] * e1[0], cz = e1[0] * e2[1] - e2[0] * e1[1];
    const double len2 = cx * cx + cy * cy + cz * cz, n2 = (double)n.x * n.x + (double)n.y * n.y + (double)n.z * n.z;
    if (!(n2 >= 0.25 && n2 <= 4.0)) return false;              // (0, 0, 0) from an overflowed length, or a length that lost its precision
    if (!(len2 >= 0x1p-200 && len2 <= 0x1p+200)) return false;  // |cross| in [2^-100, 2^100]: its float components carry full precision
    return true;
}

// The leaves are independent: contiguous ranges of them are built on worker threads into private node arrays, which are
// then concatenated in leaf order -- the result is the array a single thread would have produced, node for node.
void build_leaf_accelerators(BuiltBvh& out, int leaf_tris, int threads) {
    const int lt = leaf_tris < 1 ? 1 : leaf_tris;
    const size_t nleaves = out.leaves.size();
    unsigned nthreads = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1 || nleaves < 64) nthreads = 1;
    std::vector<std::vector<SubNode>> part(nthreads);
    auto work = [&](unsigned t) {
        SubBuilder sb{out.tris, out.tri_normals, part[t], lt, {}, {}, {}, {}};  // leaves own disjoint record ranges
        const size_t b = nleaves * t / nthreads, e = nleaves * (t + 1) / nthreads;
        for (size_t i = b; i < e; i++)
            if (!out.leaf_wild[i]) sb.run(out.leaves[i]);  // a leaf with a wild triangle keeps the reference's linear scan
    };
    if (nthreads == 1) {
        work(0);
    } else {
        std::vector<std::thread> pool;
        for (unsigned t = 0; t < nthreads; t++) pool.emplace_back(work, t);
        for (std::thread& th : pool) th.join();
    }
    for (unsigned t = 0; t < nthreads; t++) {
        const uint32_t base = (uint32_t)out.subnodes.size();
        for (SubNode N : part[t]) {
            for (uint32_t* r : {&N.ref0, &N.ref1})
                if (*r != REF_NONE && !(*r & REF_LEAF)) *r += base;  // node references were local to the part
            out.subnodes.push_back(N);
        }
        const size_t b = nleaves * t / nthreads, e = nleaves * (t + 1) / nthreads;
        for (size_t i = b; i < e; i++)
            if (out.leaves[i].sub_root != REF_NONE) out.leaves[i].sub_root += base;
    }
}


// ------------------------------------------------------------------------------------------------
// Fast tree + certificates (DESIGN.md "Certified walk", walk_fast.h).
//
// The reference's own tree is a poor search structure (median splits, 12 levels, every step two exact box tests), but
// its culling quirks are part of the result, so the exact walk must take every one of its steps.  The
// certified walk searches a better structure first and then PROVES that the reference's walk ends in the same hit:
//   * the fast tree: a 4-wide SAH tree over the reference LEAVES (their exact boxes), whose children are the leaves'
//     in-leaf accelerators -- the closest accepted triangle over ALL leaves, found with conservative box tests only;
//   * the certificate: for the leaf of that triangle, the boxes the reference tests on its way from the root to the
//     leaf (paths).  If every one of them is entered at the final t (origin strictly inside, or the exact box test
//     passes with parameter < t), the reference reaches the leaf whatever it met before -- ray.t never drops below the
//     final t -- and accepts the same triangle; anything else (a failed box, an equal-t tie, an origin-on-plane
//     acceptance) sends the ray through the exact walk.
// Nothing here changes what the exact walk reads.
// Threads.  The recursion is a pre-order walk (a node, then its four subtrees one after the other), each subtree a pure function
// of its own range of idx.  The top TOP_PAR_LEVELS levels are built by the caller's thread into a vector of their own; every
// subtree below them that holds more than TOP_TASK_MIN items becomes a task built by a worker thread into a vector of ITS own,
// and stitch() then emits top nodes and task vectors in the sequential build's pre-order: the arrays are identical to the
// single-threaded build's, node for node (tests/test_builder.py compares them).  While a part is being built a reference to one
// of its own new nodes carries TOP_LOCAL (index into the part), a reference to a task TOP_TASK (task number); item references
// (accelerator roots, runs) are absolute and carry neither.
static const uint32_t TOP_LOCAL = 0x20000000u, TOP_TASK = 0x10000000u;  // free bits of a node reference (indices use 26)
static const int TOP_PAR_LEVELS = 2;
static const uint32_t TOP_TASK_MIN = 4096;
struct TopTask {
    uint32_t b, e;
    int depth_left;
    std::vector<SubNode> nodes;
    uint32_t root = REF_NONE;
};
struct TopBuilder {
    const std::vector<Box6>& box;     // item boxes
    const std::vector<uint32_t>& ref; // item references (accelerator roots or runs, builder-local indices)
    std::vector<SubNode>& nodes;      // this part's new nodes
    std::vector<uint32_t>& idx;       // shared; a part only touches its own range
    const std::vector<float>* cen;    // [3], shared, read-only
    std::vector<TopTask>* tasks;      // where the top part leaves its tasks (nullptr: build everything here)
    std::vector<float> rarea;

    static Box6 empty_box() {
        Box6 b;
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::numeric_limits<float>::infinity();
            b.hi[a] = -std::numeric_limits<float>::infinity();
        }
        return b;
    }
    static void grow(Box6& b, const Box6& o) {
        for (int a = 0; a < 3; a++) {
            b.lo[a] = std::min(b.lo[a], o.lo[a]);
            b.hi[a] = std::max(b.hi[a], o.hi[a]);
        }
    }
    static float half_area(const Box6& b) {
        const float dx = b.hi[0] - b.lo[0], dy = b.hi[1] - b.lo[1], dz = b.hi[2] - b.lo[2];
        if (!(dx >= 0 && dy >= 0 && dz >= 0)) return 0.0f;
        return dx * dy + dy * dz + dz * dx;
    }
    Box6 range_box(uint32_t b, uint32_t e) const {
        Box6 r = empty_box();
        for (uint32_t i = b; i < e; i++) grow(r, box[idx[i]]);
        return r;
    }
    static uint64_t capacity(int depth_left) {
        uint64_t c = 1;
        for (int i = 0; i < depth_left; i++) c *= 4u;
        return c;
    }
    // Large ranges: binned SAH (32 bins per axis over the centroid bounds, O(n) per split), then an in-place partition.  A
    // plane is eligible only if both sides fit under the remaining depth; when none is, or the centroids coincide, the range
    // is cut at an admissible position of its longest centroid axis (nth_element).
    static const uint32_t BIN_THRESHOLD = 2048, NBINS = 32;
    uint32_t split_binned(uint32_t b, uint32_t e, uint32_t kmin, uint32_t kmax) {
        const uint32_t n = e - b;
        float cmin[3], cmax[3];
        for (int a = 0; a < 3; a++) {
            cmin[a] = std::numeric_limits<float>::infinity();
            cmax[a] = -cmin[a];
        }
        for (uint32_t i = b; i < e; i++)
            for (int a = 0; a < 3; a++) {
                const float c = cen[a][idx[i]];
                cmin[a] = std::min(cmin[a], c);
                cmax[a] = std::max(cmax[a], c);
            }
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1;
        uint32_t best_bin = 0;
        for (int a = 0; a < 3; a++) {
            const float ext = cmax[a] - cmin[a];
            if (!(ext > 0.0f)) continue;
            const float scale = (float)NBINS / ext;
            Box6 bb[NBINS];
            uint32_t bc[NBINS];
            for (uint32_t k = 0; k < NBINS; k++) {
                bb[k] = empty_box();
                bc[k] = 0;
            }
            for (uint32_t i = b; i < e; i++) {
                const uint32_t it = idx[i];
                uint32_t k = (uint32_t)((cen[a][it] - cmin[a]) * scale);
                if (k >= NBINS) k = NBINS - 1;
                grow(bb[k], box[it]);
                bc[k]++;
            }
            float ra[NBINS];
            Box6 acc = empty_box();
            for (uint32_t k = NBINS; k-- > 1;) {
                grow(acc, bb[k]);
                ra[k] = half_area(acc);
            }
            acc = empty_box();
            uint32_t left = 0;
            for (uint32_t k = 1; k < NBINS; k++) {  // plane between bin k-1 and bin k
                grow(acc, bb[k - 1]);
                left += bc[k - 1];
                if (left < kmin || left > kmax) continue;
                const float cost = half_area(acc) * (float)left + ra[k] * (float)(n - left);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_bin = k;
                }
            }
        }
        if (best_axis >= 0) {
            const float scale = (float)NBINS / (cmax[best_axis] - cmin[best_axis]);
            const float c0 = cmin[best_axis];
            const std::vector<float>& key = cen[best_axis];
            auto mid = std::partition(idx.begin() + b, idx.begin() + e, [&](uint32_t it) {
                uint32_t k = (uint32_t)((key[it] - c0) * scale);
                if (k >= NBINS) k = NBINS - 1;
                return k < best_bin;
            });
            return (uint32_t)(mid - idx.begin());
        }
        int axis = 0;
        for (int a = 1; a < 3; a++)
            if (cmax[a] - cmin[a] > cmax[axis] - cmin[axis]) axis = a;
        const uint32_t k = std::min(std::max(n / 2, kmin), kmax);
        const std::vector<float>& key = cen[axis];
        std::nth_element(idx.begin() + b, idx.begin() + b + k, idx.begin() + e, [&](uint32_t p, uint32_t q) { return key[p] < key[q]; });
        return b + k;
    }

    // SAH split of idx[b, e), full sweep on every axis (small ranges) or binned (large ones); each side may hold at most
    // `cap` items
    uint32_t split(uint32_t b, uint32_t e, uint64_t cap) {
        const uint32_t n = e - b;
        const uint32_t kmin = (uint32_t)std::max<int64_t>(1, (int64_t)n - (int64_t)cap);
        const uint32_t kmax = (uint32_t)std::min<uint64_t>(n - 1, cap);
        if (n > BIN_THRESHOLD) return split_binned(b, e, kmin, kmax);
        float best_cost = std::numeric_limits<float>::infinity();
        int best_axis = -1;
        uint32_t best_k = std::min(std::max(n / 2, kmin), kmax);
        std::vector<uint32_t> ord[3];
        rarea.resize(n);
        for (int a = 0; a < 3; a++) {
            ord[a].assign(idx.begin() + b, idx.begin() + e);
            std::stable_sort(ord[a].begin(), ord[a].end(), [&](uint32_t p, uint32_t q) { return cen[a][p] < cen[a][q]; });
            Box6 acc = empty_box();
            for (uint32_t i = n; i-- > 1;) {
                grow(acc, box[ord[a][i]]);
                rarea[i] = half_area(acc);
            }
            acc = empty_box();
            for (uint32_t k = 1; k < n; k++) {
                grow(acc, box[ord[a][k - 1]]);
                if (k < kmin || k > kmax) continue;
                const float cost = half_area(acc) * (float)k + rarea[k] * (float)(n - k);
                if (cost < best_cost) {
                    best_cost = cost;
                    best_axis = a;
                    best_k = k;
                }
            }
        }
        if (best_axis >= 0) std::copy(ord[best_axis].begin(), ord[best_axis].end(), idx.begin() + b);
        return b + best_k;
    }
    uint32_t child_ref(uint32_t b, uint32_t e, int depth_left, int par_levels) {
        if (e - b == 1) return ref[idx[b]];
        if (tasks && par_levels <= 0 && e - b > TOP_TASK_MIN) {  // a subtree for a worker thread
            tasks->push_back(TopTask{b, e, depth_left, {}, REF_NONE});
            return TOP_TASK | (uint32_t)(tasks->size() - 1);
        }
        return build4(b, e, depth_left, par_levels);
    }
    uint32_t build4(uint32_t b, uint32_t e, int depth_left, int par_levels = 0) {  // more than one item; returns TOP_LOCAL | index in `nodes`
        const uint32_t me = (uint32_t)nodes.size();
        nodes.push_back(SubNode());
        nodes.push_back(SubNode());
        const uint64_t cap1 = capacity(depth_left - 1);
        const uint32_t m = split(b, e, 2 * cap1);
        uint32_t cb[4], ce[4];
        int nc = 0;
        const uint32_t hb[2] = {b, m}, he[2] = {m, e};
        for (int h = 0; h < 2; h++) {
            if (he[h] - hb[h] > 1) {
                const uint32_t q = split(hb[h], he[h], cap1);
                cb[nc] = hb[h], ce[nc++] = q;
                cb[nc] = q, ce[nc++] = he[h];
            } else {
                cb[nc] = hb[h], ce[nc++] = he[h];
            }
        }
        Box6 cbox[4];
        uint32_t cref[4];
        for (int c = 0; c < 4; c++) {
            if (c < nc) {
                cbox[c] = range_box(cb[c], ce[c]);
                cref[c] = child_ref(cb[c], ce[c], depth_left - 1, par_levels - 1);
            } else {
                cbox[c] = empty_box();
                cref[c] = REF_NONE;
            }
        }
        for (int r = 0; r < 2; r++) {
            SubNode& N = nodes[me + r];
            sub_box_store(N.box0, cbox[2 * r]);
            sub_box_store(N.box1, cbox[2 * r + 1]);
            N.ref0 = cref[2 * r];
            N.ref1 = cref[2 * r + 1];
            N.pad[0] = N.pad[1] = 0;
        }
        return TOP_LOCAL | me;
    }
};

// Appends the part `src` (references TOP_LOCAL = own nodes, TOP_TASK = a task's part, anything else absolute) to `dst` in the
// sequential build's pre-order, starting at its node `r`; returns the absolute index of that node.
static uint32_t stitch(std::vector<SubNode>& dst, const std::vector<SubNode>& src, uint32_t r, const std::vector<TopTask>& tasks) {
    const uint32_t at = (uint32_t)dst.size();
    dst.push_back(src[r]);
    dst.push_back(src[r + 1]);
    for (int h = 0; h < 2; h++)
        for (int c = 0; c < 2; c++) {
            const uint32_t cr = c ? src[r + h].ref1 : src[r + h].ref0;
            uint32_t out_ref = cr;
            if (cr != REF_NONE && !(cr & REF_LEAF)) {
                if (cr & TOP_LOCAL) {
                    out_ref = stitch(dst, src, cr & ~TOP_LOCAL, tasks);
                } else if (cr & TOP_TASK) {  // a task's part is already in pre-order: append it whole, shifted
                    const TopTask& T = tasks[cr & ~TOP_TASK];
                    const uint32_t base = (uint32_t)dst.size();
                    for (SubNode N : T.nodes) {
                        for (uint32_t* q : {&N.ref0, &N.ref1})
                            if (*q != REF_NONE && !(*q & REF_LEAF) && (*q & TOP_LOCAL)) *q = (*q & ~TOP_LOCAL) + base;
                        dst.push_back(N);
                    }
                    out_ref = (T.root & ~TOP_LOCAL) + base;
                }
            }
            (c ? dst[at + h].ref1 : dst[at + h].ref0) = out_ref;
        }
    return at;
}

// Called before the references are made global: everything is still in builder-local indices (subnode index, record
// index), which the globalisation loop of build_reference_bvh then shifts like every other accelerator reference.
void build_fast_tree(BuiltBvh& out, bool force, bool leaf_accel, int open, int threads) {
    if (open < 0) open = 0;
    if (open > SUB_MAX_DEPTH) open = SUB_MAX_DEPTH;
    out.fast_root = REF_NONE;
    out.paths.clear();
    out.tri_leaf.clear();
    const size_t nleaves = out.leaves.size();
    if (nleaves == 0 || !out.geometry_finite || out.has_wild) return;
    // by default: not for scenes built without in-leaf accelerators (they are asked to scan leaves linearly) and not for a
    // handful of triangles, where the exact walk's few steps are cheaper than a search plus a certificate
    if (!force && (!leaf_accel || out.tris.size() < 16)) return;
    std::vector<Box6> box(nleaves);
    std::vector<uint32_t> ref(nleaves);
    for (size_t i = 0; i < out.nodes.size(); i++) {
        const TopoNode& n = out.nodes[i];
        if (!n.leaf) continue;
        const uint32_t li = (uint32_t)out.node_to_ref_index[i];
        const LeafRec& L = out.leaves[li];
        box[li] = n.box;
        if (L.sub_root != REF_NONE)
            ref[li] = L.sub_root;
        else if (L.count >= 1 && L.count <= SUB_RUN_MAX && (uint64_t)L.first + L.count <= SUB_MAX_RECORDS)
            ref[li] = REF_LEAF | ((L.count - 1) << 26) | L.first;
        else
            return;  // a big leaf without accelerator (accelerator disabled or beyond its encoding): exact walk only
    }
    // paths: the boxes of the nodes between the root (exclusive) and each leaf (inclusive), top-down
    std::vector<int> parent(out.nodes.size(), -1);
    for (size_t i = 0; i < out.nodes.size(); i++)
        if (!out.nodes[i].leaf) parent[out.nodes[i].child[0]] = parent[out.nodes[i].child[1]] = (int)i;
    {  // unused slots hold all of space: every origin lies strictly inside, so such a box passes the certificate unread
        const float inf = std::numeric_limits<float>::infinity();
        const float all[6] = {-inf, -inf, -inf, inf, inf, inf};
        out.paths.resize(nleaves * PATH_BOXES * 6);
        for (size_t k = 0; k < nleaves * PATH_BOXES; k++) std::memcpy(&out.paths[k * 6], all, 24);
    }
    for (size_t i = 0; i < out.nodes.size(); i++) {
        if (!out.nodes[i].leaf) continue;
        const uint32_t li = (uint32_t)out.node_to_ref_index[i];
        int chain[MAX_LEVELS], n = 0;
        for (int k = (int)i; parent[k] >= 0; k = parent[k]) chain[n++] = k;  // the root (no parent) is not part of the path
        for (int k = 0; k < n; k++) std::memcpy(&out.paths[((size_t)li * PATH_BOXES + k) * 6], &out.nodes[chain[n - 1 - k]].box, 24);
        // Boxes a certificate has to test explicitly (walk_fast.h path_certified): the leaf's own, and every box that does
        // not contain the next one down.  A box that contains a box the ray provably enters at t is entered at t too (the
        // reference's slab arithmetic is monotone in the plane coordinates), so nested ancestors need no test of their own.
        uint32_t need = n ? (1u << (n - 1)) : 0u;
        for (int k = 0; k + 1 < n; k++) {
            const Box6 &a = out.nodes[chain[n - 1 - k]].box, &b = out.nodes[chain[n - 2 - k]].box;  // a: level k, b: its child on the path
            bool nested = true;
            for (int ax = 0; ax < 3; ax++) nested = nested && (a.lo[ax] <= b.lo[ax]) && (a.hi[ax] >= b.hi[ax]);
            if (!nested) need |= 1u << k;
        }
        out.leaves[li].path_len = (uint32_t)n | (need << 8);
    }
    out.tri_leaf.resize(out.tris.size());
    for (uint32_t li = 0; li < nleaves; li++)
        for (uint32_t k = 0; k < out.leaves[li].count; k++) out.tri_leaf[out.leaves[li].first + k] = li;
    // Items of the top tree: the leaves, or -- `open` levels further down -- the children of their accelerator nodes.  An
    // opened leaf is no longer one subtree of the fast tree: the SAH then groups small clusters by position, whatever
    // reference leaf they come from, which removes most of the overlap between the median-split leaves.
    for (int lvl = 0; lvl < open; lvl++) {
        std::vector<Box6> b2;
        std::vector<uint32_t> r2;
        for (size_t i = 0; i < ref.size(); i++) {
            if (ref[i] & REF_LEAF) {
                b2.push_back(box[i]);
                r2.push_back(ref[i]);
                continue;
            }
            for (int h = 0; h < 2; h++) {
                const SubNode& N = out.subnodes[ref[i] + h];
                const uint32_t cr[2] = {N.ref0, N.ref1};
                const float* cb[2] = {N.box0, N.box1};
                for (int c = 0; c < 2; c++) {
                    if (cr[c] == REF_NONE) continue;
                    b2.push_back(sub_box_load(cb[c]));
                    r2.push_back(cr[c]);
                }
            }
        }
        box.swap(b2);
        ref.swap(r2);
    }
    const uint32_t nitems = (uint32_t)ref.size();
    if (nitems == 1) {
        out.fast_root = ref[0];
        return;
    }
    std::vector<uint32_t> idx(nitems);
    std::vector<float> cen[3];
    for (int a = 0; a < 3; a++) cen[a].resize(nitems);
    for (uint32_t i = 0; i < nitems; i++) {
        idx[i] = i;
        for (int a = 0; a < 3; a++) cen[a][i] = 0.5f * box[i].lo[a] + 0.5f * box[i].hi[a];
    }
    unsigned nthreads = threads > 0 ? (unsigned)threads : std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1 || nitems <= 4 * TOP_TASK_MIN) nthreads = 1;
    // the top levels on this thread (their subtrees become tasks), the tasks on workers, then everything in pre-order
    std::vector<SubNode> top;
    std::vector<TopTask> tasks;
    TopBuilder tb{box, ref, top, idx, cen, nthreads > 1 ? &tasks : nullptr, {}};
    const uint32_t root = tb.build4(0, nitems, TOP_MAX_DEPTH + open, TOP_PAR_LEVELS);
    if (!tasks.empty()) {
        std::vector<uint32_t> by_size(tasks.size());
        for (uint32_t i = 0; i < by_size.size(); i++) by_size[i] = i;
        std::sort(by_size.begin(), by_size.end(), [&](uint32_t x, uint32_t y) { return tasks[x].e - tasks[x].b > tasks[y].e - tasks[y].b; });
        std::atomic<uint32_t> next{0};
        auto work = [&]() {
            for (;;) {
                const uint32_t k = next.fetch_add(1);
                if (k >= by_size.size()) return;
                TopTask& T = tasks[by_size[k]];
                TopBuilder w{box, ref, T.nodes, idx, cen, nullptr, {}};
                T.root = w.build4(T.b, T.e, T.depth_left);
            }
        };
        std::vector<std::thread> pool;
        const unsigned nt = (unsigned)std::min<size_t>(nthreads, tasks.size());
        for (unsigned t = 1; t < nt; t++) pool.emplace_back(work);
        work();
        for (std::thread& th : pool) th.join();
    }
    out.fast_root = stitch(out.subnodes, top, root & ~TOP_LOCAL, tasks);
}

}  // namespace

bool build_reference_bvh(const HostScene& sc, const BuildOptions& opt, BuiltBvh& out, std::string& err) {
    auto t_start = std::chrono::steady_clock::now();
    out = BuiltBvh();
    // ---- validation (the reference has none; out-of-range input would be UB there) ----
    if (sc.pos_nrm.size() != 6 * (size_t)sc.nverts || sc.tri.size() != 3 * (size_t)sc.ntris ||
        sc.tri_mesh.size() != sc.ntris || sc.materials.size() != 8 * (size_t)sc.nmesh ||
        sc.spheres.size() != 5 * (size_t)sc.nspheres) {
        err = "scene arrays do not match their counts";
        return false;
    }
    for (uint32_t p = 0; p < sc.ntris; p++) {
        if (sc.tri[3 * p] >= sc.nverts || sc.tri[3 * p + 1] >= sc.nverts || sc.tri[3 * p + 2] >= sc.nverts) {
            err = "triangle " + std::to_string(p) + " references a vertex out of range";
            return false;
        }
        if (sc.tri_mesh[p] >= sc.nmesh || (p > 0 && sc.tri_mesh[p] < sc.tri_mesh[p - 1])) {
            err = "tri_mesh must be non-decreasing and < nmesh";
            return false;
        }
    }
    for (uint32_t s = 0; s < sc.nspheres; s++)
        out.spheres.push_back(SphereRecord{{sc.spheres[5 * s], sc.spheres[5 * s + 1], sc.spheres[5 * s + 2]}, sc.spheres[5 * s + 3]});

    if (sc.ntris == 0) {  // bvh.cpp:52-55: no meshes -> no tree; intersect() then only tests spheres (:870)
        out.levels = 1;  // numLevels() of an empty node list returns 0 + 1 (:214-224)
        out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
        return true;
    }

    out.order.resize(sc.ntris);
    for (uint32_t p = 0; p < sc.ntris; p++) out.order[p] = p;
    Builder B(sc, out.order);
    B.centroids();

    // runs[i] is only meaningful while node i is waiting to be split
    std::vector<std::vector<Run>> runs;
    {
        std::vector<Run> r;
        uint32_t p = 0;
        while (p < sc.ntris) {
            uint32_t q = p;
            while (q < sc.ntris && sc.tri_mesh[q] == sc.tri_mesh[p]) q++;
            r.push_back(Run{(int)sc.tri_mesh[p], p, q - p});
            p = q;
        }
        bool leaf = (MAX_LEVELS - 1 == 0) || (r.size() == 1 && r[0].count == 1);  // bvh.cpp:59
        out.nodes.push_back(TopoNode{leaf, 0, B.bounds(0, sc.ntris), {-1, -1}, 0, sc.ntris});
        runs.push_back(std::move(r));
    }

    // createTree, bvh.cpp:343-372, level by level: the nodes of a level own disjoint ranges of `order`, so they are split on
    // worker threads (every std::sort call is the sequential build's, on the same sequence -- the leaf order depends on
    // it); their children are appended afterwards in node order, which is the numbering of the sequential build.
    struct Split {
        std::vector<Run> lr, rr;
        uint32_t lcount = 0;
        TopoNode left, right;
    };
    auto split_node = [&](size_t i, std::vector<std::pair<float, uint32_t>>& tmp, std::vector<uint32_t>& scratch, Split& R) {
        const TopoNode nd = out.nodes[i];
        std::vector<Run> my = std::move(runs[i]);
        const float ex = nd.box.hi[0] - nd.box.lo[0], ey = nd.box.hi[1] - nd.box.lo[1], ez = nd.box.hi[2] - nd.box.lo[2];
        const int axis = (ex > ey) ? ((ex > ez) ? 0 : 2) : ((ey > ez) ? 1 : 2);  // :289

        std::vector<Run>&lr = R.lr, &rr = R.rr;
        uint32_t lcount;
        if (my.size() > 1) {
            // bvh.cpp:168-179 / :88-110: order the MESHES by the centroid coordinate of the median
            // triangle of each mesh's sorted triangle list, then split the mesh list in half.
            struct Keyed {
                float key;
                Run run;
            };
            std::vector<Keyed> ks;
            for (const Run& r : my) {
                scratch.assign(out.order.begin() + r.first, out.order.begin() + r.first + r.count);
                std::vector<std::pair<float, uint32_t>> kv(r.count);
                for (uint32_t k = 0; k < r.count; k++) kv[k] = std::make_pair(B.cen[axis][scratch[k]], scratch[k]);
                std::sort(kv.begin(), kv.end(), [](const std::pair<float, uint32_t>& a, const std::pair<float, uint32_t>& b) {
                    return a.first < b.first;
                });
                ks.push_back(Keyed{kv[r.count / 2].first, r});
            }
            std::sort(ks.begin(), ks.end(), [](const Keyed& a, const Keyed& b) { return a.key < b.key; });
            // rewrite this node's range of `order` in the new mesh order (triangle order inside a mesh is kept)
            scratch.assign(out.order.begin() + nd.first, out.order.begin() + nd.first + nd.count);
            uint32_t w = nd.first;
            const size_t half = ks.size() / 2;
            for (size_t k = 0; k < ks.size(); k++) {
                const Run& r = ks[k].run;
                std::memcpy(&out.order[w], &scratch[r.first - nd.first], sizeof(uint32_t) * r.count);
                (k < half ? lr : rr).push_back(Run{r.mesh, w, r.count});
                w += r.count;
            }
            lcount = 0;
            for (const Run& r : lr) lcount += r.count;
        } else {
            // bvh.cpp:192-207: one mesh -> sort its triangles along the axis, left = [0, n/2)
            B.sort_range(nd.first, nd.count, axis, tmp);
            lcount = nd.count / 2;
            lr.push_back(Run{my[0].mesh, nd.first, lcount});
            rr.push_back(Run{my[0].mesh, nd.first + lcount, nd.count - lcount});
        }
        const bool lvl = (nd.level + 1 == MAX_LEVELS - 1);  // :320
        const bool ll = lvl || (lr.size() == 1 && lr[0].count == 1);
        const bool rl = lvl || (rr.size() == 1 && rr[0].count == 1);
        R.lcount = lcount;
        R.left = TopoNode{ll, nd.level + 1, B.bounds(nd.first, lcount), {-1, -1}, nd.first, lcount};
        R.right = TopoNode{rl, nd.level + 1, B.bounds(nd.first + lcount, nd.count - lcount), {-1, -1}, nd.first + lcount, nd.count - lcount};
    };
    unsigned nthreads = opt.threads > 0 ? (unsigned)opt.threads : std::thread::hardware_concurrency();
    if (nthreads > 16) nthreads = 16;
    if (nthreads < 1) nthreads = 1;
    for (size_t lb = 0, le = out.nodes.size(); lb < le; lb = le, le = out.nodes.size()) {  // [lb, le) = the nodes of one level
        std::vector<size_t> todo;
        for (size_t i = lb; i < le; i++)
            if (!out.nodes[i].leaf) todo.push_back(i);
        std::vector<Split> res(todo.size());
        const unsigned nt = (unsigned)std::min<size_t>(nthreads, todo.size());
        auto work = [&](unsigned t) {
            std::vector<std::pair<float, uint32_t>> tmp;
            std::vector<uint32_t> scratch;
            for (size_t k = t; k < todo.size(); k += nt) split_node(todo[k], tmp, scratch, res[k]);
        };
        if (nt <= 1) {
            if (nt == 1) work(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, t);
            for (std::thread& th : pool) th.join();
        }
        for (size_t k = 0; k < todo.size(); k++) {
            const size_t i = todo[k];
            const int li = (int)out.nodes.size();
            out.nodes[i].child[0] = li;
            out.nodes[i].child[1] = li + 1;
            out.nodes.push_back(res[k].left);
            out.nodes.push_back(res[k].right);
            runs.push_back(std::move(res[k].lr));
            runs.push_back(std::move(res[k].rr));
        }
    }

    int maxLevel = 0;
    for (const TopoNode& n : out.nodes) maxLevel = std::max(maxLevel, n.level);
    out.levels = maxLevel + 1;
    if (out.levels > MAX_LEVELS) {
        err = "BVH deeper than 12 levels";
        return false;
    }

    // ---- flatten ----
    out.node_to_ref_index.assign(out.nodes.size(), -1);
    uint32_t npk = 0, nlf = 0;
    for (size_t i = 0; i < out.nodes.size(); i++) out.node_to_ref_index[i] = out.nodes[i].leaf ? (int)nlf++ : (int)npk++;
    out.packets.resize(npk);
    out.leaves.resize(nlf);
    out.tris.resize(sc.ntris);
    out.tri_normals.resize(sc.ntris);
    out.leaf_wild.assign(nlf, 0);
    auto ref_of = [&](int node) -> uint32_t {
        return out.nodes[node].leaf ? (REF_LEAF | (uint32_t)out.node_to_ref_index[node]) : (uint32_t)out.node_to_ref_index[node];
    };
    // leaf records on worker threads: every leaf owns a disjoint range of the record arrays, whose start is the prefix sum of the
    // counts in node order (the sequential numbering); inner nodes are few and stay on this thread
    std::vector<size_t> leaf_nodes;
    {
        uint32_t w = 0;
        for (size_t i = 0; i < out.nodes.size(); i++) {
            const TopoNode& n = out.nodes[i];
            if (!n.leaf) continue;
            LeafRec& L = out.leaves[out.node_to_ref_index[i]];
            L.first = w;
            L.count = n.count;
            L.sub_root = REF_NONE;
            L.path_len = 0;
            w += n.count;
            leaf_nodes.push_back(i);
        }
    }
    auto fill_leaf = [&](size_t i) -> bool {
        const TopoNode& n = out.nodes[i];
        const LeafRec& L = out.leaves[out.node_to_ref_index[i]];
        bool wild = false;
        uint32_t w = L.first;
        for (uint32_t k = 0; k < n.count; k++, w++) {
            const uint32_t p = out.order[n.first + k];
            const float* a = B.vpos(sc.tri[3 * (size_t)p]);
            const float* b = B.vpos(sc.tri[3 * (size_t)p + 1]);
            const float* c = B.vpos(sc.tri[3 * (size_t)p + 2]);
            TriRecord& T = out.tris[w];
            std::memcpy(T.v0, a, 12);
            std::memcpy(T.v1, b, 12);
            std::memcpy(T.v2, c, 12);
            F3 pn;
            float D;
            triangle_plane(f3(a[0], a[1], a[2]), f3(b[0], b[1], b[2]), f3(c[0], c[1], c[2]), pn, D);
            T.n[0] = pn.x;
            T.n[1] = pn.y;
            T.n[2] = pn.z;
            T.D = D;
            T.prim_id = p;
            T.mesh_id = sc.tri_mesh[p];
            T.scan_k = k;
            if (!plane_is_tame(a, b, c, pn)) wild = true;
            TriNormals& N = out.tri_normals[w];
            std::memcpy(N.n1, a + 3, 12);
            std::memcpy(N.n2, b + 3, 12);
            std::memcpy(N.n3, c + 3, 12);
        }
        return wild;
    };
    {
        const unsigned nt = (sc.ntris < 65536 || leaf_nodes.size() < 2) ? 1u : (unsigned)std::min<size_t>(opt.threads > 0 ? (unsigned)opt.threads : nthreads, leaf_nodes.size());
        auto work = [&](unsigned t) {
            for (size_t k = t; k < leaf_nodes.size(); k += nt)
                if (fill_leaf(leaf_nodes[k])) out.leaf_wild[out.node_to_ref_index[leaf_nodes[k]]] = 1;  // (one byte per leaf, one writer per leaf)
        };
        if (nt <= 1) {
            work(0);
        } else {
            std::vector<std::thread> pool;
            for (unsigned t = 0; t < nt; t++) pool.emplace_back(work, t);
            for (std::thread& th : pool) th.join();
        }
        for (uint8_t wl : out.leaf_wild) out.has_wild = out.has_wild || wl;
    }
    for (size_t i = 0; i < out.nodes.size(); i++) {
        const TopoNode& n = out.nodes[i];
        if (!n.leaf) {
            NodePacket& P = out.packets[out.node_to_ref_index[i]];
            const TopoNode &l = out.nodes[n.child[0]], &r = out.nodes[n.child[1]];
            std::memcpy(P.lbox, &l.box, 24);
            std::memcpy(P.rbox, &r.box, 24);
            P.left = ref_of(n.child[0]);
            P.right = ref_of(n.child[1]);
            P.pad[0] = P.pad[1] = 0;
        }
    }
    for (size_t v = 0; v < sc.pos_nrm.size(); v += 6)
        for (int a = 0; a < 3; a++) {
            const float c = std::fabs(sc.pos_nrm[v + a]);
            if (c > out.scene_absmax) out.scene_absmax = c;  // NaN coordinates never raise it
            if (!(c <= std::numeric_limits<float>::max())) out.geometry_finite = false;
        }
    const auto t_a = std::chrono::steady_clock::now();
    if (opt.leaf_accel) build_leaf_accelerators(out, opt.sub_leaf_tris, opt.threads);
    const auto t_b = std::chrono::steady_clock::now();
    if (opt.fast_tree != 0) build_fast_tree(out, opt.fast_tree > 0, opt.leaf_accel, opt.fast_open, opt.threads);
    if (getenv("CGRT_BUILD_TIMES"))
        fprintf(stderr, "build: reference tree + records %.3f s, leaf accelerators %.3f s, fast tree %.3f s\n",
                std::chrono::duration<double>(t_a - t_start).count(), std::chrono::duration<double>(t_b - t_a).count(),
                std::chrono::duration<double>(std::chrono::steady_clock::now() - t_b).count());
    // one 64-byte record array on the device: [packets | subnodes | tris]; make sub/tri references global
    if (out.packets.size() & 1u) {
        NodePacket pad;  // keeps every 128-byte accelerator node inside one 128-byte line (the array is 256-byte aligned)
        std::memset(&pad, 0, sizeof(pad));
        out.packets.push_back(pad);
    }
    out.sub_base = (uint32_t)out.packets.size();
    out.tri_base = out.sub_base + (uint32_t)out.subnodes.size();
    if ((uint64_t)out.tri_base + out.tris.size() > SUB_MAX_RECORDS) {
        // run references carry 26 index bits: beyond that the accelerator is dropped (linear leaves)
        out.subnodes.clear();
        for (LeafRec& L : out.leaves) L.sub_root = REF_NONE;
        out.tri_base = out.sub_base;
        out.fast_root = REF_NONE;
    }
    for (LeafRec& L : out.leaves) {
        L.first += out.tri_base;
        if (L.sub_root != REF_NONE) L.sub_root += out.sub_base;
    }
    for (SubNode& N : out.subnodes)
        for (uint32_t* r : {&N.ref0, &N.ref1})
            if (*r != REF_NONE) *r += (*r & REF_LEAF) ? out.tri_base : out.sub_base;
    if (out.fast_root != REF_NONE) out.fast_root += (out.fast_root & REF_LEAF) ? out.tri_base : out.sub_base;
    // the device reads all four child references of a node with ONE 16-byte load: the first half's last quarter
    for (size_t i = 0; i + 1 < out.subnodes.size(); i += 2) {
        out.subnodes[i].pad[0] = out.subnodes[i + 1].ref0;
        out.subnodes[i].pad[1] = out.subnodes[i + 1].ref1;
    }
    out.root_box = out.nodes[0].box;
    out.root_ref = ref_of(0);
    if (!out.subnodes.empty()) {
        // accelerated leaves are referenced by their accelerator root (cgrt_layout.h REF_LEAF_ACCEL)
        auto direct = [&](uint32_t r) -> uint32_t {
            if (r == REF_NONE || !(r & REF_LEAF)) return r;
            const uint32_t li = r & ~REF_LEAF;
            const uint32_t root = out.leaves[li].sub_root;
            if (root == REF_NONE) return r;
            out.subnodes[root - out.sub_base + 1].pad[0] = li;  // (second half: the first half's pad words carry references)
            return REF_LEAF | REF_LEAF_ACCEL | root;
        };
        for (NodePacket& P : out.packets) {
            P.left = direct(P.left);
            P.right = direct(P.right);
        }
        out.root_ref = direct(out.root_ref);
    }
    out.build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_start).count();
    return true;
}

}  // namespace cgrt
