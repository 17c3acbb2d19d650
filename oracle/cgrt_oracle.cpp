// cgrt_oracle.cpp -- CPU restatement of the reference's primary-ray hot path.
//
// TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product: only
// tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this
// library, and there only as the checker / the reported CPU baseline.  The
// product (cg-raytracer_amd/csrc) never links, imports or executes it.
//
// PARITY UNPINNED (see DESIGN.md "Oracle"): the reference ships no tests, golden
// vectors or fixtures, and its hot path cannot be compiled in
// this image without writing a stand-in for glm 0.9.9.8 (absent; pinned at
// framework/cmake/download_framework_packages.cmake:19-22), which the build
// rules forbid.  What *is* pinned: BVH level counts published in report.pdf
// Table 2 (Cornell 8, Monkey 11) and the three cube.obj "false-miss" rays the
// survey probe recorded from the reference itself; both are
// checked in tests/test_oracle_pins.py.
//
// Every function cites the reference file:line whose behaviour it restates
// (paths relative to /root/reference).  Arithmetic is IEEE binary32 with the
// reference's expression order; build with -ffp-contract=off and no fast-math
// (oracle/Makefile) so that x86-64 SSE scalar code reproduces the reference's
// -O0 build bit for bit.  glm 0.9.9.8 scalar formulas are restated from its
// published func_geometric.inl / type_quat.inl (not in the image).
//
// Layout differs from the reference on purpose (index lists instead of per-node
// deep copies of every vertex, bvh.cpp:205-206) so that 800 K-triangle scenes
// build in seconds; the algorithm, visit order and arithmetic do not.

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <string>
#include <vector>
#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

// ---------------------------------------------------------------------------
// glm 0.9.9.8 scalar vec3 semantics (restated)
// ---------------------------------------------------------------------------
struct V3 {
    float x, y, z;
};
inline V3 mk(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return mk(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return mk(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return mk(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return mk(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return mk(s * a.x, s * a.y, s * a.z); }
inline V3 operator*(V3 a, V3 b) { return mk(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return mk(a.x / s, a.y / s, a.z / s); }  // true division per component
inline V3 operator/(V3 a, V3 b) { return mk(a.x / b.x, a.y / b.y, a.z / b.z); }
// glm::dot(vec3): tmp = a*b; return tmp.x + tmp.y + tmp.z  (left to right)
inline float dot3(V3 a, V3 b) {
    V3 t = a * b;
    return t.x + t.y + t.z;
}
// glm::cross
inline V3 cross3(V3 a, V3 b) {
    return mk(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
// glm::inversesqrt(float) = 1.0f / sqrt(x); glm::normalize(v) = v * inversesqrt(dot(v, v))
inline V3 normalize3(V3 v) { return v * (1.0f / std::sqrt(dot3(v, v))); }
inline float length3(V3 v) { return std::sqrt(dot3(v, v)); }
// glm::reflect(I, N) = I - N * dot(N, I) * 2
inline V3 reflect3(V3 I, V3 N) { return I - N * dot3(N, I) * 2.0f; }

struct Ray {  // framework/include/ray.h:9-13
    V3 o, d;
    float t;
};
struct Box {  // src/scene.h:31-34
    V3 lo, hi;
};

// ---------------------------------------------------------------------------
// Primitive intersectors (src/ray_tracing.cpp)
// ---------------------------------------------------------------------------

// ray_tracing.cpp:13-16 `magnitude`: sqrt(pow(x,2)+pow(y,2)+pow(z,2)) evaluated
// in double (float/int arguments promote to ::pow(double,double)), narrowed to
// float on return.  pow(x, 2.0) of a float-valued double is exact, so x*x is
// the same value.
inline float magnitude_ref(V3 a) {
    double s = std::pow((double)a.x, 2.0) + std::pow((double)a.y, 2.0) + std::pow((double)a.z, 2.0);
    return (float)std::sqrt(s);
}
// ray_tracing.cpp:17-21 `area`
inline float area_ref(V3 v0, V3 v1, V3 v2) { return magnitude_ref(cross3(v1 - v0, v2 - v0)) / 2.0f; }

// ray_tracing.cpp:23-38 pointInTriangle
inline bool point_in_triangle(V3 v0, V3 v1, V3 v2, V3 n, V3 p) {
    V3 e01 = v1 - v0, e12 = v2 - v1, e20 = v0 - v2;
    V3 q0 = p - v0, q1 = p - v1, q2 = p - v2;
    return dot3(n, cross3(e01, q0)) >= 0 && dot3(n, cross3(e12, q1)) >= 0 && dot3(n, cross3(e20, q2)) >= 0;
}

// ray_tracing.cpp:40-72 intersectRayWithPlane (plane = {D, normal}, scene.h:26-29)
inline bool ray_plane(float D, V3 n, Ray& r) {
    if (dot3(r.o, n) == D) {  // origin on the plane: accepted with t = 0 and NO t < ray.t guard (:43-47)
        r.t = 0;
        return true;
    }
    float den = dot3(r.d, n);
    if (den == 0) return false;  // :50-54
    float num = D - dot3(r.o, n);
    float t = num / den;           // :57-58
    if (t < 0) return false;       // :59 (-0.0f passes)
    if (t >= r.t) return false;    // :65
    r.t = t;
    return true;
}

// ray_tracing.cpp:74-82 trianglePlane
inline void triangle_plane(V3 v0, V3 v1, V3 v2, V3& n, float& D) {
    V3 u = v1 - v0, v = v2 - v0;
    n = normalize3(cross3(u, v));
    D = dot3(v0, n);
}

// ray_tracing.cpp:86-114 intersectRayWithTriangle.  `nrm` receives hitInfo.normal on accept.
inline bool ray_triangle(V3 v0, V3 v1, V3 v2, Ray& r, V3& nrm, V3 n1, V3 n2, V3 n3) {
    V3 pn;
    float D;
    triangle_plane(v0, v1, v2, pn, D);
    float prevT = r.t;
    if (ray_plane(D, pn, r)) {
        if (point_in_triangle(v0, v1, v2, pn, r.o + r.d * r.t)) {
            float alpha = area_ref(r.o + r.d * r.t, v1, v2) / area_ref(v0, v1, v2);
            float beta = area_ref(r.o + r.d * r.t, v0, v2) / area_ref(v0, v1, v2);
            float gamma = area_ref(r.o + r.d * r.t, v0, v1) / area_ref(v0, v1, v2);
            V3 ni = normalize3(alpha * n1 + beta * n2 + gamma * n3);
            nrm = (dot3(pn, -r.d) > 0) ? ni : -ni;  // :99-106
            return true;
        }
        r.t = prevT;  // :110
    }
    return false;
}

// ray_tracing.cpp:118-158 intersectRayWithShape(Sphere)
inline bool ray_sphere(V3 c, float radius, Ray& r, V3& nrm) {
    V3 co = r.o - c;
    float a = dot3(r.d, r.d);
    float b = 2 * dot3(r.d, co);
    float cc = dot3(co, co) - radius * radius;
    float disc = b * b - 4 * a * cc;
    if (disc < 0) return false;
    // unqualified sqrt(float) under <cmath>: the float overload is not found by
    // ADL for a fundamental type, so ::sqrt(double) is called and the quotient is
    // evaluated in double, then narrowed (ray_tracing.cpp:133-134).
    float smallerT = (float)((-b - std::sqrt((double)disc)) / (2 * a));
    float biggerT = (float)((-b + std::sqrt((double)disc)) / (2 * a));
    float cur;
    if (smallerT >= 0)
        cur = smallerT;
    else if (biggerT >= 0)
        cur = biggerT;
    else
        return false;
    if (cur >= r.t) return false;
    r.t = cur;
    nrm = normalize3(r.o + r.d * r.t - c);
    return true;
}

// ray_tracing.cpp:162-200 intersectRayWithShape(AxisAlignedBox): true divisions,
// NaN-order-sensitive ternaries, writes ray.t on success.
inline bool ray_box(const Box& b, Ray& r) {
    V3 tMin = (b.lo - r.o) / r.d;
    V3 tMax = (b.hi - r.o) / r.d;
    float tInX = tMin.x < tMax.x ? tMin.x : tMax.x;
    float tOutX = tMin.x > tMax.x ? tMin.x : tMax.x;
    float tInY = tMin.y < tMax.y ? tMin.y : tMax.y;
    float tOutY = tMin.y > tMax.y ? tMin.y : tMax.y;
    float tInZ = tMin.z < tMax.z ? tMin.z : tMax.z;
    float tOutZ = tMin.z > tMax.z ? tMin.z : tMax.z;
    float tIn = tInX > tInY ? (tInX > tInZ ? tInX : tInZ) : (tInY > tInZ ? tInY : tInZ);
    float tOut = tOutX < tOutY ? (tOutX < tOutZ ? tOutX : tOutZ) : (tOutY < tOutZ ? tOutY : tOutZ);
    float cur;
    if (tIn > tOut || tOut < 0)
        return false;
    else if (tIn < 0)
        cur = tOut;
    else
        cur = tIn;
    if (cur >= r.t) return false;
    r.t = cur;
    return true;
}

// bounding_volume_hierarchy.cpp:647-661 startsInBox (strict: a point on a face is outside)
inline bool starts_in_box(const Ray& r, const Box& b) {
    bool inX = b.lo.x < r.o.x && r.o.x < b.hi.x;
    bool inY = b.lo.y < r.o.y && r.o.y < b.hi.y;
    bool inZ = b.lo.z < r.o.z && r.o.z < b.hi.z;
    return inX && inY && inZ;
}

// ---------------------------------------------------------------------------
// Scene + BVH (src/mesh.h:12-35, src/scene.h:36-60, bounding_volume_hierarchy.h:6-13)
// ---------------------------------------------------------------------------
struct Vert {
    V3 p, n;
};
struct Tri {
    uint32_t a, b, c;
};
struct Mat {
    float v[8];  // kd(3) ks(3) shininess transparency  (mesh.h:17-23)
};
struct SphereRec {
    V3 c;
    float radius;
    int material;  // index into mats, or -1
};

// A node-local view of one reference `Mesh`: which scene mesh it came from and
// the (ordered) list of that mesh's triangles held by the node.  `tris` holds
// GLOBAL primitive ids (prefix over meshes in load order + index in mesh.triangles).
struct MeshRef {
    int mesh;
    std::vector<uint32_t> tris;
};
struct ONode {
    bool leaf;
    int level;
    Box box;
    int child[2];
    std::vector<MeshRef> meshes;
    size_t ntris = 0;  // triangles held (kept after an inner node's lists are dropped)
};

struct HitState {
    V3 normal;
    int material;    // last accepted triangle's mesh material (hitInfo.material), -1 = untouched
    uint32_t prim;   // last accepted primitive (triangle global id, or T + sphere index), 0xffffffff = none
};

struct Counters {
    uint64_t inner_visits = 0, leaf_visits = 0, tri_tests = 0, box_tests = 0;
};

struct Oracle {
    std::vector<Vert> verts;      // global vertex array
    std::vector<Tri> tris;        // global vertex indices, in prim-id order
    std::vector<int> tri_mesh;    // mesh of every triangle (non-decreasing)
    std::vector<uint32_t> mesh_first;  // first prim id of every mesh, size nmesh+1
    std::vector<Mat> mats;        // one per mesh
    std::vector<SphereRec> spheres;
    std::vector<ONode> nodes;
    int maxDepth = 12;  // bvh.cpp:48
    double build_seconds = 0;

    // ---- build (bvh.cpp:42-76, :235-372) ----
    float centroid_coord(uint32_t prim, int axis) const {  // bvh.cpp:126-130
        const Tri& t = tris[prim];
        V3 c = (verts[t.a].p + verts[t.b].p + verts[t.c].p) / 3.0f;
        return axis == 0 ? c.x : (axis == 1 ? c.y : (axis == 2 ? c.z : -1));
    }
    void sort_tris(std::vector<uint32_t>& ids, int axis) const {  // bvh.cpp:122-134 (std::sort, unstable)
        std::sort(ids.begin(), ids.end(),
                  [this, axis](uint32_t p, uint32_t q) { return centroid_coord(p, axis) < centroid_coord(q, axis); });
    }
    Box bounds(const std::vector<MeshRef>& ms) const {  // bvh.cpp:235-268
        // :238 keeps the first index in a float before using it as an index
        float firstIdx = (float)tris[ms[0].tris[0]].a;
        const V3& f = verts[(size_t)firstIdx].p;
        float mnx = f.x, mxx = f.x, mny = f.y, mxy = f.y, mnz = f.z, mxz = f.z;
        for (const MeshRef& m : ms)
            for (uint32_t prim : m.tris) {
                const Tri& t = tris[prim];
                const uint32_t idx[3] = {t.a, t.b, t.c};
                for (int i = 0; i < 3; i++) {
                    const V3& p = verts[idx[i]].p;
                    mnx = (p.x < mnx) ? p.x : mnx;
                    mny = (p.y < mny) ? p.y : mny;
                    mnz = (p.z < mnz) ? p.z : mnz;
                    mxx = (p.x > mxx) ? p.x : mxx;
                    mxy = (p.y > mxy) ? p.y : mxy;
                    mxz = (p.z > mxz) ? p.z : mxz;
                }
            }
        return Box{mk(mnx, mny, mnz), mk(mxx, mxy, mxz)};
    }
    void split(const ONode& nd, ONode& L, ONode& R) const {  // bvh.cpp:280-331
        float ex = nd.box.hi.x - nd.box.lo.x, ey = nd.box.hi.y - nd.box.lo.y, ez = nd.box.hi.z - nd.box.lo.z;
        int axis = (ex > ey) ? ((ex > ez) ? 0 : 2) : ((ey > ez) ? 1 : 2);  // :289
        std::vector<MeshRef> lc, rc;
        if (nd.meshes.size() > 1) {
            // bvh.cpp:168-179 + :88-110: sort the MESHES by the centroid coordinate of
            // the median triangle of each mesh's (sorted copy of its) triangle list.
            std::vector<MeshRef> ms = nd.meshes;
            std::sort(ms.begin(), ms.end(), [this, axis](MeshRef& m1, MeshRef& m2) {
                std::vector<uint32_t> t1 = m1.tris;
                sort_tris(t1, axis);
                std::vector<uint32_t> t2 = m2.tris;
                sort_tris(t2, axis);
                return centroid_coord(t1[t1.size() / 2], axis) < centroid_coord(t2[t2.size() / 2], axis);
            });
            lc.assign(ms.begin(), ms.begin() + ms.size() / 2);
            rc.assign(ms.begin() + ms.size() / 2, ms.end());
        } else {
            // bvh.cpp:192-207: one mesh -> sort its triangles, split at size()/2
            std::vector<uint32_t> ids = nd.meshes[0].tris;
            sort_tris(ids, axis);
            MeshRef l{nd.meshes[0].mesh, std::vector<uint32_t>(ids.begin(), ids.begin() + ids.size() / 2)};
            MeshRef r{nd.meshes[0].mesh, std::vector<uint32_t>(ids.begin() + ids.size() / 2, ids.end())};
            lc.push_back(std::move(l));
            rc.push_back(std::move(r));
        }
        Box bl = bounds(lc), br = bounds(rc);
        bool lvl = (nd.level + 1 == maxDepth - 1);  // :320
        bool ll = lvl || (lc.size() == 1 && lc[0].tris.size() == 1);
        bool rl = lvl || (rc.size() == 1 && rc[0].tris.size() == 1);
        L = ONode{ll, nd.level + 1, bl, {-1, -1}, std::move(lc)};
        R = ONode{rl, nd.level + 1, br, {-1, -1}, std::move(rc)};
    }
    void build() {  // ctor bvh.cpp:42-76 + createTree :343-372 (breadth-first by index)
        auto t0 = std::chrono::steady_clock::now();
        nodes.clear();
        if (mats.empty() || tris.empty()) return;  // :52-55
        std::vector<MeshRef> all;
        for (size_t m = 0; m + 1 < mesh_first.size(); m++) {
            MeshRef r;
            r.mesh = (int)m;
            for (uint32_t p = mesh_first[m]; p < mesh_first[m + 1]; p++) r.tris.push_back(p);
            if (r.tris.empty()) continue;  // a mesh without triangles is UB upstream (:98 indexes an empty list); skipped
            all.push_back(std::move(r));
        }
        Box rb = bounds(all);
        bool leaf = (maxDepth - 1 == 0) || (all.size() == 1 && all[0].tris.size() == 1);  // :59
        nodes.push_back(ONode{leaf, 0, rb, {-1, -1}, std::move(all)});
        for (size_t i = 0; i < nodes.size(); i++) {
            for (const MeshRef& m : nodes[i].meshes) nodes[i].ntris += m.tris.size();
            if (nodes[i].leaf) continue;
            ONode L, R;
            split(nodes[i], L, R);
            nodes[i].child[0] = (int)nodes.size();
            nodes[i].child[1] = (int)nodes.size() + 1;
            // an inner node's triangle lists are never read again: drop them (memory only)
            std::vector<MeshRef>().swap(nodes[i].meshes);
            nodes.push_back(std::move(L));
            nodes.push_back(std::move(R));
        }
        build_seconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    }
    int num_levels() const {  // bvh.cpp:214-224
        int mx = 0;
        for (const ONode& n : nodes)
            if (n.level > mx) mx = n.level;
        return mx + 1;
    }

    // ---- traversal (bvh.cpp:535-881) ----
    bool scan_leaf(const ONode& nd, Ray& r, HitState& h, Counters* c) const {  // bvh.cpp:535-553
        bool hit = false;
        if (c) c->leaf_visits++;
        for (const MeshRef& m : nd.meshes)
            for (uint32_t prim : m.tris) {
                const Tri& t = tris[prim];
                const Vert &a = verts[t.a], &b = verts[t.b], &cc = verts[t.c];
                if (c) c->tri_tests++;
                if (ray_triangle(a.p, b.p, cc.p, r, h.normal, a.n, b.n, cc.n)) {
                    h.material = m.mesh;
                    h.prim = prim;
                    hit = true;
                }
            }
        return hit;
    }
    // bvh.cpp:572-595 intersectChildrenHierarchically
    bool ordered(int first, int second, float tSecond, Ray& r, HitState& h, Counters* c) const {
        bool hitFirst = visit(first, r, h, c);
        if (tSecond < 0) return hitFirst;
        if (hitFirst) {
            if (r.t < tSecond) return true;
            return hitFirst | visit(second, r, h, c);
        }
        return visit(second, r, h, c);
    }
    // bvh.cpp:748-758 intersectRecursive -> :715-736 intersectNonLeaf -> :679-701 intersectDeeper
    // -> :611-635 intersectRayThatStartsOutsideBoxes
    bool visit(int ni, Ray& r, HitState& h, Counters* c) const {
        const ONode& nd = nodes[ni];
        if (nd.leaf) return scan_leaf(nd, r, h, c);
        if (c) {
            c->inner_visits++;
            c->box_tests += 2;
        }
        const float t0 = r.t;
        const int li = nd.child[0], ri = nd.child[1];
        const Box &lb = nodes[li].box, &rb = nodes[ri].box;
        float tL = -1.0f, tR = -1.0f;
        if (ray_box(lb, r)) {
            tL = r.t;
            r.t = t0;
        }
        if (ray_box(rb, r)) {
            tR = r.t;
            r.t = t0;
        }
        bool inL = starts_in_box(r, lb), inR = starts_in_box(r, rb);
        if (inL && inR) {
            // :687 `recurse(left) | recurse(right)`: g++ evaluates the left operand
            // first (checked with this image's g++ 11.4 at -O0 and -O2).
            bool a = visit(li, r, h, c);
            bool b = visit(ri, r, h, c);
            return a | b;
        }
        if (inL) return ordered(li, ri, tR, r, h, c);
        if (inR) return ordered(ri, li, tL, r, h, c);
        if (tL < 0 && tR < 0) return false;
        if (tL < 0) return visit(ri, r, h, c);
        if (tR < 0) return visit(li, r, h, c);
        // :626-634 the reference forgets to `return` here.  At -O0 (its
        // only working build) the callee's bool is still in %al, i.e. the value IS
        // returned; that behaviour is what is restated.
        if (tL < tR) return ordered(li, ri, tR, r, h, c);
        return ordered(ri, li, tL, r, h, c);
    }
    // bvh.cpp:850-881 intersect + :831-844 intersectDataStructure
    bool intersect(Ray& r, HitState& h, Counters* c) const {
        bool hit = false;
        if (!nodes.empty()) {
            float t0 = r.t;
            if (c) c->box_tests++;
            if (starts_in_box(r, nodes[0].box) || ray_box(nodes[0].box, r)) {
                r.t = t0;
                hit = visit(0, r, h, c);
            }
        }
        for (size_t s = 0; s < spheres.size(); s++) {
            if (ray_sphere(spheres[s].c, spheres[s].radius, r, h.normal)) {  // :878-879: material NOT written
                h.prim = (uint32_t)(tris.size() + s);
                hit = true;
            }
        }
        return hit;
    }
    // ray_tracing.cpp:202-213 applied to every scene mesh in load order: the
    // pre-BVH brute force (bvh.cpp:854-868, commented out upstream).
    bool brute_force(Ray& r, HitState& h) const {
        bool hit = false;
        for (uint32_t prim = 0; prim < tris.size(); prim++) {
            const Tri& t = tris[prim];
            const Vert &a = verts[t.a], &b = verts[t.b], &c = verts[t.c];
            if (ray_triangle(a.p, b.p, c.p, r, h.normal, a.n, b.n, c.n)) {
                h.material = tri_mesh[prim];
                h.prim = prim;
                hit = true;
            }
        }
        for (size_t s = 0; s < spheres.size(); s++)
            if (ray_sphere(spheres[s].c, spheres[s].radius, r, h.normal)) {
                h.prim = (uint32_t)(tris.size() + s);
                hit = true;
            }
        return hit;
    }
};

// ---------------------------------------------------------------------------
// Camera (framework/src/trackball.cpp:70-73, :92-103; src/main.cpp:691-694)
// ---------------------------------------------------------------------------
struct Quat {
    float w, x, y, z;
};
// glm::qua(vec3 eulerAngle) (glm 0.9.9.8 type_quat.inl)
inline Quat quat_from_euler(V3 e) {
    V3 h = e * 0.5f;
    V3 c = mk(std::cos(h.x), std::cos(h.y), std::cos(h.z));
    V3 s = mk(std::sin(h.x), std::sin(h.y), std::sin(h.z));
    Quat q;
    q.w = c.x * c.y * c.z + s.x * s.y * s.z;
    q.x = s.x * c.y * c.z - c.x * s.y * s.z;
    q.y = c.x * s.y * c.z + s.x * c.y * s.z;
    q.z = c.x * c.y * s.z - s.x * s.y * c.z;
    return q;
}
// glm operator*(qua, vec3): v + ((uv * q.w) + uuv) * 2
inline V3 quat_rotate(Quat q, V3 v) {
    V3 qv = mk(q.x, q.y, q.z);
    V3 uv = cross3(qv, v);
    V3 uuv = cross3(qv, uv);
    return v + ((uv * q.w) + uuv) * 2.0f;
}
struct Camera {
    V3 lookAt;
    V3 euler;  // radians
    float dist, fovy /*radians*/, aspect;
};
inline V3 camera_position(const Camera& c) {  // trackball.cpp:70-73
    return c.lookAt + quat_rotate(quat_from_euler(c.euler), mk(0, 0, -c.dist));
}
inline Ray camera_ray(const Camera& c, float px, float py) {  // trackball.cpp:92-103
    const float hh = std::tan(c.fovy / 2.0f);
    const float hw = c.aspect * hh;
    V3 dir = normalize3(mk(-px * hw, py * hh, 1.0f));
    Ray r;
    r.o = camera_position(c);
    r.d = quat_rotate(quat_from_euler(c.euler), dir);
    r.t = std::numeric_limits<float>::max();
    return r;
}

// ---------------------------------------------------------------------------
// Shading + recursion driver (src/main.cpp:61-310).
// Spherical lights (:168-218) draw randomUnitVector() from std::random_device (:46-59): unreproducible upstream, so the
// draws are a caller-supplied table of unit vectors here and an integer hash of (seed, pixel, level, light, sample)
// picks the entry -- the rule include/cgrt.h documents for cgrt_render_soft, restated independently below.
// ---------------------------------------------------------------------------
struct PointLight {
    V3 pos, color;
};
struct SphericalLight {  // scene.h:47-51
    V3 pos;
    float radius;
    V3 color;
};
static uint32_t fmix32(uint32_t h) {  // murmur3 32-bit finaliser
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
struct Shader {
    const Oracle* o;
    std::vector<PointLight> lights;
    std::vector<SphericalLight> slights;
    const float* units = nullptr;  // nunits x 3
    uint32_t nunits = 0, samples = 200, seed = 0;
    int maxLevel = 2;  // main.cpp:267 `level >= 2`

    V3 unit_draw(uint32_t pixel, uint32_t level, uint32_t light, uint32_t smp) const {
        uint32_t h = fmix32(seed ^ 0x9e3779b9u);
        h = fmix32(h ^ pixel);
        h = fmix32(h ^ (level * 0x01000193u + light));
        h = fmix32(h ^ smp);
        const float* u = units + 3 * (size_t)(h % nunits);
        return mk(u[0], u[1], u[2]);
    }

    static V3 material_kd(const Mat& m) { return mk(m.v[0], m.v[1], m.v[2]); }
    static V3 material_ks(const Mat& m) { return mk(m.v[3], m.v[4], m.v[5]); }

    // main.cpp:104-135
    bool point_in_shadow(V3 pointOn, const PointLight& l, uint64_t* nrays) const {
        V3 toLight = l.pos - pointOn;
        Ray r{pointOn, normalize3(toLight), std::numeric_limits<float>::max()};
        float eps = 0.001;
        r.o = r.o + eps * r.d;
        HitState hs{mk(0, 0, 0), -1, 0xffffffffu};
        if (nrays) (*nrays)++;
        if (o->intersect(r, hs, nullptr)) {
            if (r.t + eps >= length3(toLight)) return false;
            return true;
        }
        return false;
    }
    // main.cpp:84-98
    V3 diffuse(const PointLight& l, V3 toLight, const HitState& h, const Mat& m) const {
        float c = dot3(toLight, h.normal);
        if (c <= 0) return mk(0, 0, 0);
        return l.color * material_kd(m) * c;
    }
    // main.cpp:61-82.  `light.color * ks * pow(specularCos, shininess)` only compiles when pow(float, float)
    // yields float (glm's vec3 * scalar is a template on the element type), i.e. the float overload: powf.
    V3 specular(const Ray& r, const PointLight& l, V3 toLight, const HitState& h, const Mat& m) const {
        V3 refl = normalize3(reflect3(r.d, h.normal));
        float c = dot3(refl, toLight);
        if (c <= 0) return mk(0, 0, 0);
        float p = std::pow(c, m.v[6]);
        return l.color * material_ks(m) * p;
    }
    // main.cpp:160-235 (point-light loop :219-232)
    V3 shading(const Ray& r, const HitState& h, const Mat& m, uint64_t* nrays, uint32_t pixel, int level) const {
        V3 pointOn = r.o + r.d * r.t;
        V3 res = mk(0, 0, 0);
        // main.cpp:168-218: spherical lights come first
        for (size_t li = 0; li < slights.size(); li++) {
            const SphericalLight& sl = slights[li];
            PointLight l{sl.pos, sl.color};
            V3 toLight = normalize3(l.pos - pointOn);
            V3 dif = diffuse(l, toLight, h, m);
            V3 spec = specular(r, l, toLight, h, m);
            float counter = 0.0f;
            for (uint32_t i = 1; i <= samples; i++) {
                V3 rp = sl.pos + sl.radius * unit_draw(pixel, (uint32_t)level, (uint32_t)li, i - 1);
                // :178 aggregate initialisation in member order: origin, direction, then t from the new origin
                Ray nr;
                nr.o = pointOn + (float)(0.001) * normalize3(rp - pointOn);
                nr.d = normalize3(rp - pointOn);
                nr.t = length3(nr.o - rp);
                HitState hs{mk(0, 0, 0), -1, 0xffffffffu};
                float lightT = length3(nr.o - rp);
                if (nrays) (*nrays)++;
                if (!o->intersect(nr, hs, nullptr)) {
                    counter += 1.0f;
                } else if (nr.t > lightT) {
                    counter += 1.0f;
                }
            }
            counter = counter / (float)samples;  // :200 `/ 200.0f`
            res = res + dif * counter;
            res = res + spec * counter;
        }
        for (const PointLight& l : lights) {
            V3 toLight = normalize3(l.pos - pointOn);
            if (point_in_shadow(pointOn, l, nrays)) continue;
            res = res + diffuse(l, toLight, h, m);
            res = res + specular(r, l, toLight, h, m);
        }
        return res;
    }
    // main.cpp:265-295 trace + :241-264 shade
    V3 trace(int level, Ray r, uint64_t* nrays, uint32_t pixel = 0) const {
        if (level >= maxLevel) return mk(0, 0, 0);
        HitState h{mk(0, 0, 0), -1, 0xffffffffu};
        if (nrays) (*nrays)++;
        if (!o->intersect(r, h, nullptr)) return mk(0, 0, 0);
        // A hit whose material was never written (sphere-only hit) reads an
        // uninitialised Material upstream; restated as the default Material (mesh.h:17-23).
        Mat def{{0, 0, 0, 0, 0, 0, 1.0f, 1.0f}};
        const Mat& m = h.material >= 0 ? o->mats[h.material] : def;
        V3 direct = shading(r, h, m, nrays, pixel, level);
        // main.cpp:246: comma operator -> only ks.z is tested
        if (m.v[5] <= 0.01f) return direct;
        V3 refl = normalize3(reflect3(r.d, h.normal));
        Ray rr{r.o + r.d * r.t, refl, length3(r.d)};  // :254 note t = |direction| (about 1), not FLT_MAX
        float eps = 0.001;
        rr.o = rr.o + eps * rr.d;
        V3 rc = trace(level + 1, rr, nrays, pixel);
        return direct + rc * material_ks(m);
    }
};

thread_local std::string g_err;

}  // namespace

// ===========================================================================
// C interface for ctypes (tests/, smoke(), bench.py cpu_baseline)
// ===========================================================================
extern "C" {

struct OracleHit {
    float t;
    uint32_t prim;
    int32_t material;
    uint32_t hit;
    float nx, ny, nz;
    float pad;
};

void* oracle_scene_create(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh,
                          uint32_t ntris, const float* materials, uint32_t nmesh, const float* spheres /*S x 5: c, r, mat*/,
                          uint32_t nspheres) {
    Oracle* o = new Oracle();
    o->verts.resize(nverts);
    for (uint32_t i = 0; i < nverts; i++) {
        o->verts[i].p = mk(pos_nrm[6 * i], pos_nrm[6 * i + 1], pos_nrm[6 * i + 2]);
        o->verts[i].n = mk(pos_nrm[6 * i + 3], pos_nrm[6 * i + 4], pos_nrm[6 * i + 5]);
    }
    o->tris.resize(ntris);
    o->tri_mesh.resize(ntris);
    o->mesh_first.assign(nmesh + 1, ntris);
    int prev = -1;
    for (uint32_t i = 0; i < ntris; i++) {
        o->tris[i] = Tri{tri[3 * i], tri[3 * i + 1], tri[3 * i + 2]};
        int m = (int)tri_mesh[i];
        if (m < prev || m >= (int)nmesh || tri[3 * i] >= nverts || tri[3 * i + 1] >= nverts || tri[3 * i + 2] >= nverts) {
            delete o;
            return nullptr;
        }
        o->tri_mesh[i] = m;
        for (int k = prev + 1; k <= m; k++) o->mesh_first[k] = i;
        prev = m;
    }
    o->mesh_first[nmesh] = ntris;
    o->mats.resize(nmesh);
    for (uint32_t m = 0; m < nmesh; m++) std::memcpy(o->mats[m].v, materials + 8 * m, 32);
    for (uint32_t s = 0; s < nspheres; s++)
        o->spheres.push_back(SphereRec{mk(spheres[5 * s], spheres[5 * s + 1], spheres[5 * s + 2]), spheres[5 * s + 3],
                                       (int)spheres[5 * s + 4]});
    o->build();
    return o;
}
void oracle_scene_destroy(void* h) { delete (Oracle*)h; }
int oracle_num_levels(void* h) { return ((Oracle*)h)->num_levels(); }
int oracle_num_nodes(void* h) { return (int)((Oracle*)h)->nodes.size(); }
double oracle_build_seconds(void* h) { return ((Oracle*)h)->build_seconds; }

// Per node: is_leaf, level, child0, child1, ntris ; box lo/hi (6 floats)
void oracle_get_nodes(void* h, int32_t* meta /*N x 5*/, float* boxes /*N x 6*/) {
    Oracle* o = (Oracle*)h;
    for (size_t i = 0; i < o->nodes.size(); i++) {
        const ONode& n = o->nodes[i];
        size_t nt = n.ntris;
        meta[5 * i] = n.leaf;
        meta[5 * i + 1] = n.level;
        meta[5 * i + 2] = n.child[0];
        meta[5 * i + 3] = n.child[1];
        meta[5 * i + 4] = (int32_t)nt;
        const float b[6] = {n.box.lo.x, n.box.lo.y, n.box.lo.z, n.box.hi.x, n.box.hi.y, n.box.hi.z};
        std::memcpy(boxes + 6 * i, b, 24);
    }
}
// Global prim ids held by leaf `node`, in scan order. Returns count (writes up to cap).
uint32_t oracle_leaf_prims(void* h, int node, uint32_t* out, uint32_t cap) {
    Oracle* o = (Oracle*)h;
    uint32_t k = 0;
    for (const MeshRef& m : o->nodes[node].meshes)
        for (uint32_t p : m.tris) {
            if (k < cap) out[k] = p;
            k++;
        }
    return k;
}

// rays: n x 7 floats (origin, direction, t).  counters (optional): 4 x u64 totals
// {inner_visits, leaf_visits, tri_tests, box_tests}.  mode 0 = BVH (BoundingVolumeHierarchy::intersect),
// 1 = brute force over all triangles (ray_tracing.cpp:202-213 per mesh).
void oracle_intersect_batch(void* h, const float* rays, uint64_t n, OracleHit* out, uint64_t* counters, int mode,
                            int threads) {
    const Oracle* o = (const Oracle*)h;
    Counters tot;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        Counters loc;
#pragma omp for schedule(dynamic, 256)
        for (int64_t i = 0; i < (int64_t)n; i++) {
            const float* p = rays + 7 * i;
            Ray r{mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), p[6]};
            HitState hs{mk(0, 0, 0), -1, 0xffffffffu};
            bool hit = mode == 0 ? o->intersect(r, hs, counters ? &loc : nullptr) : o->brute_force(r, hs);
            out[i] = OracleHit{r.t, hs.prim, hs.material, hit ? 1u : 0u, hs.normal.x, hs.normal.y, hs.normal.z, 0.f};
        }
#pragma omp critical
        {
            tot.inner_visits += loc.inner_visits;
            tot.leaf_visits += loc.leaf_visits;
            tot.tri_tests += loc.tri_tests;
            tot.box_tests += loc.box_tests;
        }
    }
    if (counters) {
        counters[0] = tot.inner_visits;
        counters[1] = tot.leaf_visits;
        counters[2] = tot.tri_tests;
        counters[3] = tot.box_tests;
    }
}

// cam: lookAt(3) euler_rad(3) dist fovy_rad aspect = 9 floats.  Writes (x1-x0)*(y1-y0) rays, row-major
// over the rectangle, pixel (x,y) -> ndc (float(x)/W*2-1, float(y)/H*2-1)  (main.cpp:691-693).
void oracle_generate_rays(const float* cam, int W, int H, int x0, int y0, int x1, int y1, float* rays) {
    Camera c{mk(cam[0], cam[1], cam[2]), mk(cam[3], cam[4], cam[5]), cam[6], cam[7], cam[8]};
    size_t k = 0;
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            Ray r = camera_ray(c, float(x) / W * 2.0f - 1.0f, float(y) / H * 2.0f - 1.0f);
            const float v[7] = {r.o.x, r.o.y, r.o.z, r.d.x, r.d.y, r.d.z, r.t};
            std::memcpy(rays + 7 * k++, v, 28);
        }
}

// The reference's primary frame loop timed the way main.cpp:791-797 does (std::chrono around an
// `omp parallel for` over rows, main.cpp:653-656), primary rays only.  Returns seconds; hits optional.
double oracle_trace_primary_timed(void* h, const float* cam, int W, int H, int y0, int y1, OracleHit* out, int threads) {
    const Oracle* o = (const Oracle*)h;
    Camera c{mk(cam[0], cam[1], cam[2]), mk(cam[3], cam[4], cam[5]), cam[6], cam[7], cam[8]};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    auto t0 = std::chrono::steady_clock::now();
#pragma omp parallel for
    for (int y = y0; y < y1; y++)
        for (int x = 0; x != W; x++) {
            Ray r = camera_ray(c, float(x) / W * 2.0f - 1.0f, float(y) / H * 2.0f - 1.0f);
            HitState hs{mk(0, 0, 0), -1, 0xffffffffu};
            bool hit = o->intersect(r, hs, nullptr);
            if (out)
                out[(size_t)(y - y0) * W + x] =
                    OracleHit{r.t, hs.prim, hs.material, hit ? 1u : 0u, hs.normal.x, hs.normal.y, hs.normal.z, 0.f};
        }
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

// Full getFinalColor (main.cpp:298-310) per pixel of rows [y0,y1): rgb = 3 floats per pixel (not y-flipped).
// lights: L x 6 (pos, color).  Returns the number of rays cast.
// oracle_render_soft: + spherical lights, S x 7 (pos, radius, color), sampled from the unit-vector table (see Shader).
uint64_t oracle_render_soft(void* h, const float* cam, int W, int H, int y0, int y1, const float* lights, int nlights,
                            const float* slights, int nslights, const float* units, uint32_t nunits, uint32_t samples, uint32_t seed,
                            int maxLevel, float* rgb, int threads) {
    Shader sh;
    sh.o = (const Oracle*)h;
    sh.maxLevel = maxLevel;
    for (int i = 0; i < nslights; i++)
        sh.slights.push_back(SphericalLight{mk(slights[7 * i], slights[7 * i + 1], slights[7 * i + 2]), slights[7 * i + 3],
                                            mk(slights[7 * i + 4], slights[7 * i + 5], slights[7 * i + 6])});
    sh.units = units;
    sh.nunits = nunits;
    sh.samples = samples;
    sh.seed = seed;
    for (int i = 0; i < nlights; i++)
        sh.lights.push_back(PointLight{mk(lights[6 * i], lights[6 * i + 1], lights[6 * i + 2]),
                                       mk(lights[6 * i + 3], lights[6 * i + 4], lights[6 * i + 5])});
    Camera c{mk(cam[0], cam[1], cam[2]), mk(cam[3], cam[4], cam[5]), cam[6], cam[7], cam[8]};
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#endif
    uint64_t total = 0;
#pragma omp parallel for reduction(+ : total)
    for (int y = y0; y < y1; y++)
        for (int x = 0; x != W; x++) {
            Ray r = camera_ray(c, float(x) / W * 2.0f - 1.0f, float(y) / H * 2.0f - 1.0f);
            uint64_t n = 0;
            V3 col = sh.trace(0, r, &n, (uint32_t)(y * W + x));
            float* p = rgb + 3 * ((size_t)(y - y0) * W + x);
            p[0] = col.x;
            p[1] = col.y;
            p[2] = col.z;
            total += n;
        }
    return total;
}

uint64_t oracle_render(void* h, const float* cam, int W, int H, int y0, int y1, const float* lights, int nlights,
                       int maxLevel, float* rgb, int threads) {
    return oracle_render_soft(h, cam, W, H, y0, y1, lights, nlights, nullptr, 0, nullptr, 0, 0, 0, maxLevel, rgb, threads);
}

// ---- primitive intersectors, one call per element (for kernel-level parity tests) ----
// tri: n x 18 floats (v0 v1 v2 n1 n2 n3); rays n x 7; out: hit, t, normal
void oracle_ray_triangle(const float* tri, const float* rays, uint64_t n, OracleHit* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* q = tri + 18 * i;
        const float* p = rays + 7 * i;
        Ray r{mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), p[6]};
        V3 nrm = mk(0, 0, 0);
        bool hit = ray_triangle(mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]), mk(q[6], q[7], q[8]), r, nrm,
                                mk(q[9], q[10], q[11]), mk(q[12], q[13], q[14]), mk(q[15], q[16], q[17]));
        out[i] = OracleHit{r.t, 0, -1, hit ? 1u : 0u, nrm.x, nrm.y, nrm.z, 0.f};
    }
}
// plane: n x 4 (D, normal); out t + flag
void oracle_ray_plane(const float* plane, const float* rays, uint64_t n, OracleHit* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* p = rays + 7 * i;
        Ray r{mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), p[6]};
        bool hit = ray_plane(plane[4 * i], mk(plane[4 * i + 1], plane[4 * i + 2], plane[4 * i + 3]), r);
        out[i] = OracleHit{r.t, 0, -1, hit ? 1u : 0u, 0, 0, 0, 0};
    }
}
// box: n x 6 (lower, upper); out t + flag; `pad` carries startsInBox
void oracle_ray_box(const float* box, const float* rays, uint64_t n, OracleHit* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* p = rays + 7 * i;
        const float* b = box + 6 * i;
        Ray r{mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), p[6]};
        Box bb{mk(b[0], b[1], b[2]), mk(b[3], b[4], b[5])};
        bool inside = starts_in_box(r, bb);
        bool hit = ray_box(bb, r);
        out[i] = OracleHit{r.t, 0, -1, hit ? 1u : 0u, 0, 0, 0, inside ? 1.0f : 0.0f};
    }
}
// sphere: n x 4 (center, radius)
void oracle_ray_sphere(const float* sph, const float* rays, uint64_t n, OracleHit* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* p = rays + 7 * i;
        Ray r{mk(p[0], p[1], p[2]), mk(p[3], p[4], p[5]), p[6]};
        V3 nrm = mk(0, 0, 0);
        bool hit = ray_sphere(mk(sph[4 * i], sph[4 * i + 1], sph[4 * i + 2]), sph[4 * i + 3], r, nrm);
        out[i] = OracleHit{r.t, 0, -1, hit ? 1u : 0u, nrm.x, nrm.y, nrm.z, 0.f};
    }
}
// tri: n x 9 (v0 v1 v2) -> plane n x 4 (D, normal)   (trianglePlane)
void oracle_triangle_plane(const float* tri, uint64_t n, float* plane) {
    for (uint64_t i = 0; i < n; i++) {
        const float* q = tri + 9 * i;
        V3 nn;
        float D;
        triangle_plane(mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]), mk(q[6], q[7], q[8]), nn, D);
        plane[4 * i] = D;
        plane[4 * i + 1] = nn.x;
        plane[4 * i + 2] = nn.y;
        plane[4 * i + 3] = nn.z;
    }
}
// in: n x 15 (v0 v1 v2 n p) -> u8 flags  (pointInTriangle)
void oracle_point_in_triangle(const float* in, uint64_t n, uint8_t* out) {
    for (uint64_t i = 0; i < n; i++) {
        const float* q = in + 15 * i;
        out[i] = point_in_triangle(mk(q[0], q[1], q[2]), mk(q[3], q[4], q[5]), mk(q[6], q[7], q[8]), mk(q[9], q[10], q[11]),
                                   mk(q[12], q[13], q[14]));
    }
}
int oracle_max_threads() {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
}  // extern "C"
