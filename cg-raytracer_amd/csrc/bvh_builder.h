// bvh_builder.h -- host-side construction of the reference's BVH (topology + leaf order) on a flat
// layout, and its flattening into the device records of cgrt_layout.h.
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "cgrt_layout.h"

namespace cgrt {

struct HostScene {
    std::vector<float> pos_nrm;      // nverts x 6
    std::vector<uint32_t> tri;       // ntris x 3 (global vertex indices)
    std::vector<uint32_t> tri_mesh;  // ntris
    std::vector<float> materials;    // nmesh x 8
    std::vector<float> spheres;      // nspheres x 5
    uint32_t nverts = 0, ntris = 0, nmesh = 0, nspheres = 0;
};

// A node of the reference tree.  Its triangles are order[first .. first+count), already in the order
// intersectLeaf would scan them (mesh runs in node order, triangles in run order).
struct TopoNode {
    bool leaf;
    int level;
    Box6 box;
    int child[2];
    uint32_t first, count;
};

struct BuiltBvh {
    std::vector<TopoNode> nodes;    // level order, children adjacent (bvh.cpp:358-364)
    std::vector<uint32_t> order;    // prim ids; every node owns a contiguous range
    int levels = 0;
    // device-side flattening
    std::vector<NodePacket> packets;
    std::vector<LeafRec> leaves;
    std::vector<int> node_to_ref_index;  // node -> packet index (inner) or leaf index (leaf)
    std::vector<TriRecord> tris;         // leaf order
    std::vector<TriNormals> tri_normals;
    std::vector<SubNode> subnodes;       // in-leaf accelerators of all leaves (empty when disabled)
    float scene_absmax = 0;              // largest |vertex coordinate|
    bool geometry_finite = true;         // no NaN / infinite vertex coordinate
    bool has_wild = false;               // some triangle's float plane is degenerate (bvh_builder.cpp plane_is_tame)
    std::vector<uint8_t> leaf_wild;      // per leaf: holds such a triangle -> no accelerator, linear scan
    // certified walk (walk_fast.h): 4-wide tree over the leaves (its nodes are appended to `subnodes`), per-leaf box paths,
    // leaf of every record; fast_root == REF_NONE when the scene has none
    uint32_t fast_root = REF_NONE;
    std::vector<float> paths;            // nleaves x PATH_BOXES x 6
    std::vector<uint32_t> tri_leaf;      // per TriRecord (leaf order)
    // The three kinds of 64-byte records live in ONE device array [packets | subnodes | tris]; after
    // globalize_refs() every reference to a subnode or triangle record is an index into that array.
    uint32_t sub_base = 0, tri_base = 0;
    std::vector<SphereRecord> spheres;
    Box6 root_box{};
    uint32_t root_ref = REF_NONE;
    double build_seconds = 0;
};

struct BuildOptions {
    bool leaf_accel = true;  // build the in-leaf accelerator (results are identical either way)
    int sub_leaf_tris = SUB_LEAF_TRIS;
    int fast_open = SUB_MAX_DEPTH;  // levels of the leaves' accelerators handed to the top tree's SAH as separate items (build_fast_tree): all of them = a global tree over the runs
    int threads = 0;     // worker threads of the builders (0 = hardware concurrency, at most 16); the arrays do not depend on it
    int fast_tree = -1;  // certified walk's structures: -1 = whenever possible, except without in-leaf accelerators or below 16 triangles; 0 = never; 1 = whenever possible
};

// Returns false and sets err on invalid input.
bool build_reference_bvh(const HostScene& scene, const BuildOptions& opt, BuiltBvh& out, std::string& err);

}  // namespace cgrt
