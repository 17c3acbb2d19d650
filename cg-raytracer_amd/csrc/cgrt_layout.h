// cgrt_layout.h -- records shared by the host builder and the gfx950 kernels (HBM layout).
#pragma once
#include <stdint.h>

namespace cgrt {

// Child reference of a reference-topology node: bit 31 set -> leaf, low bits = index into the leaf
// table; clear -> index of the inner node's NodePacket.
static const uint32_t REF_LEAF = 0x80000000u;
static const uint32_t REF_NONE = 0xffffffffu;
// A leaf with an in-leaf accelerator is referenced by the accelerator's root directly: REF_LEAF | REF_LEAF_ACCEL | index
// of the root SubNode in the record array (26 bits), so that entering the leaf needs no LeafRec load; the leaf-table
// index of such a leaf is kept in pad[0] of its root node's SECOND SubNode for the paths that want first/count (linear scan).
static const uint32_t REF_LEAF_ACCEL = 0x40000000u;
static const uint32_t REF_INDEX26 = 0x03ffffffu;
static const int MAX_LEVELS = 12;  // bvh.cpp:48 maxDepth; per-ray stack never exceeds MAX_LEVELS - 1

// One inner node of the reference tree with BOTH child boxes (64 B, one cache-line half): the two
// slab tests + two startsInBox tests of intersectNonLeaf/intersectDeeper (bvh.cpp:715-736, :679-701)
// need exactly this and nothing else.
struct alignas(16) NodePacket {
    float lbox[6];  // left child AABB lower.xyz, upper.xyz
    float rbox[6];  // right child AABB
    uint32_t left;  // child refs
    uint32_t right;
    uint32_t pad[2];
};
static_assert(sizeof(NodePacket) == 64, "NodePacket must be 64 B");

// One reference leaf: a run of TriRecords in intersectLeaf scan order (bvh.cpp:538-551) and, when the
// in-leaf accelerator is built, the root of its sub-tree.
struct LeafRec {
    uint32_t first;  // first TriRecord
    uint32_t count;
    uint32_t sub_root;  // index of the leaf's first SubNode, REF_NONE when the leaf is scanned linearly
    uint32_t path_len;  // bits 0..7: boxes on the way from the root's child down to this leaf (its own box last): SceneDev::paths;
                        // bits 8..: which of them a certificate tests explicitly (bit k = box k; the others contain their successor)
};
static_assert(sizeof(LeafRec) == 16, "LeafRec must be 16 B");

// Half a node of a leaf's in-leaf accelerator or of the fast tree (DESIGN.md "In-leaf accelerator", "Certified walk"): two
// child boxes inline (64 B); a node is two consecutive SubNodes, 128-byte aligned.  A child is another SubNode
// (ref = index) or a run of 1..32 TriRecords (ref = REF_LEAF | (count - 1) << 26 | first record).
// A child box is stored per axis as the pair {lower, upper}: {lo.x, hi.x, lo.y, hi.y, lo.z, hi.z}, so that the two planes
// of an axis are one operand of the packed FP32 pipe (walk_exact.h slab_cons).
// In the FIRST half of a node pad[0], pad[1] repeat the second half's ref0, ref1, so that the device reads the four child
// references with one 16-byte load (7 loads per node step instead of 8); in the second half pad[0] carries the leaf index
// of an accelerator root (REF_LEAF_ACCEL).
struct alignas(16) SubNode {
    float box0[6];
    float box1[6];
    uint32_t ref0, ref1;
    uint32_t pad[2];
};
static const uint32_t SUB_RUN_MAX = 32;            // records per run
static const uint32_t SUB_MAX_RECORDS = 1u << 26;  // run references address records with 26 bits
static_assert(sizeof(SubNode) == 64, "SubNode must be 64 B");
// The in-leaf accelerator is 4 wide: every node is TWO consecutive SubNode records (four child boxes: half the dependent
// steps of a binary tree; measured +10 %, profiles/r1_exp_accelerator_width.txt).
static const int SUB_WIDTH = 4;
#ifndef CGRT_SUB_MAX_DEPTH
#define CGRT_SUB_MAX_DEPTH 5
#endif
static const int SUB_MAX_DEPTH = CGRT_SUB_MAX_DEPTH;               // levels of the accelerator under one reference leaf
static const int SUB_STACK_ENTRIES = (SUB_WIDTH - 1) * SUB_MAX_DEPTH;  // a step defers at most WIDTH-1 children
static const int SUB_LEAF_TRIS = 2;    // target triangles per run (1..4 measure within 3 % of each other on the dragon frame)
// The "fast tree" (walk_fast.h, DESIGN.md "Certified walk"): a 4-wide tree over the reference LEAVES whose children are the leaves'
// in-leaf accelerators (or runs of records for small leaves), made of the same 128-byte nodes.  TOP_MAX_DEPTH levels
// hold the <= 2^(MAX_LEVELS-1) leaves of a reference tree; its deferred children share the per-lane LDS stack.
static const int TOP_MAX_DEPTH = 6;
static_assert((1 << (2 * TOP_MAX_DEPTH)) >= (1 << (MAX_LEVELS - 1)), "the top tree must hold every reference leaf");
static const int FAST_STACK_ENTRIES = (SUB_WIDTH - 1) * (TOP_MAX_DEPTH + SUB_MAX_DEPTH);
static_assert(FAST_STACK_ENTRIES <= 2 * (MAX_LEVELS - 1) + SUB_STACK_ENTRIES, "the fast walk's stack must fit the exact walk's LDS slice");
static const int PATH_BOXES = (MAX_LEVELS - 1 + 1) & ~1;  // box slots per leaf in SceneDev::paths (6 floats each): the <= MAX_LEVELS - 1
                                                         // boxes of the path, padded to an even count (read two at a time)

// One triangle, everything the geometric test needs (64 B).  n and D are the ray-independent
// trianglePlane (ray_tracing.cpp:74-82) evaluated once on the host with the reference's arithmetic.
struct alignas(16) TriRecord {
    float v0[3];
    float v1[3];
    float v2[3];
    float n[3];
    float D;
    uint32_t prim_id;  // global triangle index
    uint32_t mesh_id;  // -> hitInfo.material
    uint32_t scan_k;   // position in the reference leaf's scan order (bvh.cpp:538-551); decides ties
};
static_assert(sizeof(TriRecord) == 64, "TriRecord must be 64 B");

// Vertex normals of a triangle, read once per ray for the accepted hit (ray_tracing.cpp:94-98).
struct TriNormals {
    float n1[3], n2[3], n3[3];
};
static_assert(sizeof(TriNormals) == 36, "TriNormals must be 36 B");

struct SphereRecord {
    float c[3];
    float radius;
};

struct Box6 {
    float lo[3], hi[3];
};
// SubNode box slot <-> Box6
inline void sub_box_store(float dst[6], const Box6& b) {
    for (int a = 0; a < 3; a++) {
        dst[2 * a] = b.lo[a];
        dst[2 * a + 1] = b.hi[a];
    }
}
inline Box6 sub_box_load(const float src[6]) {
    Box6 b;
    for (int a = 0; a < 3; a++) {
        b.lo[a] = src[2 * a];
        b.hi[a] = src[2 * a + 1];
    }
    return b;
}

// Everything a kernel needs, passed by value.
struct SceneDev {
    // ONE device array of 64-byte records, [NodePackets | SubNodes | TriRecords]; the three typed views
    // below all point at its start and every reference is an index into it (the traversal loop has one
    // load site whatever kind of record a lane needs next).
    const NodePacket* packets;
    const SubNode* subnodes;
    const TriRecord* tris;
    uint32_t tri_base;              // index of the first TriRecord
    const LeafRec* leaves;
    const TriNormals* tri_normals;  // indexed by (record index - tri_base)
    const SphereRecord* spheres;
    // certified walk (walk_fast.h); fast_root == REF_NONE: the scene has no fast tree and every ray takes the exact walk
    const uint32_t* tri_leaf;  // leaf-table index of every TriRecord, indexed by (record index - tri_base)
    const float* paths;        // per leaf PATH_BOXES x {lower.xyz, upper.xyz}: the boxes the reference tests on the way to the leaf
    uint32_t fast_root;
    Box6 root_box;
    float scene_eps;    // 2^-16 * largest |coordinate| of the scene: conservative slack of the in-leaf boxes
    uint32_t fast_boxes;  // every reference box coordinate is 0 or in [2^-40, 2^40]: trace_kernels.hip RayFast
    uint32_t root_ref;  // REF_NONE when the scene has no meshes (bvh.cpp:870)
    uint32_t ntris;
    uint32_t nspheres;
    uint32_t npackets;
    uint32_t nleaves;
};

// Camera constants evaluated once on the host (trackball.cpp:70-73, :92-103): position, quaternion,
// half extents of the image plane.  The per-pixel part runs on the device.
struct CameraDev {
    float pos[3];
    float q[4];  // w x y z
    float half_w, half_h;
};

// Frame decomposition.  The rectangle is cut into 8x8-pixel tiles (one wave each) grouped into
// super-tiles of 8x8 tiles (64x64 pixels).  Super-tile i (row-major) belongs to rank i % nranks, and
// inside a rank its j-th super-tile is traced by the workgroups with blockIdx % 8 == j % 8, i.e. (with
// the round-robin workgroup->XCD placement observed on MI355X) by ONE XCD, whose 4 MiB L2 then holds
// that screen region's nodes and triangles instead of sharing every region with the 7 other L2s.
// Placement is a speed matter only: results do not depend on it.
// Frame hints (walk_exact.h hinted_tile_pixel / hint_finish, capi.cpp FrameHints): what the previous frame of the same shape
// learnt about its tiles, for the next one.  Every wave measures its own wall time; a wave that took long puts its 8x8 tile on the
// frame's HARD LIST.  The next frame traces the tiles of that list FIRST (its first cap * per_tile workgroups; the regular
// workgroups skip them) -- in the same 64-ray waves (per_tile = 1: the long waves no longer start in the frame's last round), or,
// for small frames whose time is the time of their longest wave, as four waves of 16 rays (per_tile = 4: the quad tail from the
// first step, two tile rows each).  Only the order and the layout change: every owned pixel is traced exactly once, by the hard
// workgroup iff the pair (list slot i, flag of the tile) is consistent -- list[i] == tile and flag[tile] == generation << 16 | i + 1
// -- which both kinds of workgroup evaluate on the same read-only memory.  Three sets rotate: a frame reads one, writes the next
// and zeroes the counter of the third; the generation stamp makes stale flags harmless without clearing them.
struct HintDev {  // one per rotation phase, resident in device memory
    const uint32_t* flag_r;  // per tile: generation << 16 | slot + 1 (what the previous frame left)
    const uint32_t* list_r;  // hard tiles (tile = ty * tiles_x + tx)
    const uint32_t* count_r;
    uint32_t* flag_w;        // what this frame leaves
    uint32_t* list_w;
    uint32_t* count_w;
    uint32_t* count_z;       // the third set's counter, zeroed by this frame
    uint32_t* ctl;           // shared by the phases: [0] the threshold in force (s_memrealtime ticks, 100 MHz, for a 64-ray wave)
    uint32_t* mailbox;       // pinned host memory: {generation, length of the list this frame reads, threshold} -- what the host
                             // learns about the lists without ever waiting for the device (capi.cpp attach_hints)
    uint32_t cap;            // tiles a list holds (<= 0xfffe)
    uint32_t per_tile;       // workgroups per hard tile: 1 or 4
    // The threshold follows the scene: how many tiles are worth treating specially is a matter of their SHARE of the frame (best
    // at 0.5 - 5 % of the tiles, 0.5 - 2 % on every scene tried; from ~7 % on the frame gets slower: profiles/r3_frame_hints.txt), not of an absolute time -- 45 us
    // lists 5 % of the regular stand-in's tiles at 960x540 and 17 % of the irregular one's at 800x800.  Workgroup 0 of every frame
    // raises the threshold by 1/8 when the list it reads is longer than `hi` and lowers it by 1/16 (not below thr_floor) when it
    // is shorter than `lo`.  A 16-ray wave counts as long from 5/9 of the threshold.
    uint32_t thr_floor, thr_ceil;
    uint32_t lo, hi;
    uint32_t thr_min;        // no wave faster than this can be hard: it ends without looking any further
};

struct FrameDev {
    int W, H;
    int x0, y0, x1, y1;
    int tiles_x, tiles_y;  // 8x8 tiles covering the rectangle
    int st_x, st_y;        // super-tiles covering the rectangle
    int rank, nranks;
    uint32_t nst_rank;     // super-tiles this rank owns
    int block;             // threads per workgroup of the launch (64, 128 or 256)
    uint32_t nblocks;      // workgroups launched: 64 / (block / 64) per super-tile, super-tiles padded to 8
    int packed;            // 1: results are written at blockIdx * block + threadIdx (the rank's pixels back to back, in the kernel's
                           // own order: one contiguous download per device) instead of at y * W + x
    const HintDev* hint;   // nullptr: no hints (the workgroups map to the tiles in launch order)
    uint32_t hint_blocks;  // workgroups in FRONT of the nblocks regular ones: per_tile for each of the hint_blocks / per_tile list entries this
                           // launch can take (<= cap; entries beyond are traced by their regular workgroups)
    uint32_t hint_rgen;    // generation of the set to read (0: nothing to read -- a first frame still writes)
    uint32_t hint_wgen;    // generation this frame stamps on what it writes
};
static const int ST_TILES = 8;  // tiles per super-tile side
// Workgroups of the traversal kernels hold 64, 128 or 256 threads (chosen per launch: FrameDev::block, blockDim.x): one wave
// per 8x8 tile, so a 64x64 super-tile takes 64 / (block / 64) consecutive workgroups of one blockIdx % 8 residue.  CGRT_BLOCK
// is the largest size (launch bounds).  The per-lane stacks are laid out per WAVE -- slot s of lane l of wave w at
// w * CGRT_STACK_SLOTS * 64 + s * 64 + l -- so that the walk code does not depend on the workgroup size.
#define CGRT_BLOCK 256
#define CGRT_STRIDE 64

}  // namespace cgrt
