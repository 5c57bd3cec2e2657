// Host-built, flattened 2-wide BVH (binned SAH).  One 64-byte node carries the boxes of BOTH
// children, so a traversal step is one aligned 64-B fetch followed by two slab tests.
#pragma once
#include <cstdint>
#include <vector>

namespace ptx
{

struct alignas(64) BvhNode
{
    float lo0[3], hi0[3]; // child 0 box (padded outward, float)
    float lo1[3], hi1[3]; // child 1 box
    int32_t c0, c1;       // >= 0: node index; < 0: leaf, ~c = (first << 3) | (count - 1), count <= 8
    int32_t pad[2];
};
static_assert(sizeof(BvhNode) == 64, "BvhNode must be one 64-byte line");

struct Bvh
{
    std::vector<BvhNode> nodes;    // nodes[0] is the root (a single-leaf scene gets one node)
    std::vector<int32_t> tri_order; // ordered triangle k -> original face id
    int n_leaves = 0, max_depth = 0, max_leaf = 0;
};

// verts: [F][3][3] float.  Boxes are padded so that every hit the exact brute-force test could
// accept (barycentric slack 1e-10, reconstruction rounding) lies strictly inside its leaf's box.
void build_bvh(const float *verts, int n_tris, Bvh &out, int max_leaf_tris = 4);

} // namespace ptx
