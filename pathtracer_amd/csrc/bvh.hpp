// Host-built 4-wide BVH.  A binned-SAH binary tree with leaves of up to four triangles is built first
// and then collapsed: every 4-wide node adopts, largest box first, the children of its inner children
// and the triangles of its leaf children until it has four children.  One 128-byte node carries the boxes of all its children, so a traversal
// step is one aligned 128-B fetch followed by four slab tests; triangles are children like any
// other (their box is the triangle's own), fetched and tested exactly when they are visited.
#pragma once
#include <cstdint>
#include <vector>

namespace ptx
{

constexpr int32_t BVH_EMPTY = INT32_MIN; // child slot without a child

struct alignas(128) BvhNode
{
    float lo[3][4], hi[3][4]; // child boxes by axis (padded outward, float)
    int32_t child[4];         // >= 0: node index; BVH_EMPTY: none; other negatives: ~k, triangle k of tri_order
    int32_t pad[4];
};
static_assert(sizeof(BvhNode) == 128, "BvhNode must be one 128-byte line");

struct Bvh
{
    std::vector<BvhNode> nodes;     // nodes[0] is the root
    std::vector<int32_t> tri_order; // ordered reference k -> original face id (a face cut by reference splitting appears more than once)
    int n_leaves = 0;   // triangle references in the tree (>= triangles)
    int max_depth = 0;  // node levels below the root
    int stack_need = 0; // most entries a near-first traversal can have on its stack
    bool median = false; // built with median splits (the SAH tree was too deep for the stack)
};

// verts: [F][3][3] float.  Boxes are padded so that every hit the exact brute-force test could
// accept (barycentric slack 1e-10, reconstruction rounding) lies strictly inside its leaf's box.
// median: split at the median of the longest axis instead of the SAH plane (a balanced tree).
// split_budget: reference splitting -- at most this fraction of n_tris extra references are made by
// cutting the references of the emptiest boxes (0: none; never with median splits).  Off by default:
// measured on the two BVH scenes it trades a third of the triangle tests for a tenth more node visits
// at unchanged throughput; BVH_SPLIT_BUDGET is the budget the checks and tests exercise it with.
constexpr double BVH_SPLIT_BUDGET = 0.3;
void build_bvh(const float *verts, int n_tris, Bvh &out, bool median = false, double split_budget = 0.0);

} // namespace ptx
