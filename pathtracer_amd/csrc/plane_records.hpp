// f32 plane records of the all-triangles intersector: one record per triangle, or per pair of
// exactly coplanar triangles that share an edge (the two halves of a quad face).  See kernels.hpp
// for the record layout and the device-side test.
#pragma once
#include "scene.hpp"

#include <cstdint>
#include <vector>

namespace ptx
{

constexpr int PLANE_REC_STRIDE = 32;

struct PlaneRecords
{
    std::vector<float> rec;        // [n_recs][32]
    std::vector<int32_t> rec_tri;  // [64] candidate bit (2 r: triangle A, 2 r + 1: triangle B) -> triangle, -1: none
    std::vector<int32_t> tri_rec;  // [n_tris] triangle -> record
    int n_recs = 0;
    int n_pairs = 0;
    float smax = 0.f;              // largest |c| (plane offset) of the records, rounded up
};

// verts: [n_tris][9] float positions by face id.  Triangles stay in face order.
void build_plane_records(const float *verts, int n_tris, PlaneRecords &out);

// Which records can possibly block a segment between a point of record i and a point of the
// parallelogram `light` (4 corners, [4][3])?  Returns one bit mask per record i (bit j: record j is
// a potential blocker).  Record j is ruled out only if its supporting plane has all of record i and
// all of the light in one closed half space, at least one of the two well clear of the plane: a
// segment between them then never crosses the plane in its interior.
std::vector<uint32_t> blocker_masks(const float *verts, const PlaneRecords &pr, const double light[4][3]);

// corners of the parallelogram a square light is sampled on: P0 + x1 (P1 - P0) + x2 (P2 - P1)
void square_light_corners(const Light &L, double corners[4][3]);

} // namespace ptx
