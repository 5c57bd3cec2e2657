// Path-tracing kernels for gfx950 (wave64).  Templated on the scalar type R and on the intersector
// (BRUTE: two-phase all-triangles test -- f32 plane records from a wave-uniform table the compiler
// reads through the scalar cache, then the exact test on the survivors; otherwise the 4-wide BVH
// with a per-lane stack in LDS).
//
// Kernel <-> reference map (paths relative to the reference root):
//   primary_ray / k_camera_rays  cPathTracer::GenerateRay        src/tracing/cPathTracer.cpp:129-213
//   closest_hit / k_raycast .... cAccelStruct::RayCast           src/geometry/cAccelStruct.cpp:28-54
//   visible / k_vistest ........ cAccelStruct::VisTest           src/geometry/cAccelStruct.cpp:56-87
//   k_trace_f64 / k_trace_f32,   the bounce loop of one primary ray, cPathTracer.cpp:555-673:
//   k_trace_rings_* (bvh_tracer.hpp)
//                                RayCast, miss / emitter termination, shading normal, NEE over all
//                                lights with their VisTest (cBxDF::Sample_Li, src/tracing/cBxDF.cpp:
//                                16-52), next direction (Sample_f); one (direct, coefficient) record
//                                per bounce
//   resolve (in trace_body) .... back-to-front combine with the per-bounce max-clamp,
//   k_accumulate                 and the per-pixel average in sample order, cPathTracer.cpp:679-692
#pragma once
#include "bvh.hpp"
#include "device_math.hpp"

#include <type_traits>

namespace ptx
{

constexpr int BLOCK = 256;      // threads per workgroup (4 waves)
constexpr int BVH_STACK_MAX = 96; // most per-lane traversal stack entries (1 KB of LDS per entry and workgroup)
constexpr int WAVE_STATS = 6;     // statistics columns per wave
constexpr int MAX_MATS = 0xFFFE; // the path state word keeps the tail material in 16 bits

template <class R> struct DScene
{
    int n_tris, n_mats, n_lights, n_square;
    int brute, enable_indirect, enable_bary, n_recs;
    float pq_smax;             // largest |c| of the plane records
    float bvh_cmax, pad_[2];   // largest |coordinate| of the BVH's root box (slab margins, trav_init)
    const R *tri_pre;          // [n_tris][12] in intersector order: p1, p1-p2, p1-p3, cofactors
    const float *tri_pq;       // [n_recs][32] f32 plane records (all-triangles intersector)
    const int32_t *rec_tri;    // [64] candidate bit -> triangle, -1: none
    const int32_t *tri_rec;    // [n_tris] triangle -> plane record
    const uint32_t *rec_block; // [32] records that can block a segment from record i to light 0 (one-square-light scenes)
    const int32_t *tri_face;   // intersector order -> face id
    const float *verts;        // [n_faces][9]  by face id
    const float *normals;      // [n_faces][9]
    const int32_t *mat_ids;    // [n_faces]
    const R *mats;             // [n_mats][MAT_STRIDE]
    const R *lights;           // [n_lights][LIGHT_STRIDE]
    const R *tri_nrm;          // [n_tris][4] intersector order: shading normal + flag (1: use it; 0: interpolate)
    const int32_t *tri_mat;    // [n_tris] intersector order: material id (-1: none)
    const uint4 *bvh_nodes;    // [n_nodes][4] 64-byte nodes of the 4-wide BVH (child boxes quantised to 8 bits)
    const uint4 *bvh_tris;     // [n_tris][3] triangle records in intersector order: the nine float coordinates, face id
    int n_nodes, id_bits;      // id_bits: bits of a stack entry that hold the record id
    int sq_light;              // index of the first square light (-1: none)
    int lds_nodes;             // the first lds_nodes node records are also in LDS, behind the stack levels (tracers only)
    int stack_levels;          // traversal stack levels held in LDS; deeper levels go to `spill`
    int spill_stride;          // lanes of the launch
    uint32_t *spill;           // [levels beyond stack_levels][spill_stride] (null when the LDS stack holds the whole bound)
};

struct DCamera
{
    double m[16];    // raster -> world
    double pos[3];
    int width, height;
};

// region of the image one batch covers, and its mapping to path slots
struct DBatch
{
    int row0, n_rows;        // local rows of this batch: [row0, row0 + n_rows)
    int col0, n_cols;        // columns [col0, col0 + n_cols)
    int band, row_stride, row_phase, region_row0; // local row -> image row (interleaved bands)
    int image_width;         // W of the full image (pixel index of the RNG key = row * W + col)
    int n_samples;           // samples of this batch
    int sample0;             // global index of its first sample
    int spp_total, depth;
    uint64_t seed;
    int64_t n_pix;           // n_rows * n_cols
    int64_t n_paths;         // n_pix * n_samples
};

__device__ __forceinline__ int local_row_to_image_row(const DBatch &b, int lr)
{
    if (b.row_stride <= 1)
        return b.region_row0 + lr;
    int bi = lr / b.band, off = lr % b.band;
    return b.region_row0 + (bi * b.row_stride + b.row_phase) * b.band + off;
}

// scalars of one record slot: direct rgb, coefficient rgb, two unused -- 64 bytes in f64 (32 in
// f32), aligned, so that a record is one memory sector however the slots are handed out
constexpr int REC_STRIDE = 8;
constexpr int LANE_SLOTS = 192; // record slots of a wave of the lane tracers: 64 paths on the lanes + at most 127 awaiting their resolve pass
constexpr int RING_SLOTS = 320; // of the ring tracer: at most 192 paths in flight + 127 awaiting their resolve pass

// per-batch device state.  pid = sample_local * n_pix + pixel_local numbers the paths of a batch.
template <class R> struct DState
{
    R *records;                     // [n_waves][slots][depth][REC_STRIDE]: the (direct, coefficient) records of the paths a wave
                                    // has in flight; a path holds one slot from its start to its resolve pass
    R *result;                      // [3][N] by pid: resolved per-sample radiance (written by the tracer)
    unsigned long long *wave_stats; // [n_waves][WAVE_STATS] segments, shadow tests, wave iterations, records, BVH node visits, BVH triangle visits
    unsigned long long *stats;      // [8] totals (filled by k_reduce_stats)
    unsigned int *pool;             // next unassigned path id of the current batch
    char *rings;                    // [n_waves][ring_bytes<R>()] ray and hit rings of the ring tracer (BVH scenes)
    int64_t N;                      // capacity (paths per batch)
    int n_waves;                    // waves of the current batch
    int slots;                      // record slots per wave (LANE_SLOTS or RING_SLOTS)
    int ring_batch;                 // ring tracer: lanes that hand over hits / take rays together
};

// the record of (wave g, slot, bounce)
template <class R> __device__ __forceinline__ R *record_at(const DState<R> &st, int g, uint32_t slot, int depth, uint32_t bounce)
{
    return st.records + ((size_t(g) * size_t(st.slots) + slot) * size_t(depth) + bounce) * REC_STRIDE;
}

// ---------------------------------------------------------------------------------------------
// intersectors
// ---------------------------------------------------------------------------------------------
// ---------------------------------------------------------------------------------------------
// two-phase all-triangles intersector.  Phase 1 walks a short list of f32 "plane records" in a
// wave-uniform loop (records read through the scalar cache).  A record is the supporting plane of
// one triangle A = (P0, P0 + e1, P0 + e2), or of two exactly coplanar triangles A and B that share
// the edge (P0 + e1, P0 + e2) -- the two halves of a quad face.  The test intersects the ray with
// the plane, expresses the point in the (e1, e2) basis and only REJECTS what the exact test is
// certain to reject, using forward error bounds that grow with 1 / cos^2 of the incidence angle;
// survivors form a per-lane bit mask (bit 2r: triangle A of record r, bit 2r + 1: triangle B).
// Phase 2 runs the exact test of the reference (in R) on each lane's own candidates, triangle data
// from LDS, with the reference's tie rule (equal distance -> lowest face id), so hits, distances
// and tie-breaking are identical to testing every triangle exactly.
//
// Record (32 floats), with u = a.(X - P0), v = b.(X - P0) the coordinates of the plane point X in
// the (e1, e2) basis:
//   0-2 unit normal n, 3 c = n.P0,   4-6 a,   8-10 b,
//   12,13 centre of the admissible range of a.X (= u + a.P0) and its half width + E0,
//   14,15 the same for b.X (A alone: u, v in [0,1]; with B: the box around both),
//   16 E, 17 2 E0, 18 1 + a.P0 + b.P0, 20 Et, 22 0 (with B) or +inf (A alone).
// With r = 1 / |n.d|, To = max|o| and W = (1.75 To + max_records |c|) r^2 the computed t, u, v
// are within Et W and E W + E0 of their exact values (E = 64 eps max(|a|_1, |b|_1),
// E0 = 16 eps (max(|a.P0|, |b.P0|) + 1), Et = 32 eps; first-order bounds are 13, 24 and 5 eps,
// including the rounding of the ray and of the record to f32).
// ---------------------------------------------------------------------------------------------
constexpr int PQ_STRIDE = 32;
#ifndef PTX_PQ_WIDE_LOADS
#define PTX_PQ_WIDE_LOADS 1
#endif
constexpr int BRUTE_MAX_TRIS = 64; // candidate mask is one 64-bit word ...
constexpr int BRUTE_MAX_RECS = 32; // ... of two bits per record

// LDS tables of the all-triangles intersector (one copy per workgroup)
template <class R> struct BruteLds
{
    R tri[BRUTE_MAX_TRIS * 12];         // exact-test records, by triangle
    int rec_tri[2 * BRUTE_MAX_RECS];    // mask bit -> triangle (-1: none)
    int tri_rec[BRUTE_MAX_TRIS];        // triangle -> plane record
    float rec_n[BRUTE_MAX_RECS * 4];    // unit normal and plane offset of the plane records
    uint32_t rec_block[BRUTE_MAX_RECS]; // potential blockers between record i and light 0
};
constexpr int LDS_WORDS_BRUTE = int(sizeof(BruteLds<double>) / 4);

// BVH kernels: the traversal stacks, [levels][BLOCK] entries, sized at launch to what the scene's
// tree can need
extern __shared__ __align__(16) int dyn_stack[];

// LDS scratch of the intersecting kernels: the tables above, or the per-lane BVH stacks
template <class R, bool BRUTE> __device__ __forceinline__ int *intersector_scratch(const DScene<R> &S, int *lds)
{
    if (BRUTE)
    {
        BruteLds<R> *L = reinterpret_cast<BruteLds<R> *>(lds);
        for (int k = threadIdx.x; k < S.n_tris * 12; k += BLOCK)
            L->tri[k] = S.tri_pre[k];
        for (int k = threadIdx.x; k < 2 * BRUTE_MAX_RECS; k += BLOCK)
            L->rec_tri[k] = S.rec_tri[k];
        for (int k = threadIdx.x; k < S.n_tris; k += BLOCK)
            L->tri_rec[k] = S.tri_rec[k];
        for (int k = threadIdx.x; k < S.n_recs * 4; k += BLOCK)
            L->rec_n[k] = S.tri_pq[(k >> 2) * PQ_STRIDE + (k & 3)]; // n and c = n.P0
        for (int k = threadIdx.x; k < BRUTE_MAX_RECS; k += BLOCK)
            L->rec_block[k] = S.rec_block[k];
        __syncthreads();
        return lds; // shared tables: same pointer for every lane
    }
    return lds + threadIdx.x; // BVH: stack slot [level][lane]
}

constexpr float PQ_ET = 32.0f / 16777216.0f; // Et of every record (32 eps)

// NEAREST: also report the candidate record with the smallest positive lower bound of its hit
// distance (near_rec, -1: none) -- the closest-hit search tests it first.
template <bool NEAREST, class R>
__device__ __forceinline__ unsigned long long prefilter_mask(const float *__restrict__ PQ, int n_recs, float smax, V3<R> o, V3<R> d,
                                                             float tmin, float tmax, uint32_t rec_mask, int &near_rec)
{
#pragma clang fp contract(fast) // rounding differences are inside the error bounds of the records
    const float ox = float(o.x), oy = float(o.y), oz = float(o.z);
    const float dx = float(d.x), dy = float(d.y), dz = float(d.z);
    const float Tmax = 1.75f * fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + smax;
    unsigned long long mask = 0;
    uint32_t near_key = 0xFFFFFFFFu;
    near_rec = -1;
    // only the records named by rec_mask (wave-uniform) are visited
    uint32_t todo = uint32_t(__builtin_amdgcn_readfirstlane(int(rec_mask)));
    if (n_recs < 32)
        todo &= (1u << n_recs) - 1u;
    // The records never change while a kernel runs.  Reading them through the constant address
    // space tells the compiler so: the wave-uniform loads stay scalar-cache loads even though the
    // tracer has an atomic (its path-id pool) that it must assume may write anywhere.
    typedef const float __attribute__((address_space(4))) const_float;
    const_float *PQc = (const_float *)(uintptr_t)PQ;
    // every field is read up front (one burst of scalar loads); the comparisons are false on NaN /
    // infinity soup, which keeps the candidate, and are combined without short-circuit evaluation
    // (no branches).  Every vector instruction below has at most one scalar operand.
    while (todo)
    {
        const int r = __builtin_ctz(todo);
        todo &= todo - 1u;
        const_float *P = PQc + r * PQ_STRIDE;
#if PTX_PQ_WIDE_LOADS
        // the 24 leading floats of the record in two scalar loads (16 + 8 dwords; the record is 128-byte
        // aligned) instead of one load per run of fields between its unused words: every instruction,
        // scalar loads included, takes an issue slot of the SIMD
        typedef float f32x16 __attribute__((ext_vector_type(16)));
        typedef float f32x8 __attribute__((ext_vector_type(8)));
        const f32x16 q0 = *(const f32x16 __attribute__((address_space(4))) *)P;
        const f32x8 q1 = *(const f32x8 __attribute__((address_space(4))) *)(P + 16);
        const float nx = q0[0], ny = q0[1], nz = q0[2], pc = q0[3];
        const float ax = q0[4], ay = q0[5], az = q0[6], bx = q0[8], by = q0[9], bz = q0[10];
        const float uc = q0[12], uh = q0[13], vc = q0[14], vh = q0[15];
        const float E = q1[0], E02 = q1[1], k1 = q1[2], Et = q1[4], bthr = q1[6];
#else
        const float nx = P[0], ny = P[1], nz = P[2], pc = P[3];
        const float ax = P[4], ay = P[5], az = P[6], bx = P[8], by = P[9], bz = P[10];
        const float uc = P[12], uh = P[13], vc = P[14], vh = P[15];
        const float E = P[16], E02 = P[17], k1 = P[18], Et = P[20], bthr = P[22];
#endif
        const float nd = nx * dx + ny * dy + nz * dz;
        const float no = pc - (nx * ox + ny * oy + nz * oz);
        const float rr = __builtin_amdgcn_rcpf(nd);
        const float t = no * rr;
        const float W = Tmax * (rr * rr);
        const float X = fmaf(t, dx, ox), Y = fmaf(t, dy, oy), Z = fmaf(t, dz, oz);
        const float u = ax * X + ay * Y + az * Z; // u + a.P0
        const float v = bx * X + by * Y + bz * Z; // v + b.P0
        float We = E * W;
        asm volatile("" : "+v"(We)); // keep the product in a register: folded into the three sums below it
                                     // would need a second scalar operand (an extra move) in each
        const float et = Et * W;
        const bool out = (fabsf(u - uc) > uh + We) | (fabsf(v - vc) > vh + We) | (t + et < tmin) | (t - et > tmax);
        const float s = (u + v) - k1, es = fmaf(We, 2.0f, E02);
        const bool ka = !(out | (s > es));
        const bool kb = !(out | (s + es < bthr));
        const uint32_t bits = (ka ? 1u : 0u) | (kb ? 2u : 0u);
        if (NEAREST)
        {
            // nearest candidate so far: one unsigned minimum over keys "bits of the positive lower bound of
            // its distance, the low five replaced by the record" (no key: all ones).  Which of two records a
            // few ulps apart comes out only orders the exact tests that follow, never their result.
            const float tlo = t - et;
            const uint32_t key = ((ka | kb) & (tlo > 0.0f)) ? ((__float_as_uint(tlo) & ~31u) | uint32_t(r)) : 0xFFFFFFFFu;
            near_key = key < near_key ? key : near_key;
        }
        // two bits per record into the 64-bit mask: one 64-bit shift by a wave-uniform amount, no branch on r
        mask |= (unsigned long long)bits << (2 * r);
    }
    if (NEAREST)
        near_rec = near_key == 0xFFFFFFFFu ? -1 : int(near_key & 31u);
    return mask;
}

// A ray that starts on triangle `own` meets the plane of that triangle (and of its coplanar
// partner) at distance ~0, which the exact test ignores unless the ray grazes the plane: the
// candidates of that record are dropped when |n.d| is far from zero.
template <class R>
__device__ __forceinline__ unsigned long long drop_own_plane(unsigned long long mask, const BruteLds<R> *L, int own, V3<R> d)
{
    if (own >= 0)
    {
        const int r = L->tri_rec[own];
        const float nd = L->rec_n[r * 4] * float(d.x) + L->rec_n[r * 4 + 1] * float(d.y) + L->rec_n[r * 4 + 2] * float(d.z);
        if (fabsf(nd) > 1e-3f)
            mask &= ~(3ull << (2 * r));
    }
    return mask;
}

// ---------------------------------------------------------------------------------------------
// 4-wide BVH.  Nodes are 64 bytes: the corner and cell size of an 8-bit grid laid over the union of
// the child boxes, the child boxes in grid coordinates (rounded outward by the host, which checks
// it in exact arithmetic), four child record ids.  Triangles are records of their own (the nine
// vertex coordinates as the floats they are and the face id: 48 bytes in either precision).  A
// traversal step fetches ONE record and either runs four slab tests (node: four 16-byte loads per
// lane) or the exact test (triangle: three), so every step costs one memory round trip,
// and as few vector-cache accesses as the data allows -- with 64 lanes at 64 different records it
// is the number of load instructions per step, not the bytes, that the memory pipeline feels
// (128-byte records with full-precision boxes and precomputed cofactors: 8 % slower on the 336 k
// triangle room; doubling the footprint of the tree instead changes nothing).
//
// The slab tests run in f32 against an interval around the ray origin that contains the true origin
// (rounding it to f32 moves it by at most 2^-24 of its magnitude; the interval is 2^-22 wide per
// side) and only ever make a box look nearer and longer: plane distances are computed as
// (cell * slope) * coordinate + (corner - origin) * slope, near planes from the end of the origin
// interval that makes them nearer and lowered by a bound of the roundings of that sum, far planes
// the other way round.  So a box that holds
// a triangle whose exact distance could win or tie is never skipped, and hits, distances and ties
// are those of testing every triangle exactly.
//
// Stack entries keep the bits of the (non-negative) entry distance above the record id (the low
// S.id_bits bits are the id: the distance is truncated, hence still a lower bound): children are
// ordered by sorting their entries, and an entry whose distance has fallen behind the closest hit is
// dropped when it is popped, without touching memory.
// ---------------------------------------------------------------------------------------------
template <class R> struct Trav
{
    int cur;      // record to visit next (< S.n_nodes: node, else triangle cur - S.n_nodes)
    int sp, target, best_face;
    int any;      // ANY == 2 only: this lane's traversal is an any-hit (visibility) one
    R best;       // closest accepted distance so far (any-hit: length of the segment)
    R best_t;     // ray parameter of that hit: the hit point is o + best_t * d
    float onx, ony, onz, ofx, ofy, ofz; // end of the origin interval that makes the near / far plane of a slab nearer / farther
    float ivx, ivy, ivz;                // 1 / d, finite
};
template <class R> __device__ __forceinline__ void trav_init(const DScene<R> &S, Trav<R> &T, V3<R> o, V3<R> d, R limit = Consts<R>::big)
{
    T.cur = 0, T.sp = 0, T.target = -1, T.best_face = 0x7fffffff;
    T.any = 0;
    T.best = limit;
    T.best_t = R(0);
    const float fx = float(o.x), fy = float(o.y), fz = float(o.z);
    const float omax = fmaxf(fmaxf(fabsf(fx), fabsf(fy)), fabsf(fz));
    // Half width of the origin interval: 2^-22 |o| for the rounding of the origin to f32, plus the
    // bound of the roundings of a slab's plane distance t = A q + B (A = cell * slope, q <= 255,
    // B = (corner - origin) * slope), taken once per ray instead of once per node: those roundings
    // are below 2^-21 (255 |A| + |B|) (bvh_step), 255 |A| is at most the extent of a node's box and
    // |B| at most |corner| + |origin|, each times |slope| -- so four times that bound is at most
    // 2^-19 (3 Cmax + |o|) |slope| with Cmax the largest |coordinate| of the root box, and moving
    // the origin end by 2^-19 (3 Cmax + |o|) moves the plane distance by exactly that much the safe way.
    const float P = 0x1p-22f * omax + 0x1p-19f * (3.0f * S.bvh_cmax + omax) + 1e-30f;
    const float big = 1.0e18f; // an axis the ray does not move along: a huge finite slope keeps infinities out of the slabs
    // 1 / d from the f32 reciprocal instruction (1 ulp) of the rounded direction: 1.5 * 2^-23 in
    // all, well inside the margin that went into P (a quarter of it bounds the roundings)
    T.ivx = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(float(d.x)), -big, big);
    T.ivy = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(float(d.y)), -big, big);
    T.ivz = __builtin_amdgcn_fmed3f(__builtin_amdgcn_rcpf(float(d.z)), -big, big);
    // (plane - origin) * slope gets smaller with the upper end of the interval when the slope is
    // positive, with the lower end when it is negative
    T.onx = T.ivx >= 0.0f ? fx + P : fx - P, T.ofx = T.ivx >= 0.0f ? fx - P : fx + P;
    T.ony = T.ivy >= 0.0f ? fy + P : fy - P, T.ofy = T.ivy >= 0.0f ? fy - P : fy + P;
    T.onz = T.ivz >= 0.0f ? fz + P : fz - P, T.ofz = T.ivz >= 0.0f ? fz - P : fz + P;
}

// triangle record -> the 12 scalars of the exact test (the three cofactors that do not involve
// the ray are recomputed with the operations the host uses for the all-triangles table)
__device__ __forceinline__ void finish_tri(double *t)
{
    double t0 = t[4] * t[8], t1 = t[7] * t[5];
    t[9] = t0 - t1;
    t0 = t[5] * t[6], t1 = t[8] * t[3];
    t[10] = t0 - t1;
    t0 = t[3] * t[7], t1 = t[6] * t[4];
    t[11] = t0 - t1;
}
__device__ __forceinline__ void finish_tri(float *t)
{
    float t0 = t[4] * t[8], t1 = t[7] * t[5];
    t[9] = t0 - t1;
    t0 = t[5] * t[6], t1 = t[8] * t[3];
    t[10] = t0 - t1;
    t0 = t[3] * t[7], t1 = t[6] * t[4];
    t[11] = t0 - t1;
}
// Triangle record of the BVH: the nine vertex coordinates as the floats they are (the reference's
// positions are float-rounded, src/geometry/cMeshLoader.cpp:138-146) and the face id -- 48 bytes,
// three 16-byte loads, one format for both precisions.  p1 - p2 and p1 - p3 are formed here in R,
// with the very operations the host uses for the all-triangles table.
template <class R> __device__ __forceinline__ int load_tri(const uint4 *rec, R *tri)
{
    const uint4 q0 = rec[0], q1 = rec[1], q2 = rec[2];
    const R p1x = R(__uint_as_float(q0.x)), p1y = R(__uint_as_float(q0.y)), p1z = R(__uint_as_float(q0.z));
    tri[0] = p1x, tri[1] = p1y, tri[2] = p1z;
    tri[3] = p1x - R(__uint_as_float(q0.w)), tri[4] = p1y - R(__uint_as_float(q1.x)), tri[5] = p1z - R(__uint_as_float(q1.y));
    tri[6] = p1x - R(__uint_as_float(q1.z)), tri[7] = p1y - R(__uint_as_float(q1.w)), tri[8] = p1z - R(__uint_as_float(q2.x));
    finish_tri(tri);
    return int(q2.y);
}
constexpr int TRI_REC_Q = 3; // 16-byte words of a triangle record

// The two homes of a stack level, addressed through explicit address spaces: left generic, the
// compiler merges the two stores of a push into one flat store behind a pointer select.
typedef __attribute__((address_space(3))) int lds_int_t;
typedef __attribute__((address_space(1))) uint32_t global_u32_t;
typedef uint32_t u32x4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) u32x4_t lds_u32x4_t;

// copies the first S.lds_nodes node records behind the workgroup's stack levels (all threads; the
// caller's barrier follows)
template <class R> __device__ __forceinline__ void stage_hot_nodes(const DScene<R> &S)
{
    uint4 *dst = reinterpret_cast<uint4 *>(dyn_stack + S.stack_levels * BLOCK);
    for (int k = threadIdx.x; k < S.lds_nodes * 4; k += BLOCK)
        dst[k] = S.bvh_nodes[k];
}

// One record visit.  ANY = false: closest hit (winner = smallest distance beyond self_eps, ties ->
// lowest face id; T.target = triangle, o + T.best_t * d = point).  ANY = true: is the segment of length T.best
// blocked (T.target = 1 and the traversal ends at the first blocker).  ANY = 2: which of the two, per lane (T.any).  The lane's stack is
// stk[level * BLOCK] in LDS.  Returns false once the traversal is complete.
// KIND = 1 / 2: the caller has established that the lane is at a node / a triangle record (the
// tracer advances the two kinds in separate wave steps so that neither runs on half-empty lanes).
template <class R, int ANY, int KIND = 0>
__device__ __forceinline__ bool bvh_step(const DScene<R> &S, V3<R> o, V3<R> d, Trav<R> &T, int *stk, uint32_t *visits)
{
    constexpr float K = 0x1p-20f;
    const R self_eps = Consts<R>::self_eps;
    const int id_bits = S.id_bits;
    uint32_t next = 0xFFFFFFFFu;
    if (KIND == 2 || (KIND == 0 && T.cur >= S.n_nodes))
    {
        visits[1]++;
        R tri[12], t;
        const int face = load_tri(S.bvh_tris + (size_t)(T.cur - S.n_nodes) * TRI_REC_Q, tri);
        if (tri_test(tri, o, d, t))
        {
            if (ANY == 1 || (ANY == 2 && T.any))
            {
                if (t < T.best - self_eps && t > self_eps)
                {
                    T.target = 1;
                    return false;
                }
            }
            else
            {
                V3<R> p = o + t * d;
                R dist = norm(p - o);
                if (dist > self_eps && (dist < T.best || (dist == T.best && face < T.best_face)))
                {
                    T.best = dist;
                    T.best_face = face;
                    T.target = T.cur - S.n_nodes;
                    T.best_t = t;
                }
            }
        }
    }
    else
    {
        visits[0]++;
        // 64-byte node: corner and cell size of the 8-bit grid its child boxes are quantised to,
        // six words of grid coordinates (low and high plane per axis, one byte per child), four ids
        // the busiest nodes (lowest ids) come from the workgroup's LDS copy: a compute unit's vector
        // cache serves about one scattered 16-byte lane access per cycle, and the top of the tree is
        // a third of all node visits
        u32x4_t n0, n1, n2, n3;
        if (T.cur < S.lds_nodes)
        {
            const lds_u32x4_t *rec = (const lds_u32x4_t *)(dyn_stack + S.stack_levels * BLOCK) + T.cur * 4;
            n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
        }
        else
        {
            const u32x4_t *rec = reinterpret_cast<const u32x4_t *>(S.bvh_nodes) + (size_t)T.cur * 4;
            n0 = rec[0], n1 = rec[1], n2 = rec[2], n3 = rec[3];
        }
        const float tmaxf = float(T.best) * (1.0f + K) + 1e-30f;
        // plane distance t = A * q + B per axis, A = cell * slope, B = (corner - origin) * slope, with
        // the origin end and the plane (low or high) that make a near plane nearer and a far plane
        // farther; the roundings of A, B and of the sum are bounded by 2^-21 (255 |A| + |B|)
        const float Ax = __uint_as_float(n1.x) * T.ivx, Ay = __uint_as_float(n1.y) * T.ivy, Az = __uint_as_float(n1.z) * T.ivz;
        // (the margin for the roundings of A, B and of the sum is part of the origin interval, see trav_init)
        const float Bnx = (__uint_as_float(n0.x) - T.onx) * T.ivx, Bfx = (__uint_as_float(n0.x) - T.ofx) * T.ivx;
        const float Bny = (__uint_as_float(n0.y) - T.ony) * T.ivy, Bfy = (__uint_as_float(n0.y) - T.ofy) * T.ivy;
        const float Bnz = (__uint_as_float(n0.z) - T.onz) * T.ivz, Bfz = (__uint_as_float(n0.z) - T.ofz) * T.ivz;
        const bool px = T.ivx >= 0.0f, py = T.ivy >= 0.0f, pz = T.ivz >= 0.0f;
        const uint32_t qnx = px ? n2.x : n2.y, qfx = px ? n2.y : n2.x; // near / far plane coordinates of the four children
        const uint32_t qny = py ? n2.z : n2.w, qfy = py ? n2.w : n2.z;
        const uint32_t qnz = pz ? n3.x : n3.y, qfz = pz ? n3.y : n3.x;
        // entry: the bits of the entry distance (non-negative: the sign bit is clear) above the record id --
        // truncated, hence still a lower bound; one and-or per child
        const uint32_t key_dist_mask = ~((1u << id_bits) - 1u);
        auto slab = [&](int i, uint32_t child) {
            const float tnx = fmaf(Ax, float((qnx >> (8 * i)) & 255u), Bnx), tfx = fmaf(Ax, float((qfx >> (8 * i)) & 255u), Bfx);
            const float tny = fmaf(Ay, float((qny >> (8 * i)) & 255u), Bny), tfy = fmaf(Ay, float((qfy >> (8 * i)) & 255u), Bfy);
            const float tnz = fmaf(Az, float((qnz >> (8 * i)) & 255u), Bnz), tfz = fmaf(Az, float((qfz >> (8 * i)) & 255u), Bfz);
            // tn: a lower bound of the true entry distance, tf: an upper bound of the exit distance
            const float tn = fmaxf(fmaxf(fmaxf(tnx, tny), tnz), 0.0f);
            const float tf = fminf(fminf(tfx, tfy), tfz);
            // (a slot without a child carries the id 0xFFFFFFFF: or-ing it in yields the "no entry" key by
            // itself, whatever the slab test says -- no compare for it; compares cost two f32 multiplies here)
            const bool hit = (tn <= tf) & (tn <= tmaxf);
            return hit ? ((__float_as_uint(tn) & key_dist_mask) | child) : 0xFFFFFFFFu;
        };
        uint32_t k0 = slab(0, n0.w);
        uint32_t k1 = slab(1, n1.w);
        uint32_t k2 = slab(2, n3.z);
        uint32_t k3 = slab(3, n3.w);
        auto order = [](uint32_t &a, uint32_t &b) {
            const uint32_t lo = a < b ? a : b, hi = a < b ? b : a;
            a = lo, b = hi;
        };
        order(k0, k1), order(k2, k3), order(k0, k2), order(k1, k3), order(k1, k2);
        // nearest child next, the others on the stack, farthest first.  The first S.stack_levels
        // levels live in LDS; the host sizes them so that deeper ones (a global array the lane
        // owns a column of) are the exception, and has checked the tree against the total
        if (k1 != 0xFFFFFFFFu)
        {
            if (T.sp + 3 <= S.stack_levels)
            {
                // the usual case: all of it fits the LDS levels
                lds_int_t *top = (lds_int_t *)(stk + T.sp * BLOCK);
                int n = 0;
                if (k3 != 0xFFFFFFFFu)
                    top[(n++) * BLOCK] = int(k3);
                if (k2 != 0xFFFFFFFFu)
                    top[(n++) * BLOCK] = int(k2);
                top[(n++) * BLOCK] = int(k1);
                T.sp += n;
            }
            else
            {
                auto push = [&](uint32_t e) {
                    if (T.sp < S.stack_levels)
                        *(lds_int_t *)(stk + T.sp * BLOCK) = int(e);
                    else
                        *(global_u32_t *)(S.spill + (uint32_t(T.sp - S.stack_levels) * uint32_t(S.spill_stride) + (blockIdx.x * BLOCK + threadIdx.x))) = e;
                    T.sp++;
                };
                if (k3 != 0xFFFFFFFFu)
                    push(k3);
                if (k2 != 0xFFFFFFFFu)
                    push(k2);
                push(k1);
            }
        }
        next = k0;
    }
    const uint32_t id_mask = (1u << id_bits) - 1u;
    if (next != 0xFFFFFFFFu)
    {
        T.cur = int(next & id_mask);
        return true;
    }
    const float tmaxf = float(T.best) * (1.0f + K) + 1e-30f;
    while (T.sp > 0)
    {
        --T.sp;
        uint32_t e;
        if (T.sp < S.stack_levels)
            e = uint32_t(*(lds_int_t *)(stk + T.sp * BLOCK));
        else
            e = *(global_u32_t *)(S.spill + (uint32_t(T.sp - S.stack_levels) * uint32_t(S.spill_stride) + (blockIdx.x * BLOCK + threadIdx.x)));
        if (__uint_as_float(e & ~id_mask) <= tmaxf)
        {
            T.cur = int(e & id_mask);
            return true;
        }
    }
    return false;
}

// closest hit.  Returns the intersector-order triangle index (or -1) and the hit point.
// Winner = smallest |p - o| among accepted hits with |p - o| > self_eps; ties -> lowest face id.
// The culling side (plane records, box slabs) measures along `du`, the direction at unit length,
// so that its bounds are in the units of the winner's distance |p - o|; the exact test runs on the
// ray (o, d) as given, whatever the length of d (the tracer's rays are unit length: du = d).
template <class R, bool BRUTE>
__device__ __forceinline__ int closest_hit(const DScene<R> &S, V3<R> o, V3<R> d, V3<R> du, V3<R> &pt, int *stk, int skip = -1,
                                           uint32_t *n_exact = nullptr) // n_exact (diagnostic build): exact tests this lane ran
{
    R best = Consts<R>::big;
    int target = -1;
    const R self_eps = Consts<R>::self_eps;
    if (BRUTE)
    {
        const BruteLds<R> *L = reinterpret_cast<const BruteLds<R> *>(stk);
        int near_rec;
        unsigned long long mask =
            prefilter_mask<true>(S.tri_pq, S.n_recs, S.pq_smax, o, du, float(self_eps) * 0.9f, __builtin_inff(), 0xFFFFFFFFu, near_rec);
        mask = drop_own_plane(mask, L, skip, du);
        auto exact = [&](int bit) {
            if (n_exact)
                ++*n_exact;
            const int f = L->rec_tri[bit];
            R t;
            if (f >= 0 && tri_test(L->tri + f * 12, o, d, t))
            {
                V3<R> p = o + t * d;
                R dist = norm(p - o);
                // intersector order = face order here, so the triangle index breaks ties
                if (dist > self_eps && (dist < best || (dist == best && f < target)))
                {
                    best = dist;
                    target = f;
                    pt = p;
                }
            }
        };
        // 1. the record that looks nearest in f32 goes first ...
        if (near_rec >= 0)
        {
            unsigned long long first = mask & (3ull << (2 * near_rec));
            mask &= ~first;
            while (first)
            {
                exact(__ffsll((long long)first) - 1);
                first &= first - 1;
            }
        }
        // 2. ... then every other candidate whose f32 distance interval lies wholly beyond that hit
        // is dropped (it can neither win nor tie); the intervals are recomputed from the LDS copy
        // of the plane normals, with the error bound of the prefilter
        if (target >= 0)
        {
            const float ox = float(o.x), oy = float(o.y), oz = float(o.z), dx = float(du.x), dy = float(du.y), dz = float(du.z);
            const float Tmax = 1.75f * fmaxf(fmaxf(fabsf(ox), fabsf(oy)), fabsf(oz)) + S.pq_smax;
            const float bound = float(best) * 1.000001f + 1e-30f;
            unsigned long long rest = mask;
            while (rest)
            {
                const int bit = __ffsll((long long)rest) - 1;
                rest &= rest - 1;
                const float *N = L->rec_n + (bit >> 1) * 4;
                const float nd = N[0] * dx + N[1] * dy + N[2] * dz;
                const float no = N[3] - (N[0] * ox + N[1] * oy + N[2] * oz);
                const float rr = __builtin_amdgcn_rcpf(nd);
                const float tlo = no * rr - PQ_ET * (Tmax * (rr * rr));
                if (tlo > bound) // false on NaN: the candidate stays
                    mask &= ~(1ull << bit);
            }
        }
        // 3. what is left gets the exact test
        while (mask)
        {
            exact(__ffsll((long long)mask) - 1);
            mask &= mask - 1;
        }
        return target;
    }
    Trav<R> T;
    uint32_t visits[2] = {0, 0};
    trav_init(S, T, o, du); // slab distances along the unit direction: same units as T.best
    while (bvh_step<R, false>(S, o, d, T, stk, visits))
    {
    }
    pt = o + T.best_t * d;
    return T.target;
}

// segment visibility p1 -> p2: blocked iff some triangle has self_eps < t < |p1 - p2| - self_eps
template <class R, bool BRUTE>
__device__ __forceinline__ bool visible(const DScene<R> &S, V3<R> p1, V3<R> p2, int *stk, int skip = -1, uint32_t rec_mask = 0xFFFFFFFFu,
                                        uint32_t *visits = nullptr)
{
    const R self_eps = Consts<R>::self_eps;
    R dist = norm(p1 - p2);
    V3<R> dir = normalized(p2 - p1);
    if (BRUTE)
    {
        const BruteLds<R> *L = reinterpret_cast<const BruteLds<R> *>(stk);
        // blockers have self_eps < t < dist - self_eps: drop what surely lies outside (0.9 self_eps, dist)
        const float far = float(dist) * 1.00001f + 1e-30f;
        int unused_rec;
        unsigned long long mask = prefilter_mask<false>(S.tri_pq, S.n_recs, S.pq_smax, p1, dir, float(self_eps) * 0.9f, far, rec_mask, unused_rec);
        // p2 lies on triangle `skip`: as seen from p1 its plane is met at distance ~dist, which is
        // not a blocker either
        mask = drop_own_plane(mask, L, skip, dir);
        bool blocked = false;
        while (mask && !blocked)
        {
            const int bit = __ffsll((long long)mask) - 1;
            mask &= mask - 1;
            const int f = L->rec_tri[bit];
            R t;
            if (f >= 0 && tri_test(L->tri + f * 12, p1, dir, t) && t < dist - self_eps && t > self_eps)
                blocked = true;
        }
        return !blocked;
    }
    Trav<R> T;
    trav_init(S, T, p1, dir, dist);
    uint32_t none[2] = {0, 0};
    while (bvh_step<R, true>(S, p1, dir, T, stk, visits ? visits : none))
    {
    }
    return T.target < 0;
}

// ---------------------------------------------------------------------------------------------
// camera rays
// ---------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ void primary_ray(const DCamera &cam, int x, int y, V3<R> &o, V3<R> &d)
{
    // always in double: the matrix comes from a float view matrix but is applied in double
    double v0 = double(x), v1 = double(y);
    double p[3];
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        p[i] = PTX_SUM4(cam.m[i * 4 + 0] * v0, cam.m[i * 4 + 1] * v1, cam.m[i * 4 + 2] * 1.0, cam.m[i * 4 + 3] * 1.0);
    }
    V3<double> g = {p[0] - cam.pos[0], p[1] - cam.pos[1], p[2] - cam.pos[2]};
    V3<double> n = normalized(g);
    o = {R(cam.pos[0]), R(cam.pos[1]), R(cam.pos[2])};
    d = {R(n.x), R(n.y), R(n.z)};
}

// writes the camera rays of the whole image to host-visible buffers (test / operator form)
__global__ void __launch_bounds__(BLOCK) k_camera_rays(DCamera cam, double *__restrict__ ro, double *__restrict__ rd)
{
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    int64_t n = (int64_t)cam.width * cam.height;
    if (i >= n)
        return;
    V3<double> o, d;
    primary_ray<double>(cam, int(i % cam.width), int(i / cam.width), o, d);
    ro[i * 3] = o.x, ro[i * 3 + 1] = o.y, ro[i * 3 + 2] = o.z;
    rd[i * 3] = d.x, rd[i * 3 + 1] = d.y, rd[i * 3 + 2] = d.z;
}

// closest hit of arbitrary rays (AoS double in, face id + point out).  from_camera: generate the
// ray of pixel i instead of reading it.
template <class R, bool BRUTE>
__global__ void __launch_bounds__(BLOCK)
    k_raycast(DScene<R> S, DCamera cam, int from_camera, const double *__restrict__ ro, const double *__restrict__ rd,
              int64_t n, int32_t *__restrict__ face_ids, double *__restrict__ pts)
{
    __shared__ __align__(16) int lds_raw[BRUTE ? LDS_WORDS_BRUTE : 1];
    int *stk = intersector_scratch<R, BRUTE>(S, BRUTE ? lds_raw : dyn_stack); // all threads: contains a barrier
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n)
        return;
    V3<R> o, d;
    if (from_camera)
        primary_ray<R>(cam, int(i % cam.width), int(i / cam.width), o, d);
    else
    {
        o = {R(ro[i * 3]), R(ro[i * 3 + 1]), R(ro[i * 3 + 2])};
        d = {R(rd[i * 3]), R(rd[i * 3 + 1]), R(rd[i * 3 + 2])};
    }
    V3<R> p = {R(0), R(0), R(0)};
    // cAccelStruct::RayCast takes a direction of any length (tRay::Init does not normalise)
    int k = closest_hit<R, BRUTE>(S, o, d, from_camera ? d : normalized(d), p, stk);
    face_ids[i] = k < 0 ? -1 : S.tri_face[k];
    if (pts)
    {
        const double qn = __builtin_nan("");
        pts[i * 3] = k < 0 ? qn : double(p.x);
        pts[i * 3 + 1] = k < 0 ? qn : double(p.y);
        pts[i * 3 + 2] = k < 0 ? qn : double(p.z);
    }
}

template <class R, bool BRUTE>
__global__ void __launch_bounds__(BLOCK)
    k_vistest(DScene<R> S, const double *__restrict__ p1, const double *__restrict__ p2, int64_t n, int32_t *__restrict__ vis)
{
    __shared__ __align__(16) int lds_raw[BRUTE ? LDS_WORDS_BRUTE : 1];
    int *stk = intersector_scratch<R, BRUTE>(S, BRUTE ? lds_raw : dyn_stack); // all threads: contains a barrier
    int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n)
        return;
    V3<R> a = {R(p1[i * 3]), R(p1[i * 3 + 1]), R(p1[i * 3 + 2])};
    V3<R> b = {R(p2[i * 3]), R(p2[i * 3 + 1]), R(p2[i * 3 + 2])};
    vis[i] = visible<R, BRUTE>(S, a, b, stk) ? 1 : 0;
}

// nine consecutive floats (three vertices or vertex normals of a face; 4-byte aligned) in three loads
struct __attribute__((packed, aligned(4))) Float9
{
    float v[9];
};
__device__ __forceinline__ Float9 load9(const float *p) { return *reinterpret_cast<const Float9 *>(p); }

// one (direct rgb, coefficient rgb) record of a path: 64-byte aligned in f64 (three 16-byte loads),
// 32-byte aligned in f32 (a 16-byte and an 8-byte load)
__device__ __forceinline__ void load_record(const double *rec, double *q)
{
    const uint4 *r = reinterpret_cast<const uint4 *>(rec);
    const uint4 a = r[0], b = r[1], c = r[2];
    q[0] = __hiloint2double(int(a.y), int(a.x)), q[1] = __hiloint2double(int(a.w), int(a.z));
    q[2] = __hiloint2double(int(b.y), int(b.x)), q[3] = __hiloint2double(int(b.w), int(b.z));
    q[4] = __hiloint2double(int(c.y), int(c.x)), q[5] = __hiloint2double(int(c.w), int(c.z));
}
__device__ __forceinline__ void load_record(const float *rec, float *q)
{
    const float4 a = *reinterpret_cast<const float4 *>(rec);
    const float2 b = *reinterpret_cast<const float2 *>(rec + 4);
    q[0] = a.x, q[1] = a.y, q[2] = a.z, q[3] = a.w, q[4] = b.x, q[5] = b.y;
}
// the store side: 48 (24) bytes in 3 (2) stores
__device__ __forceinline__ void store_record(double *rec, V3<double> direct, V3<double> coef)
{
    double2 *r = reinterpret_cast<double2 *>(rec);
    r[0] = double2{direct.x, direct.y}, r[1] = double2{direct.z, coef.x}, r[2] = double2{coef.y, coef.z};
}
__device__ __forceinline__ void store_record(float *rec, V3<float> direct, V3<float> coef)
{
    *reinterpret_cast<float4 *>(rec) = float4{direct.x, direct.y, direct.z, coef.x};
    *reinterpret_cast<float2 *>(rec + 4) = float2{coef.y, coef.z};
}

__device__ __forceinline__ int global_wave() { return int((blockIdx.x * BLOCK + threadIdx.x) >> 6); }
__device__ __forceinline__ int lane_id() { return int(threadIdx.x & 63); }

// Small tables -- materials, lights, per-triangle normals and material ids -- are read from LDS
// copies when they fit.  TAB: materials and lights fit; NRM: the per-triangle tables fit too.
constexpr int LDS_MATS = 32, LDS_LIGHTS = 32, LDS_TRIS = 64;

// ---------------------------------------------------------------------------------------------
// persistent tracer: every LANE carries one path in registers from its camera ray to its end --
// closest hit, termination rules, shading normal, next direction, NEE with its visibility tests --
// and then starts the next path of its wave's range (path regeneration).  No ray ever goes through
// memory: the only global traffic is one 6-scalar stack record per shaded segment and one state
// word per path.  Path ids come from one pool counter per batch: a wave whose idle lanes find its
// range empty takes the next `chunk` ids with a single atomic add, so the grid is just large
// enough to fill the GPU and all waves run dry together.  Every wave leaves the loop once the pool
// is used up and its 64 lanes are idle.
// ---------------------------------------------------------------------------------------------
// 64 prepared camera rays of one wave (direction and RNG key; the origin is the camera position)
template <class R> struct RayBlock
{
    R d[3][64];
    uint64_t key[64];
    uint32_t sample[64]; // sample index of the path (its id divided by the pixels of the batch: one division on full lanes)
};

// Pending visibility tests of one wave (ring buffer in LDS).  Only part of the lanes have a valid
// light sample in a given bounce; instead of running the visibility test with those lanes alone,
// they park the segment here and the wave tests 64 parked segments at a time on full lanes.  The
// light term itself is already in the bounce record; a blocked segment zeroes it afterwards.
template <class R> struct ShadowQueue
{
    static constexpr int CAP = 128; // at most 63 left over + 64 new
    R xi[2][CAP];                   // the two draws the light sample point is a function of
    R p2[3][CAP];                   // shaded point
    uint32_t own_bounce[CAP];       // (triangle the shaded point lies on << 16) | bounce
    uint8_t slot[CAP];              // own_bounce, slot: which bounce record the result belongs to
};
struct NoQueue
{
};

// Finished paths of one wave waiting for their resolve pass (ring buffer in LDS): path id,
// (record count << 16) | (tail material + 1), the record slot, and how many visibility tests the
// wave had parked when the path ended (a wrapping 16-bit count: the queues are far shorter) --
// once that many have been tested, no test of this path is pending.
struct DoneQueue
{
    static constexpr int CAP = 128;
    uint32_t pid[CAP], state[CAP];
    uint16_t need[CAP], slot[CAP];
};

// LIGHTS selects how visibility tests are compiled: 0 -- the scene has no square light, there are
// none; 1 -- exactly one light and it is a square light: every bounce has at most one test, which
// goes through the wave's ShadowQueue (all-triangles intersector only); 2 -- tests run in place.
//
// With the BVH the closest-hit search is a per-lane loop of very uneven length (on the 336 k
// triangle room: 21 records on average, several times that for the slowest lane of a wave), so it
// is not run to completion inside one bounce iteration.  Every lane keeps a resumable traversal;
// the wave advances all unfinished traversals one record at a time and only turns to shading --
// and to starting new paths -- when enough lanes are waiting for it.  Lanes with a long traversal
// simply keep going while the others shade, start their next ray and catch up.  (BVH scenes
// without square lights use the ring tracer of bvh_tracer.hpp instead.)
//
// FAST: the physically based estimator (device_math.hpp, "fast") on the same machinery.  A lane
// also carries its path's throughput and the density its current direction was sampled with; the
// record of a bounce holds (light-sampling term, everything else the bounce collects: conf lights
// its ray crossed, an emitter the camera sees), both already multiplied by the throughput, every
// segment writes one, a blocked visibility test still only zeroes the first half, and the
// resolve pass is a plain sum.  Every light needs its visibility test here, sphere lights too.
// Diagnostic build (make PHASE=1): every wave adds up the shader cycles it spends
// in each part of its loop (tracers of kernels.hpp and bvh_tracer.hpp), and a few waves print their totals when they finish.
#ifdef PTX_PHASE_TIMING
#define PTX_PHASE_END(k)                                                   \
    do                                                                     \
    {                                                                      \
        const unsigned long long now_ = __builtin_readcyclecounter();      \
        ph_t[k] += now_ - ph_last, ph_n[k]++;                              \
        ph_last = now_;                                                    \
    } while (0)
// lane occupancy of a part: `work` of this lane (0 / 1, or a trip count) against the longest of the wave
#define PTX_PHASE_LANES(k, work)                                                         \
    do                                                                                   \
    {                                                                                    \
        uint32_t w_ = uint32_t(work), mx_ = w_, sm_ = w_;                                \
        for (int off_ = 32; off_ > 0; off_ >>= 1)                                        \
        {                                                                                \
            mx_ = max(mx_, uint32_t(__shfl_xor(int(mx_), off_, 64)));                    \
            sm_ += uint32_t(__shfl_xor(int(sm_), off_, 64));                             \
        }                                                                                \
        ph_lane_work[k] += sm_, ph_lane_room[k] += 64ull * mx_;                          \
    } while (0)
#else
#define PTX_PHASE_END(k) \
    do                   \
    {                    \
    } while (0)
#define PTX_PHASE_LANES(k, work) \
    do                           \
    {                            \
    } while (0)
#endif
template <class R, bool BRUTE, bool TAB, int LIGHTS, bool FAST = false>
__device__ __forceinline__ void trace_body(const DScene<R> &S, const DCamera &cam, const DBatch &B, const DState<R> &st, int chunk)
{
    constexpr bool NRM = BRUTE; // all-triangles intersector = at most 64 triangles: the per-triangle tables fit LDS
    constexpr bool SQ1 = LIGHTS == 1;
    __shared__ __align__(16) int lds_raw[BRUTE ? LDS_WORDS_BRUTE : 1];
    __shared__ R lds_mats[TAB ? LDS_MATS * MAT_STRIDE : 1];
    __shared__ R lds_lights[TAB ? (SQ1 ? 1 : LDS_LIGHTS) * LIGHT_STRIDE : 1];
    __shared__ R lds_nrm[NRM ? LDS_TRIS * 4 : 1];
    __shared__ int lds_tmat[NRM ? LDS_TRIS : 1];
    __shared__ RayBlock<R> ray_blocks[BLOCK / 64];
    __shared__ typename std::conditional<SQ1, ShadowQueue<R>, NoQueue>::type shadow_queues[BLOCK / 64];
    __shared__ DoneQueue done_queues[BLOCK / 64];
    __shared__ uint8_t free_slots[BLOCK / 64][LANE_SLOTS]; // per wave: the record slots no path holds (a stack)
    for (int k = threadIdx.x; k < (BLOCK / 64) * LANE_SLOTS; k += BLOCK)
        free_slots[k / LANE_SLOTS][k % LANE_SLOTS] = uint8_t(k % LANE_SLOTS);
    if (TAB)
    {
        for (int k = threadIdx.x; k < S.n_mats * MAT_STRIDE; k += BLOCK)
            lds_mats[k] = S.mats[k];
        for (int k = threadIdx.x; k < S.n_lights * LIGHT_STRIDE; k += BLOCK)
            lds_lights[k] = S.lights[k];
    }
    if (NRM)
    {
        for (int k = threadIdx.x; k < S.n_tris * 4; k += BLOCK)
            lds_nrm[k] = S.tri_nrm[k];
        for (int k = threadIdx.x; k < S.n_tris; k += BLOCK)
            lds_tmat[k] = S.tri_mat[k];
    }
    int *stk = intersector_scratch<R, BRUTE>(S, BRUTE ? lds_raw : dyn_stack);
    if constexpr (!BRUTE)
        stage_hot_nodes(S);
    __syncthreads(); // the only barrier: from here on the waves of the workgroup are independent
    const int g = global_wave(), lane = lane_id();
    if (g >= st.n_waves)
        return;
    const unsigned long long lt_mask = (1ull << lane) - 1ull;
    const uint32_t n_paths = uint32_t(B.n_paths);
    uint32_t next = 0, end = 0; // the wave's current range of path ids
    bool drained = false;       // the pool has no ids left (wave-uniform)
    const uint32_t n_pix = uint32_t(B.n_pix), n_cols = uint32_t(B.n_cols);
    const int depth = B.depth;
    // camera rays are prepared 64 at a time by the whole wave (full lanes) and handed to idle lanes
    // from this block: a lane that starts a new path only copies its ray from LDS
    RayBlock<R> &blk = ray_blocks[threadIdx.x >> 6];
    uint32_t blk_base = 0, blk_pos = 0, blk_cnt = 0; // wave-uniform: first path id, consumed, size
    const V3<R> cam_pos = {R(cam.pos[0]), R(cam.pos[1]), R(cam.pos[2])};
    auto &sq = shadow_queues[threadIdx.x >> 6];
    uint32_t sq_head = 0, sq_count = 0; // wave-uniform
    uint32_t sq_pushed = 0, sq_drained = 0; // running totals (wave-uniform, wrap around)
    DoneQueue &dq = done_queues[threadIdx.x >> 6];
    uint32_t dq_head = 0, dq_count = 0; // wave-uniform
    uint8_t *free_slot = free_slots[threadIdx.x >> 6];
    uint32_t free_count = LANE_SLOTS;   // wave-uniform; never runs out: 64 paths on the lanes + at most 127 in dq
    uint32_t slot = 0;                  // the record slot of this lane's path

    bool alive = false;
    bool trav_done = false; // BVH: the lane's closest-hit traversal is complete (its path waits for shading)
    Trav<R> T;
    trav_init(S, T, V3<R>{R(0), R(0), R(0)}, V3<R>{R(1), R(1), R(1)});
    V3<R> o = {R(0), R(0), R(0)}, d = {R(0), R(0), R(0)};
    uint32_t pid = 0, sample = 0;
    int bounce = 0, prim = -1;
    uint64_t key = 0;
    uint32_t n_seg = 0, n_tests = 0, n_iter = 0, n_rec = 0;
    uint32_t n_visit[2] = {0, 0}; // BVH node / triangle records visited by this lane
    // FAST: throughput of the path so far, and the density (solid angle) its current direction was
    // sampled with -- -1 after a delta lobe, -2 while only delta lobes lie between it and the camera
    V3<R> thr = {R(1), R(1), R(1)};
    R pdf_prev = R(-2);

    // tests the first n (<= 64) parked segments, one per lane, and removes the direct light of
    // the blocked ones from their bounce records
#ifdef PTX_PHASE_TIMING
    unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
    uint32_t ph_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    // lane work done / lane slots offered (64 x the longest lane) in: 0 path starts, 1 closest hit (lanes alive),
    // 2 shading, 3 visibility passes, 4 resolve folds, 5 exact triangle tests
    unsigned long long ph_lane_work[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_lane_room[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif
    auto drain = [&](uint32_t n) {
        PTX_PHASE_END(2);
        if constexpr (SQ1)
        {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // records that can block any of this pass's segments: union over the lanes of the
            // blocker sets of the records their shaded points lie on
            const uint32_t e = (sq_head + uint32_t(lane)) % ShadowQueue<R>::CAP;
            uint32_t blockers = 0u;
            if constexpr (BRUTE)
            {
                const BruteLds<R> *T = reinterpret_cast<const BruteLds<R> *>(stk);
                if (uint32_t(lane) < n)
                    blockers = T->rec_block[T->tri_rec[sq.own_bounce[e] >> 16]];
                for (int off = 32; off > 0; off >>= 1)
                    blockers |= uint32_t(__shfl_xor(int(blockers), off, 64));
            }
            else
                blockers = 0xFFFFFFFFu;
            if (uint32_t(lane) < n)
            {
                const R *L0;
                if constexpr (TAB)
                    L0 = lds_lights;
                else
                    L0 = S.lights;
                const V3<R> b = {sq.p2[0][e], sq.p2[1][e], sq.p2[2][e]};
                V3<R> a; // the light point, recomputed from the draws: same bits as at shading time
                if constexpr (FAST)
                {
                    V3<R> unused;
                    (void)fast_light_sample(L0, b, sq.xi[0][e], sq.xi[1][e], unused, a);
                }
                else
                    a = square_light_point(L0, sq.xi[0][e], sq.xi[1][e]);
                const uint32_t ob = sq.own_bounce[e];
                if (!visible<R, BRUTE>(S, a, b, stk, int(ob >> 16), blockers))
                {
                    // same wave, later in program order than the store of the term: this one wins
                    R *rec = record_at(st, g, sq.slot[e], depth, ob & 0xffffu);
                    rec[0] = R(0), rec[1] = R(0), rec[2] = R(0);
                }
            }
            sq_head = (sq_head + n) % ShadowQueue<R>::CAP;
            sq_count -= n;
            sq_drained += n;
            PTX_PHASE_LANES(3, uint32_t(lane) < n ? 1u : 0u);
        }
        PTX_PHASE_END(3);
    };

    // folds the bounce records of the first n (<= 64) finished paths back to front with the
    // per-bounce max-clamp (cPathTracer.cpp:679-692), one path per lane, and stores the per-sample
    // results.  The records were written by this wave a few iterations ago, so they come from L2.
    auto resolve = [&](uint32_t n) {
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
        PTX_PHASE_END(5);
        // a parked visibility test may still have to zero the direct light of one of these paths
        if constexpr (SQ1)
        {
            const uint32_t need_n = uint32_t(__builtin_amdgcn_readfirstlane(int(dq.need[(dq_head + n - 1u) % DoneQueue::CAP])));
            while (int16_t(uint16_t(sq_drained) - uint16_t(need_n)) < 0)
                drain(sq_count < 64u ? sq_count : 64u);
        }
        // the records were stored by other lanes of this same wave: waiting for those stores is
        // enough, a compute unit's vector cache is coherent for its own waves (workgroup scope)
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        const uint32_t e = (dq_head + uint32_t(lane)) % DoneQueue::CAP;
        const bool mine = uint32_t(lane) < n;
        const uint32_t rp = mine ? dq.pid[e] : 0u, rs = mine ? dq.state[e] : 0u;
        const uint32_t rslot = mine ? uint32_t(dq.slot[e]) : 0u;
        const int len = int(rs >> 16), tm = int(rs & 0xffffu);
        R fr = R(0), fg = R(0), fb = R(0);
        if (tm)
        {
            const R *tmat;
            if constexpr (TAB)
                tmat = lds_mats + (tm - 1) * MAT_STRIDE;
            else
                tmat = S.mats + (size_t)(tm - 1) * MAT_STRIDE;
            fr = tmat[0], fg = tmat[1], fb = tmat[2];
        }
        int wlen = len; // longest stack of the wave: the loop below has a wave-uniform trip count
        for (int off = 32; off > 0; off >>= 1)
            wlen = max(wlen, __shfl_xor(wlen, off, 64));
        PTX_PHASE_LANES(4, len);
        const R *rec = record_at(st, g, rslot, depth, 0u);
        // four levels per trip: their loads are issued together, so the wave waits for the memory
        // system once per four bounces instead of once per bounce
        constexpr int G = 4;
        for (int top = wlen - 1; top >= 0; top -= G)
        {
            R q[G][6];
#pragma unroll
            for (int k = 0; k < G; k++)
            {
                const int i = top - k;
                const bool on = i >= 0 && i < len;
                q[k][0] = q[k][1] = q[k][2] = q[k][3] = q[k][4] = q[k][5] = R(0);
                if (on) // lanes whose stack is shorter do not touch memory
                    load_record(rec + i * REC_STRIDE, q[k]); // 48 (24) bytes in 3 (2) loads
            }
#pragma unroll
            for (int k = 0; k < G; k++)
            {
                const int i = top - k;
                if (i >= 0 && i < len)
                {
                    if constexpr (FAST)
                    {
                        // every term already carries its throughput: the path's radiance is their sum
                        fr += q[k][0] + q[k][3], fg += q[k][1] + q[k][4], fb += q[k][2] + q[k][5];
                        continue;
                    }
                    fr = q[k][0] + q[k][3] * fr;
                    fg = q[k][1] + q[k][4] * fg;
                    fb = q[k][2] + q[k][5] * fb;
                    R mx = fr;
                    if (fg > mx)
                        mx = fg;
                    if (fb > mx)
                        mx = fb;
                    if (R(0) > mx)
                        mx = R(0);
                    if (mx > R(1))
                    {
                        // one reciprocal instead of the reference's three divisions: results move by
                        // an ulp at most, like the rest of the shading arithmetic
                        const R inv = R(1) / mx;
                        fr *= inv, fg *= inv, fb *= inv;
                    }
                }
            }
        }
        if (mine)
        {
            st.result[rp] = fr, st.result[st.N + rp] = fg, st.result[2 * st.N + rp] = fb;
            free_slot[free_count + uint32_t(lane)] = uint8_t(rslot); // the slots go back on the stack
        }
        free_count += n;
        dq_head = (dq_head + n) % DoneQueue::CAP;
        dq_count -= n;
        // the next path to start reads the stack the lanes above just wrote
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        PTX_PHASE_END(4);
    };

    for (;;)
    {
        if constexpr (!BRUTE)
        {
            // ---- advance the unfinished traversals until enough lanes wait for shading / a new path --
            for (;;)
            {
                const int n_trav = __popcll(__ballot(alive && !trav_done));
                if (n_trav == 0)
                    break;
                const bool can_start = !(blk_pos == blk_cnt && next == end && drained);
                const int n_serve = __popcll(__ballot(alive && trav_done)) + (can_start ? __popcll(__ballot(!alive)) : 0);
                if (n_serve >= 32 || (n_serve > 0 && n_trav <= n_serve))
                    break;
                // nodes and triangles in separate wave steps (f32 slab tests / exact test in R), the
                // kind that serves more lanes per cycle first
                const bool act = alive && !trav_done;
                const bool at_tri = act && T.cur >= S.n_nodes;
                const int n_tri = __popcll(__ballot(at_tri));
                if (n_tri * 2 > (n_trav - n_tri) * 3)
                {
                    if (at_tri)
                        trav_done = !bvh_step<R, false, 2>(S, o, d, T, stk, n_visit);
                }
                else if (act && !at_tri)
                    trav_done = !bvh_step<R, false, 1>(S, o, d, T, stk, n_visit);
            }
        }
        // ---- idle lanes start the next paths ------------------------------------------------------
        const unsigned long long idle = __ballot(!alive);
        if (idle != 0ull && blk_pos == blk_cnt && !(next == end && drained))
        {
            if (next == end)
            {
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(st.pool, uint32_t(chunk));
                base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
                if (base >= n_paths)
                    drained = true;
                else
                    next = base, end = n_paths - base < uint32_t(chunk) ? n_paths : base + uint32_t(chunk);
            }
            blk_base = next, blk_pos = 0;
            blk_cnt = end - next < 64u ? end - next : 64u;
            next += blk_cnt;
            if (uint32_t(lane) < blk_cnt)
            {
                const uint32_t id = blk_base + uint32_t(lane);
                const uint32_t pix = id % n_pix;
                const int lr = B.row0 + int(pix / n_cols), col = B.col0 + int(pix % n_cols);
                const int row = local_row_to_image_row(B, lr);
                V3<R> ro, rd;
                primary_ray<R>(cam, col, row, ro, rd);
                blk.d[0][lane] = rd.x, blk.d[1][lane] = rd.y, blk.d[2][lane] = rd.z;
                blk.key[lane] = rng_pixel_key(B.seed, uint32_t(row) * uint32_t(B.image_width) + uint32_t(col));
                blk.sample[lane] = uint32_t(B.sample0) + id / n_pix;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        }
        {
            // the lanes that start a path now each take a record slot off the wave's stack
            const bool starts = !alive && blk_pos + uint32_t(__popcll(idle & lt_mask)) < blk_cnt;
            const unsigned long long sm = __ballot(starts);
            if (sm != 0ull)
                PTX_PHASE_LANES(0, starts ? 1u : 0u);
            if (starts)
                slot = free_slot[free_count - 1u - uint32_t(__popcll(sm & lt_mask))];
            free_count -= uint32_t(__popcll(sm));
        }
        if (!alive)
        {
            const uint32_t e = blk_pos + uint32_t(__popcll(idle & lt_mask));
            if (e < blk_cnt)
            {
                pid = blk_base + e;
                o = cam_pos;
                d = {blk.d[0][e], blk.d[1][e], blk.d[2][e]};
                key = blk.key[e];
                sample = blk.sample[e];
                bounce = 0, prim = -1;
                alive = true;
                if constexpr (FAST)
                    thr = {R(1), R(1), R(1)}, pdf_prev = R(-2);
                if constexpr (!BRUTE)
                    trav_init(S, T, o, d), trav_done = false;
            }
        }
        {
            const uint32_t taken = uint32_t(__popcll(idle));
            blk_pos = blk_cnt - blk_pos < taken ? blk_cnt : blk_pos + taken;
        }
        if (__ballot(alive) == 0ull)
            break; // pool used up and every lane idle
        n_iter++;
        PTX_PHASE_END(0);

        // ---- closest hit and the termination rules (miss / no material -> tail 0; emitter -> Ka) --
        bool shaded = false, finished = false; // finished: this lane's path ended in this iteration
        uint32_t fin_state = 0;
        V3<R> fast_other = {R(0), R(0), R(0)}; // FAST: what this segment collected besides light sampling
        V3<R> p = {R(0), R(0), R(0)};
        int hit_k = -1;
        const R *m = nullptr; // the address space (LDS or global) is fixed at compile time
#ifdef PTX_PHASE_TIMING
        uint32_t lane_exact = 0;
        PTX_PHASE_LANES(1, (BRUTE ? alive : (alive && trav_done)) ? 1u : 0u);
#endif
        if (BRUTE ? alive : (alive && trav_done))
        {
            n_seg++;
            if constexpr (BRUTE)
            {
#ifdef PTX_PHASE_TIMING
                hit_k = closest_hit<R, BRUTE>(S, o, d, d, p, stk, prim, &lane_exact);
#else
                hit_k = closest_hit<R, BRUTE>(S, o, d, d, p, stk, prim);
#endif
            }
            else
                hit_k = T.target, p = o + T.best_t * d;
            int mat_id = -1;
            if (hit_k >= 0)
                mat_id = NRM ? lds_tmat[hit_k] : S.tri_mat[hit_k];
            const int mrow = (mat_id < 0 ? 0 : mat_id) * MAT_STRIDE;
            if constexpr (TAB)
                m = lds_mats + mrow;
            else
                m = S.mats + mrow;
            alive = false;
            if constexpr (FAST)
            {
                // conf lights this ray crossed before its hit: emitters that do not occlude, seen by
                // sampled directions only (not by the camera or through delta lobes from it)
                V3<R> other = {R(0), R(0), R(0)};
                if (pdf_prev > R(-2))
                {
                    const R t_geom = hit_k >= 0 ? norm(p - o) : Consts<R>::big;
                    for (int l = 0; l < S.n_lights; l++)
                    {
                        const R *L;
                        if constexpr (TAB)
                            L = lds_lights + l * LIGHT_STRIDE;
                        else
                            L = S.lights + (size_t)l * LIGHT_STRIDE;
                        R tl;
                        const R pl = fast_light_pdf(L, o, d, tl);
                        if (pl > R(0) && tl < t_geom)
                        {
                            const R w = pdf_prev < R(0) ? R(1) : pdf_prev / (pdf_prev + pl);
                            other = other + V3<R>{thr.x * L[1] * w, thr.y * L[2] * w, thr.z * L[3] * w};
                        }
                    }
                }
                fast_other = other;
                if (mat_id < 0 || m[13] != R(0))
                {
                    // the path ends here; an emitter counts when the camera sees it
                    if (mat_id >= 0 && pdf_prev == R(-2))
                        other = other + V3<R>{thr.x * m[0], thr.y * m[1], thr.z * m[2]};
                    store_record(record_at(st, g, slot, depth, uint32_t(bounce)), V3<R>{R(0), R(0), R(0)}, other);
                    n_rec++;
                    finished = true, fin_state = uint32_t(bounce + 1) << 16;
                }
                else
                    shaded = true;
            }
            else if (mat_id < 0)
                finished = true, fin_state = uint32_t(bounce) << 16;
            else if (m[13] != R(0))
                finished = true, fin_state = (uint32_t(bounce) << 16) | uint32_t(mat_id + 1);
            else
                shaded = true;
        }

#ifdef PTX_PHASE_TIMING
        PTX_PHASE_LANES(5, lane_exact);
        PTX_PHASE_LANES(2, shaded ? 1u : 0u);
#endif
        PTX_PHASE_END(1);
        bool park = false; // this lane has a visibility test for the queue
        R park_x1 = R(0), park_x2 = R(0);
        const uint32_t seg_pid = pid, seg_bounce = uint32_t(bounce), seg_slot = slot;
        if constexpr (FAST)
        {
            if (shaded)
            {
                const V3<R> wo = -d;
                // ---- shading normal, as below ------------------------------------------------------------
                R fn0, fn1, fn2, fn3;
                if (NRM)
                    fn0 = lds_nrm[hit_k * 4], fn1 = lds_nrm[hit_k * 4 + 1], fn2 = lds_nrm[hit_k * 4 + 2], fn3 = lds_nrm[hit_k * 4 + 3];
                else
                {
                    const R *__restrict__ fn = S.tri_nrm + (size_t)hit_k * 4;
                    fn0 = fn[0], fn1 = fn[1], fn2 = fn[2], fn3 = fn[3];
                }
                V3<R> nrm = {fn0, fn1, fn2};
                if (fn3 == R(0))
                {
                    const int face = S.tri_face[hit_k];
                    const Float9 fv = load9(S.verts + (size_t)face * 9), fq = load9(S.normals + (size_t)face * 9);
                    V3<R> a = {R(fv.v[0]), R(fv.v[1]), R(fv.v[2])}, b = {R(fv.v[3]), R(fv.v[4]), R(fv.v[5])}, c = {R(fv.v[6]), R(fv.v[7]), R(fv.v[8])};
                    V3<R> w = barycentric(p, a, b, c);
                    V3<R> acc = w.x * V3<R>{R(fq.v[0]), R(fq.v[1]), R(fq.v[2])};
                    acc = acc + w.y * V3<R>{R(fq.v[3]), R(fq.v[4]), R(fq.v[5])};
                    acc = acc + w.z * V3<R>{R(fq.v[6]), R(fq.v[7]), R(fq.v[8])};
                    nrm = fast_normalized(acc);
                }
                const int D0 = 4 * S.n_lights;
                V3<R> direct = {R(0), R(0), R(0)}, wi = {R(0), R(0), R(0)};
                bool go_on = S.enable_indirect != 0;
                if (m[12] != R(0))
                {
                    // glass: a delta lobe, no light sampling
                    if (go_on)
                    {
                        wi = fast_glass_sample(m, nrm, wo, rng_draw<R>(key, sample, bounce, D0));
                        pdf_prev = pdf_prev == R(-2) ? R(-2) : R(-1);
                    }
                }
                else
                {
                    if (dot(nrm, wo) < R(0))
                        nrm = -nrm;
                    for (int l = 0; l < S.n_lights; l++)
                    {
                        const R *L;
                        if constexpr (TAB)
                            L = lds_lights + l * LIGHT_STRIDE;
                        else
                            L = S.lights + (size_t)l * LIGHT_STRIDE;
                        const R x1 = rng_draw<R>(key, sample, bounce, 4 * l + 0);
                        const R x2 = rng_draw<R>(key, sample, bounce, 4 * l + 1);
                        V3<R> wl, q;
                        const R pl = fast_light_sample(L, p, x1, x2, wl, q);
                        if (!(pl > R(0)))
                            continue;
                        R pb;
                        const V3<R> f = fast_brdf(m, nrm, wo, wl, pb);
                        const R cn = dot(nrm, wl);
                        if (!(cn > R(0)))
                            continue;
                        n_tests++;
                        if constexpr (SQ1)
                            park = true, park_x1 = x1, park_x2 = x2; // the term counts unless the test fails later
                        else
                        {
                            if (!visible<R, BRUTE>(S, q, p, stk, hit_k, 0xFFFFFFFFu, n_visit))
                                continue;
                        }
                        // cos / pl times the balance weight pl / (pl + pb); at the last bounce no sampled
                        // direction follows that could collect the other share: the light sample counts whole
                        const R w = (go_on && bounce + 1 < depth) ? cn / (pl + pb) : cn / pl;
                        direct = direct + V3<R>{thr.x * L[1] * f.x * w, thr.y * L[2] * f.y * w, thr.z * L[3] * f.z * w};
                    }
                    if (go_on)
                    {
                        const R x0 = rng_draw<R>(key, sample, bounce, D0), x1 = rng_draw<R>(key, sample, bounce, D0 + 1);
                        const R x2 = rng_draw<R>(key, sample, bounce, D0 + 2);
                        go_on = fast_brdf_sample(m, nrm, wo, x0, x1, x2, wi);
                        if (go_on)
                        {
                            R pb;
                            const V3<R> f = fast_brdf(m, nrm, wo, wi, pb);
                            go_on = pb > R(0);
                            if (go_on)
                            {
                                const R c = dot(nrm, wi) / pb;
                                thr = {thr.x * (f.x * c), thr.y * (f.y * c), thr.z * (f.z * c)};
                                pdf_prev = pb;
                            }
                        }
                    }
                }
                store_record(record_at(st, g, slot, depth, uint32_t(bounce)), direct, fast_other);
                n_rec++;
                // Russian roulette from bounce 3 on
                if (go_on && bounce >= 3)
                {
                    const R q = fmin(max3(thr.x, thr.y, thr.z), R(0.95));
                    go_on = rng_draw<R>(key, sample, bounce, D0 + 3) < q;
                    if (go_on)
                        thr = {thr.x / q, thr.y / q, thr.z / q};
                }
                if (go_on && bounce + 1 < depth)
                {
                    o = p, d = wi, prim = hit_k;
                    bounce++;
                    alive = true;
                    if constexpr (!BRUTE)
                        trav_init(S, T, o, d), trav_done = false;
                }
                else
                    finished = true, fin_state = uint32_t(bounce + 1) << 16;
            }
        }
        else if (shaded)
        {
            // ---- shading normal: precomputed per triangle when it is flat (its three vertex normals
            // are equal) or when barycentric normals are off; else interpolated ---------------------
            const V3<R> wo = -d;
            R fn0, fn1, fn2, fn3;
            if (NRM)
                fn0 = lds_nrm[hit_k * 4], fn1 = lds_nrm[hit_k * 4 + 1], fn2 = lds_nrm[hit_k * 4 + 2], fn3 = lds_nrm[hit_k * 4 + 3];
            else
            {
                const R *__restrict__ fn = S.tri_nrm + (size_t)hit_k * 4;
                fn0 = fn[0], fn1 = fn[1], fn2 = fn[2], fn3 = fn[3];
            }
            V3<R> nrm = {fn0, fn1, fn2};
            if (fn3 == R(0))
            {
                const int face = S.tri_face[hit_k];
                const float *__restrict__ vp = S.verts + (size_t)face * 9;
                const float *__restrict__ np = S.normals + (size_t)face * 9;
                const Float9 fv = load9(vp), fq = load9(np);
                V3<R> a = {R(fv.v[0]), R(fv.v[1]), R(fv.v[2])}, b = {R(fv.v[3]), R(fv.v[4]), R(fv.v[5])}, c = {R(fv.v[6]), R(fv.v[7]), R(fv.v[8])};
                V3<R> w = barycentric(p, a, b, c);
                V3<R> acc = w.x * V3<R>{R(fq.v[0]), R(fq.v[1]), R(fq.v[2])};
                acc = acc + w.y * V3<R>{R(fq.v[3]), R(fq.v[4]), R(fq.v[5])};
                acc = acc + w.z * V3<R>{R(fq.v[6]), R(fq.v[7]), R(fq.v[8])};
                nrm = fast_normalized(acc);
            }

            // ---- NEE over all lights, in order; a square-light sample needs a visibility test ----
            V3<R> direct = {R(0), R(0), R(0)};
            for (int l = 0; l < S.n_lights; l++)
            {
                const R *L;
                if constexpr (TAB)
                    L = lds_lights + l * LIGHT_STRIDE;
                else
                    L = S.lights + (size_t)l * LIGHT_STRIDE;
                V3<R> ro = {R(0), R(0), R(0)}, rd = {R(0), R(0), R(0)};
                R pdf;
                const bool square = SQ1 || (LIGHTS != 0 && L[0] == R(0)); // LIGHTS == 0: the scene has no square light
                // a square light takes two draws; a sphere light makes its own, and only when the point is inside it
                R x1 = R(0), x2 = R(0);
                if (square)
                    x1 = rng_draw<R>(key, sample, bounce, 4 * l + 0), x2 = rng_draw<R>(key, sample, bounce, 4 * l + 1);
                if (square)
                    pdf = square_light_sample(L, p, x1, x2, ro, rd);
                else
                    pdf = sphere_light_sample(L, p, key, uint32_t(sample), uint32_t(bounce), l, ro, rd);
                if (pdf < R(1e-10))
                    continue;
                const R cn = dot(nrm, -rd);
                if (cn < R(0))
                    continue;
                const V3<R> f = bxdf_eval(m, rd, wo, nrm);
                const R w = cn / pdf;
                const V3<R> term = {(L[1] * w) * f.x, (L[2] * w) * f.y, (L[3] * w) * f.z};
                if (square)
                {
                    n_tests++;
                    if constexpr (SQ1)
                        park = true, park_x1 = x1, park_x2 = x2; // the term counts unless the test fails later
                    else if constexpr (LIGHTS == 2)
                    {
                        if (!visible<R, BRUTE>(S, ro, p, stk, hit_k, 0xFFFFFFFFu, n_visit))
                            continue;
                    }
                }
                direct = direct + term;
            }

            // ---- next direction and indirect coefficient (after the light loop: fewer values live in it) --
            V3<R> coef = {R(0), R(0), R(0)}, wi = {R(0), R(0), R(0)};
            if (S.enable_indirect)
            {
                const R x1 = rng_draw<R>(key, sample, bounce, 4 * S.n_lights + 0);
                const R x2 = rng_draw<R>(key, sample, bounce, 4 * S.n_lights + 1);
                bxdf_sample(m, nrm, wo, x1, x2, wi, coef);
            }

            // ---- one stack record per shaded segment -------------------------------------------------
            store_record(record_at(st, g, slot, depth, uint32_t(bounce)), direct, coef);
            n_rec++;
            if (S.enable_indirect && bounce + 1 < depth)
            {
                o = p, d = wi, prim = hit_k;
                bounce++;
                alive = true;
                if constexpr (!BRUTE)
                    trav_init(S, T, o, d), trav_done = false;
            }
            else
                finished = true, fin_state = uint32_t(bounce + 1) << 16;
        }

        PTX_PHASE_END(2);
        if constexpr (SQ1)
        {
            // all lanes are back together here: park this bounce's tests, test 64 once they are there
            const unsigned long long pm = __ballot(park);
            if (park)
            {
                const uint32_t e = (sq_head + sq_count + uint32_t(__popcll(pm & lt_mask))) % ShadowQueue<R>::CAP;
                sq.xi[0][e] = park_x1, sq.xi[1][e] = park_x2;
                sq.p2[0][e] = p.x, sq.p2[1][e] = p.y, sq.p2[2][e] = p.z;
                sq.own_bounce[e] = (uint32_t(hit_k) << 16) | seg_bounce;
                sq.slot[e] = uint8_t(seg_slot);
            }
            sq_count += uint32_t(__popcll(pm));
            sq_pushed += uint32_t(__popcll(pm));
            if (sq_count >= 64u)
                drain(64u);
        }
        // finished paths queue up for their resolve pass
        {
            const unsigned long long fm = __ballot(finished);
            if (finished)
            {
                const uint32_t e = (dq_head + dq_count + uint32_t(__popcll(fm & lt_mask))) % DoneQueue::CAP;
                dq.pid[e] = seg_pid, dq.state[e] = fin_state, dq.need[e] = uint16_t(sq_pushed), dq.slot[e] = uint16_t(seg_slot);
            }
            dq_count += uint32_t(__popcll(fm));
            if (dq_count >= 64u)
            {
                // resolve once the 64th path in line has no parked test left (or the ring fills up)
                const uint32_t need64 = uint32_t(__builtin_amdgcn_readfirstlane(int(dq.need[(dq_head + 63u) % DoneQueue::CAP])));
                if (int16_t(uint16_t(sq_drained) - uint16_t(need64)) >= 0 || dq_count > uint32_t(DoneQueue::CAP) - 64u)
                    resolve(64u);
            }
        }
        PTX_PHASE_END(5);
    }
    while (dq_count > 0u)
        resolve(dq_count < 64u ? dq_count : 64u);
#ifdef PTX_PHASE_TIMING
    if (lane == 0 && g % 500 == 3)
    {
        printf("wave %d cycles: start %llu (%u) hit %llu (%u) shade %llu (%u) tests %llu (%u) resolve %llu (%u) queues %llu (%u)\n", g, ph_t[0], ph_n[0],
               ph_t[1], ph_n[1], ph_t[2], ph_n[2], ph_t[3], ph_n[3], ph_t[4], ph_n[4], ph_t[5], ph_n[5]);
        printf("wave %d lanes: start %llu/%llu hit %llu/%llu shade %llu/%llu tests %llu/%llu resolve %llu/%llu exact %llu/%llu\n", g, ph_lane_work[0],
               ph_lane_room[0], ph_lane_work[1], ph_lane_room[1], ph_lane_work[2], ph_lane_room[2], ph_lane_work[3], ph_lane_room[3], ph_lane_work[4],
               ph_lane_room[4], ph_lane_work[5], ph_lane_room[5]);
    }
#endif
    // per-wave statistics (the wave owns its row)
    for (int off = 32; off > 0; off >>= 1)
    {
        n_seg += __shfl_down(n_seg, off, 64);
        n_tests += __shfl_down(n_tests, off, 64);
        n_rec += __shfl_down(n_rec, off, 64);
    }
    unsigned long long visits0 = n_visit[0], visits1 = n_visit[1];
    for (int off = 32; off > 0; off >>= 1)
    {
        visits0 += __shfl_down(visits0, off, 64);
        visits1 += __shfl_down(visits1, off, 64);
    }
    if (lane == 0)
    {
        st.wave_stats[g * WAVE_STATS + 0] = n_seg;
        st.wave_stats[g * WAVE_STATS + 1] = n_tests;
        st.wave_stats[g * WAVE_STATS + 2] = n_iter;
        st.wave_stats[g * WAVE_STATS + 3] = n_rec;
        st.wave_stats[g * WAVE_STATS + 4] = visits0;
        st.wave_stats[g * WAVE_STATS + 5] = visits1;
    }
}

// The two precisions are separate kernels because their register budgets differ: the f64 tracer
// runs 3 waves per SIMD (168 vector registers; with in-place visibility tests it would take 174 left
// alone, and holding it to 168 costs no spill on LDS-resident scenes and 16 to 176 bytes on BVH ones:
// cbox_cube 2 680 -> 3 232 Mpaths/s), the f32 tracer 4 (128) or whatever its LDS allows.
template <bool BRUTE, bool TAB, int LIGHTS>
__global__ void __launch_bounds__(BLOCK, 3) k_trace_f64(DScene<double> S, DCamera cam, DBatch B, DState<double> st, int chunk)
{
    trace_body<double, BRUTE, TAB, LIGHTS>(S, cam, B, st, chunk);
}
template <bool BRUTE, bool TAB, int LIGHTS>
__global__ void __launch_bounds__(BLOCK, BRUTE ? 4 : 3) k_trace_f32(DScene<float> S, DCamera cam, DBatch B, DState<float> st, int chunk)
{
    trace_body<float, BRUTE, TAB, LIGHTS>(S, cam, B, st, chunk);
}
// the physically based estimator on the same tracer (LIGHTS: 1 -- one light, its tests parked;
// 2 -- tests in place; 0 -- no light).  On LDS-resident scenes the f64 kernel is held to three waves
// per SIMD (168 registers; 64 bytes of spills with parked tests, 112 with tests in place: cbox 2 793 ->
// 3 115, cbox_cube 2 053 -> 2 207 Mpaths/s); the BVH variants would spill 272 bytes and stay at two.
template <bool BRUTE, bool TAB, int LIGHTS>
__global__ void __launch_bounds__(BLOCK, BRUTE ? 3 : 2) k_trace_fast_f64(DScene<double> S, DCamera cam, DBatch B, DState<double> st, int chunk)
{
    trace_body<double, BRUTE, TAB, LIGHTS, true>(S, cam, B, st, chunk);
}
template <bool BRUTE, bool TAB, int LIGHTS>
__global__ void __launch_bounds__(BLOCK, (BRUTE && LIGHTS == 1) ? 4 : 3) k_trace_fast_f32(DScene<float> S, DCamera cam, DBatch B, DState<float> st, int chunk)
{
    trace_body<float, BRUTE, TAB, LIGHTS, true>(S, cam, B, st, chunk);
}

} // namespace ptx
#include "bvh_tracer.hpp"
#include "wavefront.hpp"
namespace ptx
{

// one thread per pixel of the batch: continues the running per-pixel sum in sample order
template <class R> __global__ void __launch_bounds__(BLOCK) k_accumulate(DBatch B, DState<R> st, double *__restrict__ sum)
{
    const int64_t pix = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (pix >= B.n_pix)
        return;
    const int64_t N = st.N;
    const int lr = B.row0 + int(pix / B.n_cols), col = B.col0 + int(pix % B.n_cols);
    const int row = local_row_to_image_row(B, lr);
    double *dst = sum + ((int64_t)row * B.image_width + col) * 3;
    double cr = dst[0], cg = dst[1], cb = dst[2];
    for (int s = 0; s < B.n_samples; s++)
    {
        const int64_t pid = (int64_t)s * B.n_pix + pix;
        cr += double(st.result[pid]) / B.spp_total;
        cg += double(st.result[N + pid]) / B.spp_total;
        cb += double(st.result[2 * N + pid]) / B.spp_total;
    }
    dst[0] = cr, dst[1] = cg, dst[2] = cb;
}

// sums the per-wave statistics of a batch into the totals (single workgroup, no atomics)
template <class R> __global__ void __launch_bounds__(BLOCK) k_reduce_stats(DBatch B, DState<R> st)
{
    __shared__ unsigned long long part[WAVE_STATS][BLOCK];
    unsigned long long a[WAVE_STATS] = {0};
    for (int g = threadIdx.x; g < st.n_waves; g += BLOCK)
        for (int c = 0; c < WAVE_STATS; c++)
            a[c] += st.wave_stats[g * WAVE_STATS + c];
    for (int c = 0; c < WAVE_STATS; c++)
        part[c][threadIdx.x] = a[c];
    __syncthreads();
    for (int w = BLOCK / 2; w > 0; w >>= 1)
    {
        if (int(threadIdx.x) < w)
            for (int c = 0; c < WAVE_STATS; c++)
                part[c][threadIdx.x] += part[c][threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0)
    {
        st.stats[0] += part[0][0]; // segments
        st.stats[1] += part[1][0]; // shadow tests
        st.stats[2] += (unsigned long long)B.n_paths;
        st.stats[3] += part[2][0]; // wave iterations of k_trace (64 lanes each)
        st.stats[4] += part[3][0]; // stack records written (shaded segments)
        st.stats[5] += part[4][0]; // BVH node records visited
        st.stats[6] += part[5][0]; // BVH triangle records visited
        *st.pool = 0u;             // the next batch starts from path id 0
    }
}

} // namespace ptx
