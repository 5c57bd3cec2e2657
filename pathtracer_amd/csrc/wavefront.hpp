// Wavefront tracer of BVH scenes without square lights: traversal and shading are separate
// kernels.  Included by kernels.hpp.
//
// The closest-hit traversal of a BVH scene waits for one dependent record fetch per step, and what
// hides that wait is the number of waves a SIMD holds.  Traversal alone needs about 90 vector
// registers in double precision -- five waves per SIMD -- while a kernel that also shades needs
// 168 (three).  So the segments of a batch of paths go through memory between two kernels:
//
//   k_wf_camera    one thread per pixel of the batch: its camera ray -> ray queue (the samples of a
//                  pixel share the ray -- the reference has no per-sample jitter -- and its hit)
//   k_wf_traverse  persistent waves; a lane takes the next ray of the queue whenever its traversal
//                  is complete (one atomic add per refill of a wave) and writes (t, triangle)
//   k_wf_shade     one thread per ray of the queue: termination rules, shading normal, light
//                  sampling, next direction (cPathTracer.cpp:598-673 of the reference); writes the
//                  bounce record and, compacted per wave, the continuation ray -> the other queue
//   k_wf_resolve   one thread per path: back-to-front fold of its records (cPathTracer.cpp:679-689)
//
// A queue entry is (path id, origin, direction); the hit comes back in (t, triangle) at the same
// index.  Queue counts stay on the device: a launch reads the count the previous one produced.
#pragma once

namespace ptx
{

// one ray queue (structure of arrays over `cap` entries) and the hits of its rays
template <class R> struct WfQueue
{
    R *o, *d;          // [3][cap]
    uint32_t *pid;     // [cap] path id within the batch
    R *t;              // [cap] ray parameter of the closest hit
    int32_t *tri;      // [cap] intersector-order triangle, -1: miss
    uint32_t *count;   // rays in the queue
    uint32_t *pool;    // next ray no lane has taken yet (k_wf_traverse)
    int64_t cap;
};
// The queue of visibility tests keeps a test in the slot light x segments + segment (where the gather
// looks for its answer) and names the slots that hold one in a dense list per light: list l is
// pid[l * segments ...], its length count[WF_TEST_COUNTS + l], the segment count count[WF_TEST_SEGS].
// The any-hit launch walks the lists one after the other, so no lane ever takes an empty slot and
// neighbouring lanes test neighbouring segments against the same light.
constexpr int WF_MAX_LIGHTS = 32, WF_TEST_SEGS = 2, WF_TEST_COUNTS = 4; // a segment names its tests in a 32-bit set
constexpr int WF_QUEUE_HEADER_WORDS = WF_TEST_COUNTS + WF_MAX_LIGHTS;

// a wave runs a triangle step (exact test in R) instead of a node step (four f32 slab tests) when
// lanes at triangles x A > lanes at nodes x B
#ifndef PTX_WF_TRI_A
#define PTX_WF_TRI_A 2
#define PTX_WF_TRI_B 3
#endif
#ifndef PTX_WF_CHUNK
#define PTX_WF_CHUNK 256
#endif
constexpr int WF_CHUNK = PTX_WF_CHUNK; // queue entries a wave of the traversal kernel takes per atomic add

// Persistent traversal.  `refill`: a wave fetches new rays when that many of its lanes are idle.
// ANY: the queue holds visibility tests -- segment from o along d of length t (an input here);
// tri comes back 1 when something blocks it, else -1 (the any-hit traversal of cAccelStruct::VisTest).
template <class R, int ANY> __device__ __forceinline__ void wf_traverse_body(const DScene<R> &S, const WfQueue<R> &Q, int refill, unsigned long long *wave_stats)
{
    int *stk = dyn_stack + threadIdx.x;
    __shared__ uint32_t list_begin[ANY ? WF_MAX_LIGHTS + 1 : 1]; // ANY: where each light's list starts in the walk over all of them
    if constexpr (ANY)
        if (threadIdx.x == 0)
        {
            uint32_t a = 0;
            for (int l = 0; l < S.n_lights; l++)
                list_begin[l] = a, a += Q.count[WF_TEST_COUNTS + l];
            list_begin[S.n_lights] = a;
        }
    stage_hot_nodes(S);
    __syncthreads(); // the only barrier
    const int lane = lane_id(), g = global_wave();
    const uint32_t n = ANY ? list_begin[S.n_lights] : *Q.count;
    const size_t n_seg = ANY ? size_t(Q.count[WF_TEST_SEGS]) : size_t(0);
    bool busy = false, drained = false;
    uint32_t idx = 0;
    uint32_t next = 0, end = 0; // the wave's current range of queue entries (wave-uniform)
    Trav<R> T;
    trav_init(S, T, V3<R>{R(0), R(0), R(0)}, V3<R>{R(1), R(1), R(1)});
    V3<R> o = {R(0), R(0), R(0)}, d = {R(0), R(0), R(0)};
    uint32_t n_visit[2] = {0, 0};
    const size_t cap = size_t(Q.cap);
    for (;;)
    {
        // ---- idle lanes take the next rays of the queue -----------------------------------------
        const unsigned long long idle = __ballot(!busy);
        const uint32_t n_idle = uint32_t(__popcll(idle));
        if (!drained && n_idle > 0u)
        {
            if (next == end)
            {
                // the wave's range of queue entries is used up: one atomic add for the next WF_CHUNK
                uint32_t base = 0;
                if (lane == 0)
                    base = atomicAdd(Q.pool, uint32_t(WF_CHUNK));
                base = uint32_t(__builtin_amdgcn_readfirstlane(int(base)));
                next = base < n ? base : n;
                end = n - next < uint32_t(WF_CHUNK) ? n : next + uint32_t(WF_CHUNK);
            }
            uint32_t i = next + rank_in(idle);
            if (!busy && i < end)
            {
                if constexpr (ANY)
                {
                    // entry i of the lists laid end to end -> the slot it names
                    int l = 0;
                    while (l + 1 < S.n_lights && i >= list_begin[l + 1])
                        l++;
                    i = Q.pid[size_t(l) * n_seg + (i - list_begin[l])];
                }
                o = {Q.o[i], Q.o[cap + i], Q.o[2 * cap + i]};
                d = {Q.d[i], Q.d[cap + i], Q.d[2 * cap + i]};
                idx = i, busy = true;
                if constexpr (ANY)
                {
                    const R len = Q.t[i];
                    trav_init(S, T, o, d, len);
                    busy = len >= R(0);
                }
                else
                    trav_init(S, T, o, d);
            }
            next = end - next < n_idle ? end : next + n_idle;
            if (next >= n)
                drained = true; // every ray of the queue has a lane
        }
        if (__ballot(busy) == 0ull)
        {
            if (drained)
                break; // nothing left to take and every lane idle
            continue;  // nothing taken: take the next ones (ends: the range only moves forward)
        }
        // ---- traversal steps until enough lanes are idle again ----------------------------------
        for (;;)
        {
            const int n_trav = __popcll(__ballot(busy));
            if (n_trav == 0 || (!drained && 64 - n_trav >= refill))
                break;
            const bool at_tri = busy && T.cur >= S.n_nodes;
            const int n_tri = __popcll(__ballot(at_tri));
            bool more = true;
            if (n_tri * PTX_WF_TRI_A > (n_trav - n_tri) * PTX_WF_TRI_B)
            {
                if (at_tri)
                    more = bvh_step<R, ANY, 2>(S, o, d, T, stk, n_visit);
            }
            else if (busy && !at_tri)
                more = bvh_step<R, ANY, 1>(S, o, d, T, stk, n_visit);
            if (!more)
            {
                if constexpr (!ANY)
                    Q.t[idx] = T.best_t;
                Q.tri[idx] = T.target;
                busy = false;
            }
        }
    }
    unsigned long long v0 = n_visit[0], v1 = n_visit[1];
    for (int off = 32; off > 0; off >>= 1)
        v0 += __shfl_down(v0, off, 64), v1 += __shfl_down(v1, off, 64);
    if (lane == 0)
    {
        wave_stats[g * WAVE_STATS + 4] += v0;
        wave_stats[g * WAVE_STATS + 5] += v1;
    }
}

// workgroups per CU the f64 traversal kernel is compiled for (80 registers and 12 bytes of spills at 6: room +3 % over 5 with 96 registers; 7 spills more and loses 13 %)
#ifndef PTX_WF_F64_BLOCKS
#define PTX_WF_F64_BLOCKS 6
#endif
template <int UNUSED = 0> __global__ void __launch_bounds__(BLOCK, PTX_WF_F64_BLOCKS) k_wf_traverse_f64(DScene<double> S, WfQueue<double> Q, int refill, unsigned long long *wave_stats)
{
    wf_traverse_body<double, 0>(S, Q, refill, wave_stats);
}
template <int UNUSED = 0> __global__ void __launch_bounds__(BLOCK, PTX_WF_F64_BLOCKS) k_wf_anyhit_f64(DScene<double> S, WfQueue<double> Q, int refill, unsigned long long *wave_stats)
{
    wf_traverse_body<double, 1>(S, Q, refill, wave_stats);
}
template <int UNUSED = 0> __global__ void __launch_bounds__(BLOCK, 8) k_wf_traverse_f32(DScene<float> S, WfQueue<float> Q, int refill, unsigned long long *wave_stats)
{
    wf_traverse_body<float, 0>(S, Q, refill, wave_stats);
}
template <int UNUSED = 0> __global__ void __launch_bounds__(BLOCK, 8) k_wf_anyhit_f32(DScene<float> S, WfQueue<float> Q, int refill, unsigned long long *wave_stats)
{
    wf_traverse_body<float, 1>(S, Q, refill, wave_stats);
}

// per-batch state of the wavefront tracer besides the two queues
template <class R> struct WfState
{
    R *rec;          // [6][depth][N]: direct rgb, coefficient rgb of every bounce record, by path id
    uint32_t *fin;   // [N] (record count << 16) | (tail material + 1) of the finished path
    R *result;       // [3][N] resolved per-sample radiance (what k_accumulate reads)
    unsigned long long *wave_stats; // [rows][WAVE_STATS], added to
    int64_t N;
    int rows;
    // physically based estimator: what a path carries from bounce to bounce, by path id -- throughput
    // and the density its current direction was sampled with (-1: a delta lobe, -2: only delta lobes
    // since the camera) -- and the light-sampling terms of a bounce whose visibility tests are out:
    // the test of segment i of the queue's n against light l is entry l * n + i of the test queue
    // (length < 0: no test), its term beside it
    R *thr, *pdf;                   // [3][N], [N]
    R *term;                        // [3][test queue capacity]
};

// pixel of a path of the batch and the key of its random numbers
__device__ __forceinline__ void wf_path_pixel(const DBatch &B, uint32_t pid, int &col, int &row)
{
    const uint32_t pix = pid % uint32_t(B.n_pix);
    const int lr = B.row0 + int(pix / uint32_t(B.n_cols));
    col = B.col0 + int(pix % uint32_t(B.n_cols));
    row = local_row_to_image_row(B, lr);
}

// camera rays of the batch -> queue A (cPathTracer::GenerateRay), ONE PER PIXEL: the reference's
// camera rays carry no per-sample jitter, so all samples of a pixel share the ray and its closest
// hit; the first shading launch reads entry (path id mod pixels of the batch) for every path
template <class R> __global__ void __launch_bounds__(BLOCK) k_wf_camera(DCamera cam, DBatch B, WfQueue<R> A)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= B.n_pix)
        return;
    int col, row;
    wf_path_pixel(B, uint32_t(i), col, row);
    V3<R> o, d;
    primary_ray<R>(cam, col, row, o, d);
    A.o[i] = o.x, A.o[A.cap + i] = o.y, A.o[2 * A.cap + i] = o.z;
    A.d[i] = d.x, A.d[A.cap + i] = d.y, A.d[2 * A.cap + i] = d.z;
    A.pid[i] = uint32_t(i);
}

// sets the counters of a queue (single thread; between two launches on the stream)
template <class R> __global__ void k_wf_counts(WfQueue<R> A, uint32_t count_a, WfQueue<R> Bq)
{
    *A.count = count_a, *A.pool = 0u;
    *Bq.count = 0u, *Bq.pool = 0u;
}

// One segment per thread: the hit of ray i of queue `in` is shaded (termination rules, shading
// normal, light sampling over all lights in order, next direction: cPathTracer.cpp:598-673,
// cBxDF::Sample_Li cBxDF.cpp:16-52 of the reference; scenes without square lights, so no
// visibility test), its record stored, and the continuation ray appended to queue `out`.
// The place of a workgroup's continuation rays in the other queue costs ONE atomic add (the counts
// of its waves are summed in LDS first) -- an add per wave on the one counter took longer than the
// shading itself.  The kernel waits on chains of dependent scattered loads (hit -> triangle ->
// material), so what pays is waves in flight: workgroups of 512 threads compiled for 6 waves per
// SIMD (80 registers; three workgroups per CU) are 9 % (room) and 11 % (mis) faster over the whole
// render in f64 than workgroups of 1024 at the 84 registers the compiler takes unasked (one
// workgroup per CU); 256 threads are no better in f64 and 8 % slower in f32 (four times the
// atomic adds), 7 waves per SIMD spill and lose 3 %.
#ifndef PTX_WF_SHADE_BLOCK
#define PTX_WF_SHADE_BLOCK 512
#endif
#ifndef PTX_WF_SHADE_WAVES
#define PTX_WF_SHADE_WAVES 6 // waves per SIMD k_wf_shade is compiled for
#endif
constexpr int WF_SHADE_BLOCK = PTX_WF_SHADE_BLOCK;
// Appends the test slots of a workgroup's segments to the per-light lists of the test queue: one
// atomic add per light and workgroup (the counts of its waves are summed in LDS first), the slots of
// a wave side by side.  tmask: the lights this thread's segment has a test against (bit l: slot
// t0 + l * tstep).  Every thread of the workgroup calls it.
template <class R>
__device__ __forceinline__ void wf_list_tests(const WfQueue<R> &tests, uint32_t tmask, int n_lights, size_t t0, size_t tstep)
{
    __shared__ uint32_t wave_cnt[WF_SHADE_BLOCK / 64][WF_MAX_LIGHTS]; // tests per wave and light, then the wave's offset in the workgroup's range
    __shared__ uint32_t list_base[WF_MAX_LIGHTS];
    const int lane = lane_id(), wave = int(threadIdx.x >> 6);
    for (int l = 0; l < n_lights; l++)
    {
        const unsigned long long b = __ballot((tmask >> l) & 1u);
        if (lane == 0)
            wave_cnt[wave][l] = uint32_t(__popcll(b));
    }
    __syncthreads();
    if (int(threadIdx.x) < n_lights)
    {
        const int l = int(threadIdx.x);
        uint32_t total = 0;
        for (int w = 0; w < WF_SHADE_BLOCK / 64; w++)
        {
            const uint32_t c = wave_cnt[w][l];
            wave_cnt[w][l] = total;
            total += c;
        }
        list_base[l] = total ? atomicAdd(tests.count + WF_TEST_COUNTS + l, total) : 0u;
    }
    __syncthreads();
    for (int l = 0; l < n_lights; l++)
    {
        const bool has = (tmask >> l) & 1u;
        const unsigned long long b = __ballot(has);
        if (has)
            tests.pid[size_t(l) * tstep + list_base[l] + wave_cnt[wave][l] + rank_in(b)] = uint32_t(t0 + size_t(l) * tstep);
    }
}

// CAM: the launch of bounce 0 -- one thread per path of the batch, the hit of its pixel's camera ray.
// SQ: the scene has square lights (at most 12 lights in all), whose samples need the reference's
// visibility test: every light with a term gets its slot light x segments + segment of the test
// queue -- a square light's the segment from the light point to the shaded point, a sphere light's
// the mark "counts without a test" (length -2) -- with the term beside it, the record's direct half
// is left to k_wf_gather, which adds the terms in light order once the any-hit launch has run.
template <class R, bool CAM, bool SQ>
__global__ void __launch_bounds__(WF_SHADE_BLOCK, PTX_WF_SHADE_WAVES)
    k_wf_shade(DScene<R> S, DBatch B, WfQueue<R> in, WfQueue<R> out, WfQueue<R> tests, WfState<R> W, int bnc)
{
    __shared__ uint32_t wave_cont[WF_SHADE_BLOCK / 64]; // continuation rays per wave, then their offsets in the workgroup's range
    __shared__ uint32_t block_base;
    __shared__ uint32_t block_stats[4]; // segments, shading iterations (waves), records, tests of this workgroup
    const uint32_t i = blockIdx.x * WF_SHADE_BLOCK + threadIdx.x;
    const uint32_t n = CAM ? uint32_t(B.n_paths) : *in.count;
    if (blockIdx.x * WF_SHADE_BLOCK >= n)
        return; // the whole workgroup is beyond the queue
    if (threadIdx.x < 4)
        block_stats[threadIdx.x] = 0u;
    const bool mine = i < n;
    const int depth = B.depth;
    const size_t N = size_t(W.N);
    bool cont = false;
    V3<R> p = {R(0), R(0), R(0)}, wi = {R(0), R(0), R(0)};
    uint32_t pid = 0, n_rec = 0, n_test = 0, tmask = 0;
    const size_t tcap = size_t(tests.cap), t0 = size_t(i), tstep = size_t(n); // this segment's test slots: t0 + light * tstep
    if (SQ && i == 0u)
        tests.count[WF_TEST_SEGS] = n;
    if (mine)
    {
        if constexpr (SQ)
            for (int l = 0; l < S.n_lights; l++)
                tests.t[t0 + l * tstep] = R(-1); // no term until the light loop says otherwise
        const size_t cap = size_t(in.cap);
        const uint32_t e = CAM ? i % uint32_t(B.n_pix) : i; // the queue entry of this thread's segment
        pid = CAM ? i : in.pid[e];
        const V3<R> o = {in.o[e], in.o[cap + e], in.o[2 * cap + e]};
        const V3<R> sd = {in.d[e], in.d[cap + e], in.d[2 * cap + e]};
        const int hit_k = in.tri[e];
        p = o + in.t[e] * sd;
        int col, row;
        wf_path_pixel(B, pid, col, row);
        const uint64_t key = rng_pixel_key(B.seed, uint32_t(row) * uint32_t(B.image_width) + uint32_t(col));
        const uint32_t sample = uint32_t(B.sample0) + pid / uint32_t(B.n_pix);
        bool finished = false;
        uint32_t fin_state = 0;
        // ---- termination rules (miss / no material -> tail 0; emitter -> Ka) ----------------------
        int mat_id = -1;
        if (hit_k >= 0)
            mat_id = S.tri_mat[hit_k];
        const R *m = S.mats + (size_t)(mat_id < 0 ? 0 : mat_id) * MAT_STRIDE;
        if (mat_id < 0)
            finished = true, fin_state = uint32_t(bnc) << 16;
        else if (m[13] != R(0))
            finished = true, fin_state = (uint32_t(bnc) << 16) | uint32_t(mat_id + 1);
        else
        {
            // ---- shading normal ----------------------------------------------------------------------
            const V3<R> wo = -sd;
            const R *__restrict__ fn = S.tri_nrm + (size_t)hit_k * 4;
            const R fn0 = fn[0], fn1 = fn[1], fn2 = fn[2], fn3 = fn[3];
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
            // ---- light sampling over all lights, in order (sphere lights: never tested) ---------------
            V3<R> direct = {R(0), R(0), R(0)};
            for (int l = 0; l < S.n_lights; l++)
            {
                const R *L = S.lights + (size_t)l * LIGHT_STRIDE;
                V3<R> ro = {R(0), R(0), R(0)}, rd = {R(0), R(0), R(0)};
                const bool square = SQ && L[0] == R(0);
                // a square light takes two draws; a sphere light makes its own, and only when the point is inside it
                R x1 = R(0), x2 = R(0);
                if (square)
                    x1 = rng_draw<R>(key, sample, bnc, 4 * l + 0), x2 = rng_draw<R>(key, sample, bnc, 4 * l + 1);
                R pdf;
                if (square)
                    pdf = square_light_sample(L, p, x1, x2, ro, rd);
                else
                    pdf = sphere_light_sample(L, p, key, uint32_t(sample), uint32_t(bnc), l, ro, rd);
                if (pdf < R(1e-10))
                    continue;
                const R cn = dot(nrm, -rd);
                if (cn < R(0))
                    continue;
                const V3<R> f = bxdf_eval(m, rd, wo, nrm);
                const R w = cn / pdf;
                const V3<R> term = {(L[1] * w) * f.x, (L[2] * w) * f.y, (L[3] * w) * f.z};
                if constexpr (SQ)
                {
                    const size_t r = t0 + l * tstep;
                    W.term[r] = term.x, W.term[tcap + r] = term.y, W.term[2 * tcap + r] = term.z;
                    if (square)
                    {
                        // the segment from the light point to the shaded point, as visible() runs it
                        const R dist = norm(ro - p);
                        const V3<R> dir = normalized(p - ro);
                        tests.o[r] = ro.x, tests.o[tcap + r] = ro.y, tests.o[2 * tcap + r] = ro.z;
                        tests.d[r] = dir.x, tests.d[tcap + r] = dir.y, tests.d[2 * tcap + r] = dir.z;
                        tests.t[r] = dist;
                        n_test++, tmask |= 1u << l;
                    }
                    else
                        tests.t[r] = R(-2); // a sphere light: its term counts without a test
                }
                else
                    direct = direct + term;
            }
            // ---- next direction and indirect coefficient ---------------------------------------------
            V3<R> coef = {R(0), R(0), R(0)};
            if (S.enable_indirect)
            {
                const R x1 = rng_draw<R>(key, sample, bnc, 4 * S.n_lights + 0);
                const R x2 = rng_draw<R>(key, sample, bnc, 4 * S.n_lights + 1);
                bxdf_sample(m, nrm, wo, x1, x2, wi, coef);
            }
            R *rec = W.rec + size_t(bnc) * N + pid;
            const size_t comp = size_t(depth) * N;
            rec[0] = direct.x, rec[comp] = direct.y, rec[2 * comp] = direct.z;
            rec[3 * comp] = coef.x, rec[4 * comp] = coef.y, rec[5 * comp] = coef.z;
            n_rec = 1;
            if (S.enable_indirect && bnc + 1 < depth)
                cont = true;
            else
                finished = true, fin_state = uint32_t(bnc + 1) << 16;
        }
        if (finished)
            W.fin[pid] = fin_state;
    }
    // ---- continuation rays -> the other queue: compacted per wave, one range per workgroup ---------
    const unsigned long long cm = __ballot(cont);
    const unsigned long long mm = __ballot(mine);
    const unsigned long long rm = __ballot(n_rec != 0u);
    const int lane = lane_id(), wave = int(threadIdx.x >> 6);
    uint32_t wave_tests = n_test;
    if constexpr (SQ)
        for (int off = 32; off > 0; off >>= 1)
            wave_tests += uint32_t(__shfl_xor(int(wave_tests), off, 64));
    __syncthreads(); // block_stats is zero
    if (lane == 0)
    {
        wave_cont[wave] = uint32_t(__popcll(cm));
        if (mm != 0ull)
        {
            // statistics: segments, one 64-lane shading iteration, records, visibility tests
            atomicAdd(&block_stats[0], uint32_t(__popcll(mm)));
            atomicAdd(&block_stats[1], 1u);
            atomicAdd(&block_stats[2], uint32_t(__popcll(rm)));
            if (SQ)
                atomicAdd(&block_stats[3], wave_tests);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t total = 0;
        for (int w = 0; w < WF_SHADE_BLOCK / 64; w++)
        {
            const uint32_t c = wave_cont[w];
            wave_cont[w] = total;
            total += c;
        }
        block_base = total ? atomicAdd(out.count, total) : 0u;
        unsigned long long *row = W.wave_stats + size_t(blockIdx.x % uint32_t(W.rows)) * WAVE_STATS;
        atomicAdd(row + 0, (unsigned long long)block_stats[0]);
        if (SQ)
            atomicAdd(row + 1, (unsigned long long)block_stats[3]);
        atomicAdd(row + 2, (unsigned long long)block_stats[1]);
        atomicAdd(row + 3, (unsigned long long)block_stats[2]);
    }
    __syncthreads();
    if (cont)
    {
        const size_t r = size_t(block_base) + wave_cont[wave] + rank_in(cm), cap = size_t(out.cap);
        out.o[r] = p.x, out.o[cap + r] = p.y, out.o[2 * cap + r] = p.z;
        out.d[r] = wi.x, out.d[cap + r] = wi.y, out.d[2 * cap + r] = wi.z;
        out.pid[r] = pid;
    }
    if constexpr (SQ)
        wf_list_tests(tests, tmask, S.n_lights, t0, tstep);
}

// The physically based estimator (device_math.hpp, "fast"; the FAST branches of trace_body and
// trace_body_rings) on the wavefront tracer.  One segment per thread, in ONE pass: what the segment
// collected besides light sampling, the termination rules, the light samples with their terms --
// every one of them needs its visibility test, which goes to the test queue `tests` (slot light x
// segments + segment, so that neighbouring threads write neighbouring slots: from the light point to
// the shaded point, as the reference's VisTest runs) with the term beside it -- the
// next direction, Russian roulette, the continuation ray.  The record's light-sampling half is
// filled in by k_wf_gather once the any-hit launch has answered the tests.
#ifndef PTX_WF_SHADE_FAST_WAVES
#define PTX_WF_SHADE_FAST_WAVES 1 // waves per SIMD k_wf_shade_fast is compiled for (1: what its registers allow, 4 in f64; at 5 / 6 it spills: -6 / -14 % over the whole render)
#endif
template <class R, bool CAM>
__global__ void __launch_bounds__(WF_SHADE_BLOCK, PTX_WF_SHADE_FAST_WAVES)
    k_wf_shade_fast(DScene<R> S, DBatch B, WfQueue<R> in, WfQueue<R> out, WfQueue<R> tests, WfState<R> W, int bnc)
{
    __shared__ uint32_t wave_cont[WF_SHADE_BLOCK / 64];
    __shared__ uint32_t block_base;
    __shared__ uint32_t block_stats[4]; // segments, shading iterations (waves), records, tests of this workgroup
    const uint32_t i = blockIdx.x * WF_SHADE_BLOCK + threadIdx.x;
    const uint32_t n = CAM ? uint32_t(B.n_paths) : *in.count;
    if (blockIdx.x * WF_SHADE_BLOCK >= n)
        return; // the whole workgroup is beyond the queue
    if (threadIdx.x < 4)
        block_stats[threadIdx.x] = 0u;
    const bool mine = i < n;
    const int depth = B.depth;
    const size_t N = size_t(W.N);
    bool cont = false;
    V3<R> p = {R(0), R(0), R(0)}, wi = {R(0), R(0), R(0)};
    uint32_t pid = 0, n_rec = 0, n_test = 0, tmask = 0;
    const size_t tcap = size_t(tests.cap), t0 = size_t(i), tstep = size_t(n); // this segment's test slots: t0 + light * tstep
    if (i == 0u)
        tests.count[WF_TEST_SEGS] = n;
    if (mine)
    {
        for (int l = 0; l < S.n_lights; l++)
            tests.t[t0 + l * tstep] = R(-1); // no test until the light loop says otherwise
        const size_t cap = size_t(in.cap);
        const uint32_t e = CAM ? i % uint32_t(B.n_pix) : i; // the queue entry of this thread's segment
        pid = CAM ? i : in.pid[e];
        const V3<R> org = {in.o[e], in.o[cap + e], in.o[2 * cap + e]};
        const V3<R> sd = {in.d[e], in.d[cap + e], in.d[2 * cap + e]};
        const int hit_k = in.tri[e];
        p = org + in.t[e] * sd;
        int col, row;
        wf_path_pixel(B, pid, col, row);
        const uint64_t key = rng_pixel_key(B.seed, uint32_t(row) * uint32_t(B.image_width) + uint32_t(col));
        const uint32_t sample = uint32_t(B.sample0) + pid / uint32_t(B.n_pix);
        bool finished = false;
        uint32_t fin_state = 0;
        int mat_id = -1;
        if (hit_k >= 0)
            mat_id = S.tri_mat[hit_k];
        const R *m = S.mats + (size_t)(mat_id < 0 ? 0 : mat_id) * MAT_STRIDE;
        // what the path carries: throughput, the density of its current direction
        V3<R> thr = {R(1), R(1), R(1)};
        R pdf_prev = R(-2);
        V3<R> other = {R(0), R(0), R(0)};
        if (bnc > 0)
        {
            thr = {W.thr[pid], W.thr[N + pid], W.thr[2 * N + pid]};
            pdf_prev = W.pdf[pid];
        }
        // conf lights the ray crossed before its hit: emitters that do not occlude
        if (pdf_prev > R(-2))
        {
            const R t_geom = hit_k >= 0 ? norm(p - org) : Consts<R>::big;
            for (int l = 0; l < S.n_lights; l++)
            {
                const R *L = S.lights + (size_t)l * LIGHT_STRIDE;
                R tl;
                const R pl = fast_light_pdf(L, org, sd, tl);
                if (pl > R(0) && tl < t_geom)
                {
                    const R w = pdf_prev < R(0) ? R(1) : pdf_prev / (pdf_prev + pl);
                    other = other + V3<R>{thr.x * L[1] * w, thr.y * L[2] * w, thr.z * L[3] * w};
                }
            }
        }
        R *rec = W.rec + size_t(bnc) * N + pid;
        const size_t comp = size_t(depth) * N;
        if (mat_id < 0 || m[13] != R(0))
        {
            if (mat_id >= 0 && pdf_prev == R(-2))
                other = other + V3<R>{thr.x * m[0], thr.y * m[1], thr.z * m[2]}; // an emitter the camera sees
            rec[0] = R(0), rec[comp] = R(0), rec[2 * comp] = R(0);
            rec[3 * comp] = other.x, rec[4 * comp] = other.y, rec[5 * comp] = other.z;
            n_rec = 1;
            finished = true, fin_state = uint32_t(bnc + 1) << 16;
        }
        else
        {
            const V3<R> wo = -sd;
            const R *__restrict__ fn = S.tri_nrm + (size_t)hit_k * 4;
            const R fn0 = fn[0], fn1 = fn[1], fn2 = fn[2], fn3 = fn[3];
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
            bool go_on = S.enable_indirect != 0;
            R pdf_next = pdf_prev;
            if (m[12] != R(0))
            {
                // glass: a delta lobe, no light sampling
                if (go_on)
                {
                    wi = fast_glass_sample(m, nrm, wo, rng_draw<R>(key, sample, bnc, D0));
                    pdf_next = pdf_prev == R(-2) ? R(-2) : R(-1);
                }
            }
            else
            {
                if (dot(nrm, wo) < R(0))
                    nrm = -nrm;
                for (int l = 0; l < S.n_lights; l++)
                {
                    const R *L = S.lights + (size_t)l * LIGHT_STRIDE;
                    const R x1 = rng_draw<R>(key, sample, bnc, 4 * l + 0);
                    const R x2 = rng_draw<R>(key, sample, bnc, 4 * l + 1);
                    V3<R> wl, q;
                    const R pl = fast_light_sample(L, p, x1, x2, wl, q);
                    if (!(pl > R(0)))
                        continue;
                    R pb;
                    const V3<R> f = fast_brdf(m, nrm, wo, wl, pb);
                    const R cn = dot(nrm, wl);
                    if (!(cn > R(0)))
                        continue;
                    // cos / pl times the balance weight pl / (pl + pb); at the last bounce the light sample counts whole
                    const R w = (go_on && bnc + 1 < depth) ? cn / (pl + pb) : cn / pl;
                    // the test: the segment from the light point to the shaded point, as visible() runs it
                    const size_t r = t0 + l * tstep;
                    const R dist = norm(q - p);
                    const V3<R> dir = normalized(p - q);
                    tests.o[r] = q.x, tests.o[tcap + r] = q.y, tests.o[2 * tcap + r] = q.z;
                    tests.d[r] = dir.x, tests.d[tcap + r] = dir.y, tests.d[2 * tcap + r] = dir.z;
                    tests.t[r] = dist;
                    W.term[r] = thr.x * L[1] * f.x * w, W.term[tcap + r] = thr.y * L[2] * f.y * w, W.term[2 * tcap + r] = thr.z * L[3] * f.z * w;
                    n_test++, tmask |= 1u << l;
                }
                if (go_on)
                {
                    const R x0 = rng_draw<R>(key, sample, bnc, D0), x1 = rng_draw<R>(key, sample, bnc, D0 + 1);
                    const R x2 = rng_draw<R>(key, sample, bnc, D0 + 2);
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
                            pdf_next = pb;
                        }
                    }
                }
            }
            // the light-sampling half of the record is k_wf_gather's
            rec[0] = R(0), rec[comp] = R(0), rec[2 * comp] = R(0);
            rec[3 * comp] = other.x, rec[4 * comp] = other.y, rec[5 * comp] = other.z;
            n_rec = 1;
            if (go_on && bnc >= 3)
            {
                // Russian roulette
                const R q = fmin(max3(thr.x, thr.y, thr.z), R(0.95));
                go_on = rng_draw<R>(key, sample, bnc, D0 + 3) < q;
                if (go_on)
                    thr = {thr.x / q, thr.y / q, thr.z / q};
            }
            if (go_on && bnc + 1 < depth)
            {
                cont = true;
                W.thr[pid] = thr.x, W.thr[N + pid] = thr.y, W.thr[2 * N + pid] = thr.z;
                W.pdf[pid] = pdf_next;
            }
            else
                finished = true, fin_state = uint32_t(bnc + 1) << 16;
        }
        if (finished)
            W.fin[pid] = fin_state;
    }
    // ---- continuation rays -> the other queue: one range per workgroup ----------------------------------
    const unsigned long long cm = __ballot(cont);
    const unsigned long long mm = __ballot(mine);
    const unsigned long long rm = __ballot(n_rec != 0u);
    const int lane = lane_id(), wave = int(threadIdx.x >> 6);
    uint32_t wave_tests = n_test;
    for (int off = 32; off > 0; off >>= 1)
        wave_tests += uint32_t(__shfl_xor(int(wave_tests), off, 64));
    __syncthreads(); // block_stats is zero
    if (lane == 0)
    {
        wave_cont[wave] = uint32_t(__popcll(cm));
        if (mm != 0ull)
        {
            atomicAdd(&block_stats[0], uint32_t(__popcll(mm)));
            atomicAdd(&block_stats[1], 1u);
            atomicAdd(&block_stats[2], uint32_t(__popcll(rm)));
            atomicAdd(&block_stats[3], wave_tests);
        }
    }
    __syncthreads();
    if (threadIdx.x == 0)
    {
        uint32_t total = 0;
        for (int w = 0; w < WF_SHADE_BLOCK / 64; w++)
        {
            const uint32_t c = wave_cont[w];
            wave_cont[w] = total;
            total += c;
        }
        block_base = total ? atomicAdd(out.count, total) : 0u;
        unsigned long long *row = W.wave_stats + size_t(blockIdx.x % uint32_t(W.rows)) * WAVE_STATS;
        atomicAdd(row + 0, (unsigned long long)block_stats[0]);
        atomicAdd(row + 1, (unsigned long long)block_stats[3]);
        atomicAdd(row + 2, (unsigned long long)block_stats[1]);
        atomicAdd(row + 3, (unsigned long long)block_stats[2]);
    }
    __syncthreads();
    if (cont)
    {
        const size_t r = size_t(block_base) + wave_cont[wave] + rank_in(cm), cap = size_t(out.cap);
        out.o[r] = p.x, out.o[cap + r] = p.y, out.o[2 * cap + r] = p.z;
        out.d[r] = wi.x, out.d[cap + r] = wi.y, out.d[2 * cap + r] = wi.z;
        out.pid[r] = pid;
    }
    wf_list_tests(tests, tmask, S.n_lights, t0, tstep);
}

// After the any-hit launch: the light-sampling half of every record of the bounce = the terms of the
// segment's unblocked tests, added in light order (the order of the sequential light loop).
template <class R, bool CAM> __global__ void __launch_bounds__(BLOCK) k_wf_gather(DBatch B, WfQueue<R> in, WfQueue<R> tests, WfState<R> W, int bnc, int n_lights)
{
    const uint32_t i = blockIdx.x * BLOCK + threadIdx.x;
    const uint32_t n = CAM ? uint32_t(B.n_paths) : *in.count;
    if (i >= n)
        return;
    const size_t cap = size_t(tests.cap), N = size_t(W.N);
    V3<R> direct = {R(0), R(0), R(0)};
    bool any = false;
    for (int l = 0; l < n_lights; l++)
    {
        const size_t r = size_t(i) + size_t(l) * size_t(n); // light by light, consecutive segments side by side
        const R len = tests.t[r];
        if (len >= R(0) || len == R(-2)) // a test, or a term that counts without one
        {
            any = true;
            if (len < R(0) || tests.tri[r] != 1)
                direct = direct + V3<R>{W.term[r], W.term[cap + r], W.term[2 * cap + r]};
        }
    }
    if (!any)
        return; // the record's light-sampling half stays zero
    const uint32_t pid = CAM ? i : in.pid[i];
    R *rec = W.rec + size_t(bnc) * N + pid;
    const size_t comp = size_t(B.depth) * N;
    rec[0] = direct.x, rec[comp] = direct.y, rec[2 * comp] = direct.z;
}

// One path per thread: its bounce records folded back to front with the per-bounce max-clamp
// (cPathTracer.cpp:679-689), the per-sample result stored for k_accumulate.  FAST: every term
// already carries its throughput, the path's radiance is their plain sum.
template <class R, bool FAST> __global__ void __launch_bounds__(BLOCK) k_wf_resolve(DScene<R> S, DBatch B, WfState<R> W)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    const int64_t pid = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (pid >= B.n_paths)
        return;
    const size_t N = size_t(W.N);
    const uint32_t rs = W.fin[pid];
    const int len = int(rs >> 16), tm = int(rs & 0xffffu);
    R fr = R(0), fg = R(0), fb = R(0);
    if (tm)
    {
        const R *tmat = S.mats + (size_t)(tm - 1) * MAT_STRIDE;
        fr = tmat[0], fg = tmat[1], fb = tmat[2];
    }
    const size_t comp = size_t(B.depth) * N;
    for (int i = len - 1; i >= 0; i--)
    {
        const R *q = W.rec + size_t(i) * N + size_t(pid);
        if constexpr (FAST)
        {
            fr += q[0] + q[3 * comp], fg += q[comp] + q[4 * comp], fb += q[2 * comp] + q[5 * comp];
            continue;
        }
        fr = q[0] + q[3 * comp] * fr;
        fg = q[comp] + q[4 * comp] * fg;
        fb = q[2 * comp] + q[5 * comp] * fb;
        R mx = fr;
        if (fg > mx)
            mx = fg;
        if (fb > mx)
            mx = fb;
        if (R(0) > mx)
            mx = R(0);
        if (mx > R(1))
        {
            const R inv = R(1) / mx;
            fr *= inv, fg *= inv, fb *= inv;
        }
    }
    W.result[pid] = fr, W.result[N + pid] = fg, W.result[2 * N + pid] = fb;
}

// operator form: rays given as [n][3] doubles -> the queue; hits of the queue -> face ids and points
template <class R>
__global__ void __launch_bounds__(BLOCK) k_wf_rays_in(const double *__restrict__ ro, const double *__restrict__ rd, int64_t n, WfQueue<R> Q)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i == 0)
        *Q.count = uint32_t(n), *Q.pool = 0u;
    if (i >= n)
        return;
    const V3<R> dn = normalized(V3<R>{R(rd[i * 3]), R(rd[i * 3 + 1]), R(rd[i * 3 + 2])});
    for (int k = 0; k < 3; k++)
        Q.o[k * Q.cap + i] = R(ro[i * 3 + k]);
    Q.d[i] = dn.x, Q.d[Q.cap + i] = dn.y, Q.d[2 * Q.cap + i] = dn.z;
    Q.pid[i] = uint32_t(i);
}
template <class R>
__global__ void __launch_bounds__(BLOCK) k_wf_hits_out(DScene<R> S, WfQueue<R> Q, int64_t n, int32_t *__restrict__ face_ids, double *__restrict__ pts)
{
    const int64_t i = (int64_t)blockIdx.x * BLOCK + threadIdx.x;
    if (i >= n)
        return;
    const int k = Q.tri[i];
    face_ids[i] = k < 0 ? -1 : S.tri_face[k];
    if (pts)
    {
        const double qn = __builtin_nan("");
        const R t = Q.t[i];
        for (int c = 0; c < 3; c++)
            pts[i * 3 + c] = k < 0 ? qn : double(Q.o[c * Q.cap + i] + t * Q.d[c * Q.cap + i]);
    }
}

} // namespace ptx
