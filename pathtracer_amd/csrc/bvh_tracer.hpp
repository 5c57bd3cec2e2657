// Tracer of BVH scenes (up to 12 square lights): a wave keeps more paths in flight than it has lanes.  Included by kernels.hpp.
//
// With the BVH, the 64 lanes of a wave are at three different kinds of work -- node tests, exact
// triangle tests, shading -- and whatever runs, runs on the lanes that happen to be there.  Here the
// lanes only traverse.  A lane whose traversal is complete hands the hit (point, direction,
// triangle, path id, bounce, RNG key) to the wave's HIT RING and takes its next ray from the wave's
// RAY RING; when 64 hits are waiting, the whole wave shades them, one per lane, and the
// continuation rays go to the ray ring; camera rays enter through the same ring, 64 at a time.
// Both rings live in global memory (a wave-private region, written and read with consecutive
// entries per lane, i.e. coalesced, and by this wave only) because the LDS belongs to the traversal
// stacks; their traffic, about 130 B (f32) to 270 B (f64) per segment, is small next to the BVH
// records a segment fetches.  Shading runs on full waves, the traversal steps on the lanes that are
// not between two rays; every path is in exactly one place (a lane, the ray ring or the hit ring),
// camera rays are only admitted while fewer than 128 paths are in flight, so neither ring
// (256 entries) can overflow, and the wave ends when the pool is used up and all three are empty.
//
// SQ (the scene has square lights, at most 12): a bounce needs the reference's visibility test for
// every square light whose sample is valid before its record can be written.  Shading a fresh hit
// then only finds out which lights those are and sends one ray-ring entry per test -- the hit
// (point, incoming direction, triangle), the light, which of the hit's tests it is and how many the
// hit has; the ray ring is four times longer for that, and a shading pass waits until it has room
// for the most its 64 hits can send.  The lane that takes an entry recomputes the light sample
// point from the path's random numbers and runs the any-hit traversal from the light to the shaded
// point, so the tests of one hit run on as many lanes at once.  A finished test adds its result to
// the hit's word in LDS (per wave and record slot: tests done, blocker bits); the test that
// completes the count hands the hit back to the hit ring with all results in its state word, and
// the next shading pass finishes it: a hit is shaded twice whatever its number of tests, and no
// record is ever patched after the fact.
#pragma once

namespace ptx
{

constexpr int RING_CAP = 256;       // entries of the hit ring, and of the ray ring when no light needs visibility tests
constexpr int RING_CAP_TESTS = 1024; // entries of the ray ring when hits send out visibility tests (up to 12 each)
constexpr int RING_ADMIT = 128;     // camera rays are admitted while at most this many paths are in flight
// state word of a ring entry: bounce in the low 16 bits, then how many visibility tests of this
// bounce are done (4 bits) and their results (bit 20 + k: test k found a blocker)
constexpr uint32_t RING_BOUNCE = 0xFFFFu;
constexpr int RING_TESTS_SHIFT = 16, RING_RESULT_SHIFT = 20, RING_MAX_TESTS = 12;
// lanes that hand over hits / take rays together (DState::ring_batch, option "ring_batch"): a
// hand-over drains the wave's outstanding loads; measured plateau: 32 to 48, default 40

// bytes of one wave's two rings and of its per-slot path table (fast: the table also holds what the
// physically based estimator carries from bounce to bounce)
template <class R> constexpr size_t ring_bytes(bool fast = false, bool tests = false)
{
    return size_t(tests ? RING_CAP_TESTS : RING_CAP) * (6 * sizeof(R) + 4 * 4) + size_t(RING_CAP) * (6 * sizeof(R) + 4 * 3) +
           size_t(RING_SLOTS) * (12 + (fast ? 7 * sizeof(R) : 0));
}


// views of the two rings of one wave (structure of arrays: one field of 256 entries after the
// other) and of its path table.  What a path is -- its id and the key of its random numbers --
// stays in the table, indexed by the record slot the path holds; ring entries and lanes carry the
// slot and the state word only (registers are what the traversal loop is short of).
template <class R, int RQ> struct Rings
{
    // ray ring
    R *ro, *rd;        // [3][CAP] each
    uint32_t *r_slot, *r_bounce; // r_bounce: state word
    int32_t *r_tri;    // shadow entries: the triangle the shaded point lies on
    int32_t *r_light;  // -1: a ray; else a visibility test (shadow entry) against this light
    // hit ring
    R *hp, *hd;        // [3][CAP]: hit point, ray direction
    int32_t *h_tri;
    uint32_t *h_slot, *h_bounce; // h_bounce: state word
    // path table, by record slot
    uint32_t *s_pid, *s_klo, *s_khi;
    // physically based estimator: throughput [3][SLOTS], the density the path's current direction
    // was sampled with (-1: a delta lobe, -2: only delta lobes since the camera), and the origin
    // [3][SLOTS] of its current ray -- written by the shading pass that sends the ray out, read by
    // the one that shades its hit; the traversal in between needs none of it
    R *s_thr, *s_pdf, *s_org;
    __device__ explicit Rings(void *base)
    {
        R *f = reinterpret_cast<R *>(base);
        ro = f, rd = f + 3 * RQ, hp = f + 6 * RQ, hd = hp + 3 * RING_CAP;
        uint32_t *u = reinterpret_cast<uint32_t *>(hd + 3 * RING_CAP);
        r_slot = u, r_bounce = u + RQ;
        r_tri = reinterpret_cast<int32_t *>(u + 2 * RQ);
        r_light = reinterpret_cast<int32_t *>(u + 3 * RQ);
        u += 4 * RQ;
        h_tri = reinterpret_cast<int32_t *>(u);
        h_slot = u + RING_CAP, h_bounce = u + 2 * RING_CAP;
        s_pid = u + 3 * RING_CAP, s_klo = s_pid + RING_SLOTS, s_khi = s_klo + RING_SLOTS;
        s_thr = reinterpret_cast<R *>(s_khi + RING_SLOTS), s_pdf = s_thr + 3 * RING_SLOTS, s_org = s_pdf + RING_SLOTS;
    }
};

// entries written by some lanes of the wave are read by others: wait for the stores (the compute
// unit's vector cache is coherent for its own waves)
__device__ __forceinline__ void ring_sync()
{
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
}

// rank of this lane among the set bits of `mask`
__device__ __forceinline__ uint32_t rank_in(unsigned long long mask)
{
    return __builtin_amdgcn_mbcnt_hi(uint32_t(mask >> 32), __builtin_amdgcn_mbcnt_lo(uint32_t(mask), 0u));
}

// FAST: the physically based estimator (device_math.hpp) on the same rings.  Every light needs its
// visibility test (SQ is set), sent out and joined as described above, every segment writes a record (light-sampling
// term, what else the segment collected), both already multiplied by the throughput, and the
// resolve pass is a plain sum; a path that loses the roulette simply sends no ray out.
template <class R, bool TAB, bool SQ, bool FAST = false>
__device__ __forceinline__ void trace_body_rings(const DScene<R> &S, const DCamera &cam, const DBatch &B, const DState<R> &st, int chunk)
{
    __shared__ R lds_mats[TAB ? LDS_MATS * MAT_STRIDE : 1];
    __shared__ R lds_lights[TAB ? LDS_LIGHTS * LIGHT_STRIDE : 1];
    __shared__ DoneQueue done_queues[BLOCK / 64];
    __shared__ uint16_t free_slots[BLOCK / 64][RING_SLOTS]; // per wave: the record slots no path holds (a stack)
    // per wave and slot: the visibility tests of the path's current hit that are done (count << 16)
    // and which of them found a blocker (bit k: test k)
    __shared__ uint32_t join_words[SQ ? BLOCK / 64 : 1][SQ ? RING_SLOTS : 1];
    for (int k = threadIdx.x; k < (BLOCK / 64) * RING_SLOTS; k += BLOCK)
        free_slots[k / RING_SLOTS][k % RING_SLOTS] = uint16_t(k % RING_SLOTS);
    if (TAB)
    {
        for (int k = threadIdx.x; k < S.n_mats * MAT_STRIDE; k += BLOCK)
            lds_mats[k] = S.mats[k];
        for (int k = threadIdx.x; k < S.n_lights * LIGHT_STRIDE; k += BLOCK)
            lds_lights[k] = S.lights[k];
    }
    int *stk = dyn_stack + threadIdx.x;
    stage_hot_nodes(S);
    __syncthreads(); // the only barrier: from here on the waves of the workgroup are independent
    const int g = global_wave(), lane = lane_id();
    if (g >= st.n_waves)
        return;
    constexpr int RQ = SQ ? RING_CAP_TESTS : RING_CAP; // entries of the ray ring
    const Rings<R, RQ> Q(st.rings + size_t(g) * ring_bytes<R>(FAST, SQ));
    const uint32_t n_paths = uint32_t(B.n_paths);
    uint32_t next = 0, end = 0; // the wave's current range of path ids
    bool drained = false;       // the pool has no ids left (wave-uniform)
    const uint32_t n_pix = uint32_t(B.n_pix), n_cols = uint32_t(B.n_cols);
    const int depth = B.depth;
    const int RING_BATCH = st.ring_batch;
    const V3<R> cam_pos = {R(cam.pos[0]), R(cam.pos[1]), R(cam.pos[2])};
    DoneQueue &dq = done_queues[threadIdx.x >> 6];
    uint32_t dq_head = 0, dq_count = 0;               // wave-uniform
    uint32_t rq_head = 0, rq_count = 0, hq_head = 0, hq_count = 0; // wave-uniform
    // record slots: a path takes one when its camera ray is admitted and returns it in its resolve
    // pass; at most 192 paths are in flight and at most 127 wait in dq, so the stack never runs dry
    uint16_t *free_slot = free_slots[threadIdx.x >> 6];
    uint32_t *join = join_words[SQ ? threadIdx.x >> 6 : 0];
    uint32_t free_count = RING_SLOTS; // wave-uniform
    uint32_t slot = 0;                // of the path whose ray this lane traverses

    // the ray this lane is traversing
    bool busy = false, done = false; // done: traversal complete, the hit waits to be handed over
    Trav<R> T;
    trav_init(S, T, V3<R>{R(0), R(0), R(0)}, V3<R>{R(1), R(1), R(1)});
    V3<R> o = {R(0), R(0), R(0)}, d = {R(0), R(0), R(0)};
    uint32_t bounce = 0; // state word of the ray this lane traverses
    // a visibility test in flight: the hit it belongs to (handed back when the test is done)
    V3<R> vp = {R(0), R(0), R(0)}, vd = {R(0), R(0), R(0)};
    int vtri = -1;
    uint32_t vcode = 0; // which test of that hit: k << 16 | the hit's number of tests << 20
    uint32_t n_seg = 0, n_iter = 0, n_rec = 0, n_tests = 0;
    uint32_t n_visit[2] = {0, 0};
#ifdef PTX_PHASE_TIMING
    unsigned long long ph_t[8] = {0, 0, 0, 0, 0, 0, 0, 0}, ph_last = __builtin_readcyclecounter();
    uint32_t ph_n[8] = {0, 0, 0, 0, 0, 0, 0, 0};
#endif

    // folds 64 (or the last n) finished paths back to front, one per lane
    auto resolve = [&](uint32_t n) {
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
        PTX_PHASE_END(3);
        ring_sync(); // the records were stored by other lanes of this wave
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
        int wlen = len;
        for (int off = 32; off > 0; off >>= 1)
            wlen = max(wlen, __shfl_xor(wlen, off, 64));
        const R *rec = record_at(st, g, rslot, depth, 0u);
        constexpr int G = 2; // levels whose loads are in flight together
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
                        const R inv = R(1) / mx;
                        fr *= inv, fg *= inv, fb *= inv;
                    }
                }
            }
        }
        if (mine)
        {
            st.result[rp] = fr, st.result[st.N + rp] = fg, st.result[2 * st.N + rp] = fb;
            free_slot[free_count + uint32_t(lane)] = uint16_t(rslot); // the slots go back on the stack
        }
        free_count += n;
        dq_head = (dq_head + n) % DoneQueue::CAP;
        dq_count -= n;
        // the next camera rays read the stack the lanes above just wrote
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        PTX_PHASE_END(4);
    };

    // shades the first n (<= 64) hits of the hit ring, one per lane
    auto shade = [&](uint32_t n) {
        PTX_PHASE_END(7);
        ring_sync();
        n_iter++;
        const uint32_t e = (hq_head + uint32_t(lane)) % RING_CAP;
        const bool mine = uint32_t(lane) < n;
        bool finished = false, cont = false, defer = false;
        uint32_t defer_mask = 0; // lights whose segment to this hit needs a visibility test (first visit)
        uint32_t defer_state = 0;
        uint32_t fin_state = 0;
        V3<R> p = {R(0), R(0), R(0)}, wi = {R(0), R(0), R(0)}, keep_d = {R(0), R(0), R(0)};
        uint32_t s_pid = 0, s_bounce = 0, s_klo = 0, s_khi = 0, s_slot = 0;
        int keep_tri = -1;
        if (mine)
        {
            p = {Q.hp[e], Q.hp[RING_CAP + e], Q.hp[2 * RING_CAP + e]};
            const V3<R> sd = {Q.hd[e], Q.hd[RING_CAP + e], Q.hd[2 * RING_CAP + e]};
            const int hit_k = Q.h_tri[e];
            s_slot = Q.h_slot[e], s_bounce = Q.h_bounce[e];
            s_pid = Q.s_pid[s_slot], s_klo = Q.s_klo[s_slot], s_khi = Q.s_khi[s_slot];
            const uint32_t state = s_bounce;
            const int tested = int((state >> RING_TESTS_SHIFT) & 15u); // visibility tests of this bounce already done
            s_bounce &= RING_BOUNCE;
            keep_d = sd, keep_tri = hit_k;
            if (tested == 0)
                n_seg++;
            const uint64_t key = (uint64_t(s_khi) << 32) | s_klo;
            const uint32_t sample = uint32_t(B.sample0) + s_pid / n_pix;
            const int bnc = int(s_bounce);
            // ---- termination rules (miss / no material -> tail 0; emitter -> Ka) ------------------
            int mat_id = -1;
            if (hit_k >= 0)
                mat_id = S.tri_mat[hit_k];
            const int mrow = (mat_id < 0 ? 0 : mat_id) * MAT_STRIDE;
            const R *m;
            if constexpr (TAB)
                m = lds_mats + mrow;
            else
                m = S.mats + mrow;
            if constexpr (FAST)
            {
                // what the path carries: throughput, the density of its current direction, its ray's origin
                V3<R> thr = {R(1), R(1), R(1)};
                R pdf_prev = R(-2);
                V3<R> other = {R(0), R(0), R(0)};
                if (bnc > 0)
                {
                    thr = {Q.s_thr[s_slot], Q.s_thr[RING_SLOTS + s_slot], Q.s_thr[2 * RING_SLOTS + s_slot]};
                    pdf_prev = Q.s_pdf[s_slot];
                }
                // conf lights the ray crossed before its hit: emitters that do not occlude
                if (pdf_prev > R(-2))
                {
                    const V3<R> org = {Q.s_org[s_slot], Q.s_org[RING_SLOTS + s_slot], Q.s_org[2 * RING_SLOTS + s_slot]};
                    const R t_geom = hit_k >= 0 ? norm(p - org) : Consts<R>::big;
                    for (int l = 0; l < S.n_lights; l++)
                    {
                        const R *L;
                        if constexpr (TAB)
                            L = lds_lights + l * LIGHT_STRIDE;
                        else
                            L = S.lights + (size_t)l * LIGHT_STRIDE;
                        R tl;
                        const R pl = fast_light_pdf(L, org, sd, tl);
                        if (pl > R(0) && tl < t_geom)
                        {
                            const R w = pdf_prev < R(0) ? R(1) : pdf_prev / (pdf_prev + pl);
                            other = other + V3<R>{thr.x * L[1] * w, thr.y * L[2] * w, thr.z * L[3] * w};
                        }
                    }
                }
                if (mat_id < 0 || m[13] != R(0))
                {
                    if (mat_id >= 0 && pdf_prev == R(-2))
                        other = other + V3<R>{thr.x * m[0], thr.y * m[1], thr.z * m[2]}; // an emitter the camera sees
                    store_record(record_at(st, g, s_slot, depth, uint32_t(bnc)), V3<R>{R(0), R(0), R(0)}, other);
                    n_rec++;
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
                    V3<R> direct = {R(0), R(0), R(0)};
                    bool go_on = S.enable_indirect != 0;
                    R pdf_next = pdf_prev;
                    if (m[12] != R(0))
                    {
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
                        int k_test = 0; // lights with a valid sample met so far
                        for (int l = 0; l < S.n_lights; l++)
                        {
                            const R *L;
                            if constexpr (TAB)
                                L = lds_lights + l * LIGHT_STRIDE;
                            else
                                L = S.lights + (size_t)l * LIGHT_STRIDE;
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
                            if (tested == 0)
                            {
                                defer_mask |= 1u << l; // first visit: only find out which tests this hit needs
                                continue;
                            }
                            const bool blocked = ((state >> (RING_RESULT_SHIFT + k_test)) & 1u) != 0u;
                            k_test++;
                            if (blocked)
                                continue;
                            const R w = (go_on && bnc + 1 < depth) ? cn / (pl + pb) : cn / pl;
                            direct = direct + V3<R>{thr.x * L[1] * f.x * w, thr.y * L[2] * f.y * w, thr.z * L[3] * f.z * w};
                        }
                        if (defer_mask != 0u)
                            defer = true, defer_state = state; // out they go, all at once
                        if (!defer && go_on)
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
                    if (!defer)
                    {
                        store_record(record_at(st, g, s_slot, depth, uint32_t(bnc)), direct, other);
                        n_rec++;
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
                            Q.s_thr[s_slot] = thr.x, Q.s_thr[RING_SLOTS + s_slot] = thr.y, Q.s_thr[2 * RING_SLOTS + s_slot] = thr.z;
                            Q.s_pdf[s_slot] = pdf_next;
                            Q.s_org[s_slot] = p.x, Q.s_org[RING_SLOTS + s_slot] = p.y, Q.s_org[2 * RING_SLOTS + s_slot] = p.z;
                        }
                        else
                            finished = true, fin_state = uint32_t(bnc + 1) << 16;
                    }
                    else
                        n_tests += uint32_t(__popc(defer_mask));
                }
            }
            else if (mat_id < 0)
                finished = true, fin_state = uint32_t(bnc) << 16;
            else if (m[13] != R(0))
                finished = true, fin_state = (uint32_t(bnc) << 16) | uint32_t(mat_id + 1);
            else
            {
                // ---- shading normal ------------------------------------------------------------------
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
                // ---- NEE over all lights, in order.  Sphere-light segments are never tested; the square
                // light's is: a fresh hit with a valid sample goes back out as a visibility test -----------
                V3<R> direct = {R(0), R(0), R(0)};
                int k_test = 0; // square lights with a valid sample met so far
                for (int l = 0; l < S.n_lights; l++)
                {
                    const R *L;
                    if constexpr (TAB)
                        L = lds_lights + l * LIGHT_STRIDE;
                    else
                        L = S.lights + (size_t)l * LIGHT_STRIDE;
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
                    if (square)
                    {
                        if (tested == 0)
                        {
                            defer_mask |= 1u << l; // first visit: only find out which tests this hit needs
                            continue;
                        }
                        const bool blocked = ((state >> (RING_RESULT_SHIFT + k_test)) & 1u) != 0u;
                        k_test++;
                        if (blocked)
                            continue;
                    }
                    const V3<R> f = bxdf_eval(m, rd, wo, nrm);
                    const R w = cn / pdf;
                    direct = direct + V3<R>{(L[1] * w) * f.x, (L[2] * w) * f.y, (L[3] * w) * f.z};
                }
                if (defer_mask != 0u)
                    defer = true, defer_state = state; // out they go, all at once
                if (!defer)
                {
                    // ---- next direction and indirect coefficient -----------------------------------------
                    V3<R> coef = {R(0), R(0), R(0)};
                    if (S.enable_indirect)
                    {
                        const R x1 = rng_draw<R>(key, sample, bnc, 4 * S.n_lights + 0);
                        const R x2 = rng_draw<R>(key, sample, bnc, 4 * S.n_lights + 1);
                        bxdf_sample(m, nrm, wo, x1, x2, wi, coef);
                    }
                    store_record(record_at(st, g, s_slot, depth, uint32_t(bnc)), direct, coef);
                    n_rec++;
                    if (S.enable_indirect && bnc + 1 < depth)
                        cont = true;
                    else
                        finished = true, fin_state = uint32_t(bnc + 1) << 16;
                }
                else
                    n_tests += uint32_t(__popc(defer_mask));
            }
        }
        hq_head = (hq_head + n) % RING_CAP;
        hq_count -= n;
        // continuation rays -> ray ring
        {
            const unsigned long long cm = __ballot(cont);
            if (cont)
            {
                const uint32_t r = (rq_head + rq_count + rank_in(cm)) % RQ;
                Q.ro[r] = p.x, Q.ro[RQ + r] = p.y, Q.ro[2 * RQ + r] = p.z;
                Q.rd[r] = wi.x, Q.rd[RQ + r] = wi.y, Q.rd[2 * RQ + r] = wi.z;
                Q.r_slot[r] = s_slot, Q.r_bounce[r] = s_bounce + 1u;
                if (SQ)
                    Q.r_tri[r] = keep_tri, Q.r_light[r] = -1;
            }
            rq_count += uint32_t(__popcll(cm));
        }
        // visibility tests -> ray ring, one entry per test (the hit: point, incoming direction,
        // triangle; which light, which of the hit's tests, how many it has), so that the tests of a
        // hit run on as many lanes at once; round j writes the j-th test of every lane that has one
        if constexpr (SQ)
        {
            uint32_t rest = defer ? defer_mask : 0u;
            const uint32_t n_mine = uint32_t(__popc(rest));
            if (defer)
                join[s_slot] = 0u;
            for (uint32_t j = 0;; j++)
            {
                const bool has = rest != 0u;
                const unsigned long long hm = __ballot(has);
                if (hm == 0ull)
                    break;
                if (has)
                {
                    const int l = __builtin_ctz(rest);
                    rest &= rest - 1u;
                    const uint32_t r = (rq_head + rq_count + rank_in(hm)) % RQ;
                    Q.ro[r] = p.x, Q.ro[RQ + r] = p.y, Q.ro[2 * RQ + r] = p.z;
                    Q.rd[r] = keep_d.x, Q.rd[RQ + r] = keep_d.y, Q.rd[2 * RQ + r] = keep_d.z;
                    Q.r_slot[r] = s_slot, Q.r_bounce[r] = defer_state;
                    Q.r_tri[r] = keep_tri, Q.r_light[r] = int(uint32_t(l) | (j << 16) | (n_mine << 20));
                }
                rq_count += uint32_t(__popcll(hm));
            }
        }
        // finished paths -> resolve
        {
            const unsigned long long fm = __ballot(finished);
            if (finished)
            {
                const uint32_t r = (dq_head + dq_count + rank_in(fm)) % DoneQueue::CAP;
                dq.pid[r] = s_pid, dq.state[r] = fin_state, dq.need[r] = 0u, dq.slot[r] = uint16_t(s_slot);
            }
            dq_count += uint32_t(__popcll(fm));
            if (dq_count >= 64u)
                resolve(64u);
        }
        PTX_PHASE_END(3);
    };

    for (;;)
    {
        // ---- traversal steps until a batch of lanes has a hit to hand over or is free for a ray ----
        int n_busy, n_done;
        for (;;)
        {
            n_busy = __popcll(__ballot(busy));
            n_done = __popcll(__ballot(busy && done));
            const int n_trav = n_busy - n_done;
            if (n_trav == 0 || n_done >= RING_BATCH)
                break;
            if (64 - n_busy >= RING_BATCH && (rq_count > 0u || (!drained && rq_count + hq_count + uint32_t(n_busy) <= uint32_t(RING_ADMIT))))
                break;
            const bool act = busy && !done;
            const bool at_tri = act && T.cur >= S.n_nodes;
            const int n_tri = __popcll(__ballot(at_tri));
            if (n_tri * 2 > (n_trav - n_tri) * 3)
            {
                if (at_tri)
                    done = !bvh_step<R, SQ ? 2 : 0, 2>(S, o, d, T, stk, n_visit);
                PTX_PHASE_END(1);
            }
            else
            {
                if (act && !at_tri)
                    done = !bvh_step<R, SQ ? 2 : 0, 1>(S, o, d, T, stk, n_visit);
                PTX_PHASE_END(0);
            }
        }
        PTX_PHASE_END(7);
        // ---- completed traversals -> hit ring -----------------------------------------------------
        if (n_done >= RING_BATCH || (n_done > 0 && n_busy == n_done))
        {
            const bool give = busy && done;
            const unsigned long long gm = __ballot(give);
            // a finished visibility test adds its result to its hit's word; the last of a hit's tests
            // hands the hit back, all results in its state word
            bool hand = give;
            uint32_t word = bounce;
            if constexpr (SQ)
            {
                if (give && T.any != 0)
                {
                    const uint32_t k = (vcode >> 16) & 15u, of = (vcode >> 20) & 15u;
                    const uint32_t add = (1u << 16) | (T.target == 1 ? 1u << k : 0u);
                    const uint32_t old = atomicAdd(&join[slot], add);
                    hand = (old >> 16) + 1u == of;
                    word = (bounce & RING_BOUNCE) | (of << RING_TESTS_SHIFT) | (((old + add) & 0xFFFu) << RING_RESULT_SHIFT);
                }
            }
            const unsigned long long pm = __ballot(hand);
            if (hand)
            {
                const uint32_t r = (hq_head + hq_count + rank_in(pm)) % RING_CAP;
                const bool test = SQ && T.any != 0;
                const V3<R> p = test ? vp : o + T.best_t * d;
                const V3<R> hd = test ? vd : d;
                Q.hp[r] = p.x, Q.hp[RING_CAP + r] = p.y, Q.hp[2 * RING_CAP + r] = p.z;
                Q.hd[r] = hd.x, Q.hd[RING_CAP + r] = hd.y, Q.hd[2 * RING_CAP + r] = hd.z;
                Q.h_tri[r] = test ? vtri : T.target;
                Q.h_slot[r] = slot;
                Q.h_bounce[r] = word;
            }
            if (give)
                busy = false, done = false;
            hq_count += uint32_t(__popcll(pm));
            n_busy -= __popcll(gm);
        }
        PTX_PHASE_END(2);
        // ---- 64 hits wait: shade them (fewer only when nothing else is left to do) -----------------
        // (with visibility tests a pass may add an entry per tested light and hit to the ray ring: it
        // waits for that much room, while the lanes go on taking rays)
        if (hq_count >= 64u && (!SQ || rq_count + 64u * uint32_t(FAST ? S.n_lights : S.n_square) <= uint32_t(RQ)))
            shade(64u);
        else if (hq_count > 0u && n_busy == 0 && rq_count == 0u && drained)
            shade(hq_count);
        // ---- camera rays enter through the ray ring, 64 at a time ----------------------------------
        if (!drained && rq_count < 64u && rq_count + hq_count + uint32_t(n_busy) <= uint32_t(RING_ADMIT))
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
            const uint32_t cnt = end - next < 64u ? end - next : 64u;
            if (uint32_t(lane) < cnt)
            {
                const uint32_t id = next + uint32_t(lane);
                const uint32_t pix = id % n_pix;
                const int lr = B.row0 + int(pix / n_cols), col = B.col0 + int(pix % n_cols);
                const int row = local_row_to_image_row(B, lr);
                V3<R> co, cd;
                primary_ray<R>(cam, col, row, co, cd);
                const uint64_t key = rng_pixel_key(B.seed, uint32_t(row) * uint32_t(B.image_width) + uint32_t(col));
                const uint32_t r = (rq_head + rq_count + uint32_t(lane)) % RQ;
                Q.ro[r] = cam_pos.x, Q.ro[RQ + r] = cam_pos.y, Q.ro[2 * RQ + r] = cam_pos.z;
                Q.rd[r] = cd.x, Q.rd[RQ + r] = cd.y, Q.rd[2 * RQ + r] = cd.z;
                const uint32_t sl = free_slot[free_count - 1u - uint32_t(lane)];
                Q.r_slot[r] = sl, Q.r_bounce[r] = 0u;
                Q.s_pid[sl] = id, Q.s_klo[sl] = uint32_t(key), Q.s_khi[sl] = uint32_t(key >> 32);
                if (SQ)
                    Q.r_light[r] = -1;
            }
            free_count -= cnt;
            next += cnt;
            rq_count += cnt;
        }
        PTX_PHASE_END(5);
        // ---- free lanes take rays --------------------------------------------------------------------
        if (rq_count > 0u && (64 - n_busy >= RING_BATCH || n_busy == 0 || __popcll(__ballot(busy && !done)) == 0))
        {
            ring_sync();
            const unsigned long long fm = __ballot(!busy);
            const uint32_t rk = rank_in(fm);
            if (!busy && rk < rq_count)
            {
                const uint32_t r = (rq_head + rk) % RQ;
                o = {Q.ro[r], Q.ro[RQ + r], Q.ro[2 * RQ + r]};
                d = {Q.rd[r], Q.rd[RQ + r], Q.rd[2 * RQ + r]};
                bounce = Q.r_bounce[r], slot = Q.r_slot[r];
                const int test_code = SQ ? Q.r_light[r] : -1;
                const int test_light = test_code & 0xFFFF;
                if (SQ && test_code >= 0)
                {
                    vcode = uint32_t(test_code);
                    // a visibility test: the entry is the hit (point, incoming direction, triangle); the
                    // segment runs from the light sample point, a function of the path's random numbers,
                    // to the shaded point, as in visible()
                    vp = o, vd = d, vtri = Q.r_tri[r];
                    const R *L;
                    if constexpr (TAB)
                        L = lds_lights + test_light * LIGHT_STRIDE;
                    else
                        L = S.lights + (size_t)test_light * LIGHT_STRIDE;
                    const uint64_t key = (uint64_t(Q.s_khi[slot]) << 32) | Q.s_klo[slot];
                    const uint32_t sample = uint32_t(B.sample0) + Q.s_pid[slot] / n_pix;
                    const R x1 = rng_draw<R>(key, sample, bounce & RING_BOUNCE, 4 * test_light + 0);
                    const R x2 = rng_draw<R>(key, sample, bounce & RING_BOUNCE, 4 * test_light + 1);
                    V3<R> s;
                    if constexpr (FAST)
                    {
                        V3<R> unused;
                        (void)fast_light_sample(L, vp, x1, x2, unused, s); // the same point the shading pass computed
                    }
                    else
                        s = square_light_point(L, x1, x2);
                    const R dist = norm(s - vp);
                    o = s, d = normalized(vp - s);
                    trav_init(S, T, o, d, dist);
                    T.any = 1;
                }
                else
                    trav_init(S, T, o, d);
                busy = true, done = false;
            }
            const uint32_t taken = min(uint32_t(__popcll(fm)), rq_count);
            rq_head = (rq_head + taken) % RQ;
            rq_count -= taken;
            if (rq_count == 0u)
                rq_head = 0u; // an empty ring starts over: the entries in use stay in a few cache lines
            n_busy += int(taken);
        }
        PTX_PHASE_END(6);
        if (n_busy == 0 && rq_count == 0u && hq_count == 0u && drained)
            break;
    }
    while (dq_count > 0u)
        resolve(dq_count < 64u ? dq_count : 64u);
#ifdef PTX_PHASE_TIMING
    PTX_PHASE_END(7);
    if (lane == 0 && g % 500 == 3)
        printf("wave %d cycles: node %llu (%u) tri %llu (%u) handover %llu (%u) shade %llu (%u) resolve %llu (%u) admit %llu (%u) take %llu (%u) other %llu (%u)\n", g,
               ph_t[0], ph_n[0], ph_t[1], ph_n[1], ph_t[2], ph_n[2], ph_t[3], ph_n[3], ph_t[4], ph_n[4], ph_t[5], ph_n[5], ph_t[6], ph_n[6], ph_t[7], ph_n[7]);
#endif
    // per-wave statistics (the wave owns its row)
    unsigned long long tot[5] = {n_seg, n_rec, n_visit[0], n_visit[1], n_tests};
    for (int off = 32; off > 0; off >>= 1)
        for (int c = 0; c < 5; c++)
            tot[c] += __shfl_down(tot[c], off, 64);
    if (lane == 0)
    {
        st.wave_stats[g * WAVE_STATS + 0] = tot[0];
        st.wave_stats[g * WAVE_STATS + 1] = tot[4];
        st.wave_stats[g * WAVE_STATS + 2] = n_iter;
        st.wave_stats[g * WAVE_STATS + 3] = tot[1];
        st.wave_stats[g * WAVE_STATS + 4] = tot[2];
        st.wave_stats[g * WAVE_STATS + 5] = tot[3];
    }
}

// workgroups per CU the reference-estimator f64 ring kernel is compiled for (168 registers at 3)
#ifndef PTX_RINGS_F64_BLOCKS
#define PTX_RINGS_F64_BLOCKS 3
#endif
template <bool TAB>
__global__ void __launch_bounds__(BLOCK, 2) k_trace_rings_fast_f64(DScene<double> S, DCamera cam, DBatch B, DState<double> st, int chunk)
{
    trace_body_rings<double, TAB, true, true>(S, cam, B, st, chunk);
}
template <bool TAB>
__global__ void __launch_bounds__(BLOCK, 3) k_trace_rings_fast_f32(DScene<float> S, DCamera cam, DBatch B, DState<float> st, int chunk)
{
    trace_body_rings<float, TAB, true, true>(S, cam, B, st, chunk);
}
template <bool TAB, bool SQ>
__global__ void __launch_bounds__(BLOCK, PTX_RINGS_F64_BLOCKS) k_trace_rings_f64(DScene<double> S, DCamera cam, DBatch B, DState<double> st, int chunk)
{
    trace_body_rings<double, TAB, SQ>(S, cam, B, st, chunk);
}
template <bool TAB, bool SQ>
__global__ void __launch_bounds__(BLOCK, 4) k_trace_rings_f32(DScene<float> S, DCamera cam, DBatch B, DState<float> st, int chunk)
{
    trace_body_rings<float, TAB, SQ>(S, cam, B, st, chunk);
}

} // namespace ptx
