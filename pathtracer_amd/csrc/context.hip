// Context: device scene, per-batch path state and the render loop (HIP, gfx950).
//
// Replaces cPathTracer::Init / Process / RayTracing (src/tracing/cPathTracer.cpp:20-57, 65-86,
// 517-553 of the reference): instead of an OpenMP loop over rows, a batch of paths (pixels x
// samples) is traced by one persistent kernel in which every lane carries one path at a time and
// finished paths are resolved (their (direct, coefficient) records folded back to front) 64 at a
// time, and an accumulate kernel adds the per-sample results to the per-pixel sums in sample order.
#include "context.hpp"
#include "kernels.hpp"
#include "plane_records.hpp"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <stdexcept>
#include <string>

namespace ptx
{

#define HIP_CHECK(expr)                                                                                  \
    do                                                                                                   \
    {                                                                                                    \
        hipError_t e_ = (expr);                                                                          \
        if (e_ != hipSuccess)                                                                            \
            throw std::runtime_error(std::string("device: ") + #expr + ": " + hipGetErrorString(e_));   \
    } while (0)

namespace
{

struct DeviceBuffer
{
    void *p = nullptr;
    size_t bytes = 0;
    void alloc(size_t n)
    {
        release();
        if (n == 0)
            n = 16;
        HIP_CHECK(hipMalloc(&p, n));
        bytes = n;
    }
    void release()
    {
        if (p)
            (void)hipFree(p);
        p = nullptr;
        bytes = 0;
    }
    ~DeviceBuffer() { release(); }
    DeviceBuffer() = default;
    DeviceBuffer(const DeviceBuffer &) = delete;
    DeviceBuffer &operator=(const DeviceBuffer &) = delete;
};

template <class T> void upload(DeviceBuffer &b, const std::vector<T> &v, hipStream_t s)
{
    b.alloc(v.size() * sizeof(T));
    if (!v.empty())
        HIP_CHECK(hipMemcpyAsync(b.p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice, s));
}

constexpr int MAX_TRACE_WAVES = 1 << 16; // rows of the per-wave statistics table

// The dynamic-LDS limit of a kernel is a property of the kernel function, shared by every context of
// the process, and contexts may plan and launch from several host threads (one per GPU, or several
// on one GPU): the limit is only ever RAISED (a launch asks for what its own plan says, which the
// limit then still allows), and planning is serialised.
std::mutex g_lds_limit_mutex;
std::map<std::pair<int, const void *>, int> g_lds_limit; // (device, kernel) -> limit set so far
void raise_lds_limit(const void *kernel, int bytes)
{
    int dev = 0;
    HIP_CHECK(hipGetDevice(&dev));
    int &cur = g_lds_limit[{dev, kernel}];
    if (bytes > cur)
    {
        HIP_CHECK(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        cur = bytes;
    }
}

// dynamic LDS of a BVH kernel (its traversal stacks); large requests must be announced
template <class K> inline void allow_dynamic_lds(K kernel, size_t bytes)
{
    if (bytes > 0)
    {
        std::lock_guard<std::mutex> lock(g_lds_limit_mutex);
        raise_lds_limit(reinterpret_cast<const void *>(kernel), static_cast<int>(bytes));
    }
}

inline unsigned blocks_for(int64_t n) { return static_cast<unsigned>(std::max<int64_t>(1, (n + BLOCK - 1) / BLOCK)); }

} // namespace

// what layout_state sizes the state region for
struct StateNeeds
{
    int64_t paths = 0;     // paths per batch
    int depth = 1;
    int waves = 0;         // waves of the tracer's grid
    int slots = LANE_SLOTS; // record slots per wave
    size_t ring_bytes = 0; // per wave (ring tracer)
    int spill_levels = 0;  // traversal stack levels kept in global memory, per lane
    int ring_batch = 40;   // ring tracer: lanes per hand-over
    bool wavefront = false; // wavefront tracer: two ray queues, records and end states by path id instead of record slots
    int wf_tests = 0;       // ... with visibility tests (the physically based estimator): test slots per segment = lights
};

// State region of ptx_render_static: device memory the library owns from load to unload, one
// copy per GPU, so that a render call through that entry point allocates nothing.  Host code must
// not use the array's name as an address (it is a host-side stand-in): the device address comes
// from hipGetSymbolAddress, once per device, in Context::Context.
constexpr size_t STATIC_STATE_BYTES = size_t(256) << 20;
static_assert(WF_MAX_LIGHTS >= RING_MAX_TESTS && WF_MAX_LIGHTS <= 32 && WF_QUEUE_HEADER_WORDS * 4 <= 256, "the test queue's header holds a list length per light");
static __device__ __align__(256) unsigned char scratch[STATIC_STATE_BYTES];

// per-precision device tables + per-batch path state
// device-side pieces of one wavefront queue carved out of a buffer
template <class R> static size_t wf_queue_bytes(int64_t cap) { return static_cast<size_t>(cap) * (7 * sizeof(R) + 8) + 256; }
template <class R> static WfQueue<R> wf_queue_at(char *base, int64_t cap)
{
    WfQueue<R> q;
    q.cap = cap;
    q.count = reinterpret_cast<uint32_t *>(base);
    q.pool = q.count + 1;
    R *f = reinterpret_cast<R *>(base + 256);
    q.o = f, q.d = f + 3 * cap, q.t = f + 6 * cap;
    q.pid = reinterpret_cast<uint32_t *>(f + 7 * cap);
    q.tri = reinterpret_cast<int32_t *>(q.pid + cap);
    return q;
}
template <class R> struct Context::Impl
{
    DeviceBuffer tri_pre, mats, lights, tri_nrm;
    DeviceBuffer bvh; // triangle records of the 4-wide BVH
    DeviceBuffer state_mem; // one slab carved into the arrays of a render call (layout_state)
    DState<R> st;
    WfQueue<R> wfq[2]; // wavefront tracer: the two ray queues ...
    WfQueue<R> wft;    // ... the queue of visibility tests (physically based estimator) ...
    WfState<R> wfs;    // ... and the per-path records and end states
    bool ready = false;

    void build_tables(const Scene &sc, const Bvh &bvh, bool brute, hipStream_t s)
    {
        // intersector order: the faces themselves (all-triangles intersector) or the references of the
        // BVH's leaves (a face cut by reference splitting appears once per reference)
        const int F = brute ? sc.n_tris() : static_cast<int>(bvh.tri_order.size());
        std::vector<R> pre(static_cast<size_t>(std::max(F, 1)) * 12, R(0));
        for (int k = 0; k < F; k++)
        {
            const int f = brute ? k : bvh.tri_order[k];
            const float *v = &sc.verts[static_cast<size_t>(f) * 9];
            // columns of the reference's 3x3 system, computed in R exactly as the reference does
            R p1[3] = {R(v[0]), R(v[1]), R(v[2])}, a[3], b[3];
            for (int c = 0; c < 3; c++)
            {
                a[c] = p1[c] - R(v[3 + c]);
                b[c] = p1[c] - R(v[6 + c]);
            }
            R *o = &pre[static_cast<size_t>(k) * 12];
            for (int c = 0; c < 3; c++)
                o[c] = p1[c], o[3 + c] = a[c], o[6 + c] = b[c];
            R t0 = a[1] * b[2], t1 = b[1] * a[2];
            o[9] = t0 - t1;
            t0 = a[2] * b[0], t1 = b[2] * a[0];
            o[10] = t0 - t1;
            t0 = a[0] * b[1], t1 = b[0] * a[1];
            o[11] = t0 - t1;
        }
        upload(tri_pre, pre, s);

        // triangle records of the BVH, intersector order: the nine float coordinates, face id, two unused words
        {
            const int stride = TRI_REC_Q * 4; // dwords
            std::vector<uint32_t> recs(static_cast<size_t>(std::max(brute ? 0 : F, 1)) * stride, 0u);
            if (!brute)
                for (int k = 0; k < F; k++)
                {
                    uint32_t *o = &recs[static_cast<size_t>(k) * stride];
                    std::memcpy(o, &sc.verts[static_cast<size_t>(bvh.tri_order[k]) * 9], 9 * sizeof(float));
                    o[9] = static_cast<uint32_t>(bvh.tri_order[k]);
                }
            upload(this->bvh, recs, s);
        }

        std::vector<R> m(static_cast<size_t>(std::max<size_t>(sc.mats.size(), 1)) * MAT_STRIDE, R(0));
        for (size_t i = 0; i < sc.mats.size(); i++)
        {
            const Material &src = sc.mats[i];
            R *o = &m[i * MAT_STRIDE];
            for (int c = 0; c < 3; c++)
                o[c] = R(src.Ka[c]), o[3 + c] = R(src.Kd[c]), o[6 + c] = R(src.Ks[c]);
            o[9] = R(src.Ni), o[10] = R(src.Ns);
            o[11] = R(std::pow((1.0 - src.Ni) / (1.0 + src.Ni), 2));           // Schlick R0
            o[12] = (std::fabs(src.Ni - 1.0) < 1e-5) ? R(0) : R(1);              // BSDF when Ni != 1
            double ka = std::sqrt(src.Ka[0] * src.Ka[0] + src.Ka[1] * src.Ka[1] + src.Ka[2] * src.Ka[2]);
            o[13] = ka > 1e-6 ? R(1) : R(0);                                     // emitter
            o[14] = (src.Ns >= 0 && src.Ns <= 1048576.0 && std::floor(src.Ns) == src.Ns && double(R(src.Ns)) == src.Ns) ? R(1) : R(0); // whole-number Ns
        }
        upload(mats, m, s);

        std::vector<R> L(static_cast<size_t>(std::max<size_t>(sc.lights.size(), 1)) * LIGHT_STRIDE, R(0));
        for (size_t i = 0; i < sc.lights.size(); i++)
        {
            const Light &src = sc.lights[i];
            R *o = &L[i * LIGHT_STRIDE];
            o[0] = R(src.type);
            for (int c = 0; c < 3; c++)
                o[1 + c] = R(src.radiance[c]);
            if (src.type == 0)
            {
                double e10[3], e21[3], n[3];
                for (int c = 0; c < 3; c++)
                {
                    e10[c] = src.p[1][c] - src.p[0][c];
                    e21[c] = src.p[2][c] - src.p[1][c];
                }
                double t0 = e10[1] * e21[2], t1 = e10[2] * e21[1];
                n[0] = t0 - t1;
                t0 = e10[2] * e21[0], t1 = e10[0] * e21[2];
                n[1] = t0 - t1;
                t0 = e10[0] * e21[1], t1 = e10[1] * e21[0];
                n[2] = t0 - t1;
                double z = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
                if (z > 0)
                {
                    double nn = std::sqrt(z);
                    n[0] /= nn, n[1] /= nn, n[2] /= nn;
                }
                double l21 = std::sqrt(e21[0] * e21[0] + e21[1] * e21[1] + e21[2] * e21[2]);
                double l10 = std::sqrt(e10[0] * e10[0] + e10[1] * e10[1] + e10[2] * e10[2]);
                for (int c = 0; c < 3; c++)
                    o[4 + c] = R(src.p[0][c]), o[7 + c] = R(e10[c]), o[10 + c] = R(e21[c]), o[13 + c] = R(n[c]);
                o[16] = R(l21 * l10);
            }
            else
            {
                for (int c = 0; c < 3; c++)
                    o[4 + c] = R(src.center[c]);
                o[7] = R(src.radius);
            }
        }
        upload(lights, L, s);

        // per-face shading normal where it does not depend on the hit point: flat faces (three equal
        // vertex normals) when barycentric normals are on, the geometric normal when they are off
        // (or when the OBJ carries no normals at all)
        std::vector<R> fn(static_cast<size_t>(std::max(F, 1)) * 4, R(0));
        const bool bary = sc.enable_bary_normal && sc.has_normals;
        for (int k = 0; k < F; k++)
        {
            const int f = brute ? k : bvh.tri_order[k];
            const float *v = &sc.verts[static_cast<size_t>(f) * 9];
            const float *n = &sc.normals[static_cast<size_t>(f) * 9];
            double g[3];
            bool use = false;
            if (!bary)
            {
                double e1[3], e2[3];
                for (int c = 0; c < 3; c++)
                {
                    e1[c] = double(v[3 + c]) - double(v[c]);
                    e2[c] = double(v[6 + c]) - double(v[3 + c]);
                }
                g[0] = e1[1] * e2[2] - e1[2] * e2[1];
                g[1] = e1[2] * e2[0] - e1[0] * e2[2];
                g[2] = e1[0] * e2[1] - e1[1] * e2[0];
                use = true;
            }
            else if (std::memcmp(n, n + 3, 12) == 0 && std::memcmp(n, n + 6, 12) == 0 && !std::isnan(n[0]) &&
                     !std::isnan(n[1]) && !std::isnan(n[2]))
            {
                g[0] = n[0], g[1] = n[1], g[2] = n[2];
                use = true;
            }
            if (use)
            {
                double z = g[0] * g[0] + g[1] * g[1] + g[2] * g[2];
                double inv = z > 0 ? 1.0 / std::sqrt(z) : 1.0;
                R *o = &fn[static_cast<size_t>(k) * 4];
                o[0] = R(g[0] * inv), o[1] = R(g[1] * inv), o[2] = R(g[2] * inv), o[3] = R(1);
            }
        }
        upload(tri_nrm, fn, s);
        ready = true;
    }

    // Everything a render call keeps in device memory besides the scene and the image, carved out
    // of ONE region (`base`; null: only measure): the record slots of the launch's waves, the
    // per-path results of a batch, statistics, the path-id pool, and for BVH scenes the rings of the
    // ring tracer and the traversal stack levels the LDS does not hold.  Returns the bytes needed.
    size_t layout_state(char *base, const StateNeeds &n, uint32_t **spill)
    {
        size_t total = 0;
        auto take = [&](size_t bytes) {
            const size_t off = total;
            total += (bytes + 255) & ~size_t(255);
            return base ? base + off : nullptr;
        };
        const size_t N = static_cast<size_t>(n.paths);
        st.records = n.wavefront ? nullptr : reinterpret_cast<R *>(take(static_cast<size_t>(n.waves) * n.slots * n.depth * REC_STRIDE * sizeof(R)));
        st.result = reinterpret_cast<R *>(take(N * 3 * sizeof(R)));
        if (n.wavefront)
        {
            for (int q = 0; q < 2; q++)
            {
                char *at = take(wf_queue_bytes<R>(n.paths));
                if (base)
                    wfq[q] = wf_queue_at<R>(at, n.paths);
            }
            wfs.rec = reinterpret_cast<R *>(take(N * 6 * static_cast<size_t>(n.depth) * sizeof(R)));
            wfs.fin = reinterpret_cast<uint32_t *>(take(N * sizeof(uint32_t)));
            wfs.result = st.result;
            wfs.N = n.paths;
            wfs.rows = std::max(n.waves, 1);
            wfs.thr = wfs.pdf = wfs.term = nullptr;
            if (n.wf_tests > 0)
            {
                const int64_t tcap = n.paths * n.wf_tests;
                char *at = take(wf_queue_bytes<R>(tcap));
                if (base)
                    wft = wf_queue_at<R>(at, tcap);
                wfs.term = reinterpret_cast<R *>(take(static_cast<size_t>(tcap) * 3 * sizeof(R)));
                wfs.thr = reinterpret_cast<R *>(take(N * 3 * sizeof(R)));
                wfs.pdf = reinterpret_cast<R *>(take(N * sizeof(R)));
            }
        }
        st.wave_stats = reinterpret_cast<unsigned long long *>(take(static_cast<size_t>(n.waves) * WAVE_STATS * sizeof(unsigned long long)));
        st.stats = reinterpret_cast<unsigned long long *>(take(sizeof(unsigned long long) * 8));
        st.pool = reinterpret_cast<unsigned int *>(take(sizeof(unsigned int)));
        st.rings = n.ring_bytes ? take(static_cast<size_t>(n.waves) * n.ring_bytes) : nullptr;
        *spill = n.spill_levels > 0 ? reinterpret_cast<uint32_t *>(take(static_cast<size_t>(n.waves) * 64 * n.spill_levels * sizeof(uint32_t))) : nullptr;
        st.N = n.paths;
        st.n_waves = 0;
        st.slots = n.slots;
        st.ring_batch = n.ring_batch;
        if (n.wavefront)
            wfs.wave_stats = st.wave_stats;
        return total;
    }

    // grows the slab to `bytes`; the old one is only let go once the new one exists
    void reserve_state(size_t bytes)
    {
        if (bytes <= state_mem.bytes)
            return;
        // nothing in the region is live between two calls, and the caller's budget counts the old region
        // as reusable: release it before the larger one is asked for
        state_mem.release();
        state_mem.alloc(bytes);
    }
};

struct Context::Geometry
{
    DeviceBuffer tri_face, tri_mat, verts, normals, mat_ids, nodes;
    DeviceBuffer tri_pq, rec_tri, tri_rec, rec_block; // plane records of the all-triangles intersector
    int n_recs = 0;
    float smax = 0.f;
    int n_nodes = 0, id_bits = 1; // 4-wide BVH: node records, bits of the record id in a stack entry
    int stack_need = 1;           // stack entries a traversal can need
    float cmax = 0.f;             // largest |coordinate| of the root node's child boxes
};

// 64-byte device nodes: corner and cell size of an 8-bit grid over the union of the child boxes,
// the child boxes in grid coordinates (rounded outward), the child record ids (nodes first, then
// triangles; 0xFFFFFFFF: no child).  Word layout: corner.xyz, id0 | cell.xyz, id1 | x lo, x hi, y lo,
// y hi | z lo, z hi, id2, id3; the six coordinate words hold one byte per child.
static std::vector<uint32_t> pack_nodes(const Bvh &bvh, int n_tris)
{
    const size_t n_nodes = bvh.nodes.size();
    std::vector<uint32_t> out(std::max<size_t>(n_nodes, 1) * 16, 0u);
    for (size_t i = 0; i < n_nodes; i++)
    {
        const BvhNode &nd = bvh.nodes[i];
        uint32_t *o = &out[i * 16];
        uint32_t id[4];
        for (int j = 0; j < 4; j++)
        {
            const int32_t ch = nd.child[j];
            id[j] = ch == BVH_EMPTY ? 0xFFFFFFFFu : ch >= 0 ? static_cast<uint32_t>(ch) : static_cast<uint32_t>(n_nodes + static_cast<size_t>(~ch));
        }
        uint32_t q[3][2] = {{0, 0}, {0, 0}, {0, 0}};
        float corner[3], cell[3];
        for (int k = 0; k < 3; k++)
        {
            float lo = std::numeric_limits<float>::infinity(), hi = -std::numeric_limits<float>::infinity();
            for (int j = 0; j < 4; j++)
                if (nd.child[j] != BVH_EMPTY)
                    lo = std::min(lo, nd.lo[k][j]), hi = std::max(hi, nd.hi[k][j]);
            if (!(lo <= hi))
                lo = hi = 0.f; // a node without children
            // cell >= extent / 255 in exact arithmetic, so that coordinate 255 reaches the far side
            float cs = static_cast<float>((static_cast<double>(hi) - lo) / 255.0);
            while (static_cast<double>(lo) + 255.0 * cs < hi)
                cs = std::nextafter(cs, std::numeric_limits<float>::infinity());
            if (!(cs > 0.f))
                cs = std::numeric_limits<float>::min();
            corner[k] = lo, cell[k] = cs;
            for (int j = 0; j < 4; j++)
            {
                int ql = 255, qh = 0; // an absent child: an inverted slab
                if (nd.child[j] != BVH_EMPTY)
                {
                    // outward rounding, checked in exact (double) arithmetic on the float corner and cell
                    ql = static_cast<int>(std::floor((static_cast<double>(nd.lo[k][j]) - lo) / cs));
                    qh = static_cast<int>(std::ceil((static_cast<double>(nd.hi[k][j]) - lo) / cs));
                    ql = std::min(std::max(ql, 0), 255), qh = std::min(std::max(qh, 0), 255);
                    while (ql > 0 && static_cast<double>(lo) + static_cast<double>(ql) * cs > nd.lo[k][j])
                        ql--;
                    while (qh < 255 && static_cast<double>(lo) + static_cast<double>(qh) * cs < nd.hi[k][j])
                        qh++;
                    if (static_cast<double>(lo) + static_cast<double>(ql) * cs > nd.lo[k][j] ||
                        static_cast<double>(lo) + static_cast<double>(qh) * cs < nd.hi[k][j])
                        throw std::runtime_error("state: BVH node quantisation is not conservative");
                }
                q[k][0] |= static_cast<uint32_t>(ql) << (8 * j);
                q[k][1] |= static_cast<uint32_t>(qh) << (8 * j);
            }
        }
        std::memcpy(&o[0], corner, 12);
        o[3] = id[0];
        std::memcpy(&o[4], cell, 12);
        o[7] = id[1];
        o[8] = q[0][0], o[9] = q[0][1], o[10] = q[1][0], o[11] = q[1][1];
        o[12] = q[2][0], o[13] = q[2][1], o[14] = id[2], o[15] = id[3];
    }
    (void)n_tris;
    return out;
}


Context::Context(int device) : device_(device)
{
    int count = 0;
    hipError_t e = hipGetDeviceCount(&count);
    if (e != hipSuccess || count <= 0)
        throw std::runtime_error("device: no HIP device available (the path-tracing core needs a GPU; "
                                 "there is no CPU fallback)");
    if (device < 0 || device >= count)
        throw std::runtime_error("arg: hip_device out of range");
    HIP_CHECK(hipSetDevice(device));
    hipDeviceProp_t prop;
    HIP_CHECK(hipGetDeviceProperties(&prop, device));
    n_cus_ = std::max(prop.multiProcessorCount, 1);
    // this device's copy of the static state region (every GPU has its own, at its own address)
    HIP_CHECK(hipGetSymbolAddress(reinterpret_cast<void **>(&static_state_), HIP_SYMBOL(scratch)));
    // what a render call needs besides device memory is created here, not in the call: two events
    // and a pinned landing place for the statistics
    // (a constructor that throws runs no destructor: what was made so far is released here)
    try
    {
        HIP_CHECK(hipEventCreate(&ev_begin_));
        HIP_CHECK(hipEventCreate(&ev_end_));
        HIP_CHECK(hipHostMalloc(reinterpret_cast<void **>(&host_stats_), sizeof(unsigned long long) * 8, hipHostMallocDefault));
        f64_ = new Impl<double>();
        f32_ = new Impl<float>();
    }
    catch (...)
    {
        if (ev_begin_)
            (void)hipEventDestroy(ev_begin_);
        if (ev_end_)
            (void)hipEventDestroy(ev_end_);
        if (host_stats_)
            (void)hipHostFree(host_stats_);
        delete f64_;
        delete f32_;
        throw;
    }
}

Context::~Context()
{
    if (shard_)
        free_shard_slot(shard_); // joins the context's worker thread, frees its frame buffers
    (void)hipSetDevice(device_);
    delete f64_;
    delete f32_;
    delete geo_;
    if (ev_begin_)
        (void)hipEventDestroy(ev_begin_);
    if (ev_end_)
        (void)hipEventDestroy(ev_end_);
    if (host_stats_)
        (void)hipHostFree(host_stats_);
}

void Context::set_stream(void *s) { stream_ = s; }

void Context::build_accel(const Scene &sc)
{
    HIP_CHECK(hipSetDevice(device_));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    // a build that fails half way leaves the context without a scene (the host copy below and the
    // device tables of the previous scene would not belong together)
    have_scene_ = false;
    if (sc.lights.size() > 16383)
        throw std::runtime_error("arg: at most 16383 lights are supported (dimension field of the random-number counter)");
    scene_ = sc; // host copy (camera, tracer settings)
    // all-triangles intersector: small scenes whose plane records fit the candidate mask
    PlaneRecords pr;
    brute_ = sc.n_tris() <= std::min(brute_force_limit, BRUTE_MAX_TRIS);
    if (brute_)
    {
        build_plane_records(sc.verts.data(), sc.n_tris(), pr);
        brute_ = pr.n_recs <= BRUTE_MAX_RECS;
    }
    if (!brute_)
    {
        pr = PlaneRecords();
        pr.rec.assign(PLANE_REC_STRIDE, 0.f);
        pr.rec_tri.assign(64, -1);
        pr.tri_rec.assign(1, 0);
    }
    static_assert(PLANE_REC_STRIDE == PQ_STRIDE, "record stride");
    bvh_ = Bvh();
    build_bvh(sc.verts.data(), sc.n_tris(), bvh_, /*median=*/false, bvh_split / 100.0);
    if (!brute_ && bvh_.stack_need > BVH_STACK_MAX)
        build_bvh(sc.verts.data(), sc.n_tris(), bvh_, /*median=*/true); // balanced tree: depth ~ log4 n
    if (!brute_ && bvh_.stack_need > BVH_STACK_MAX)
        throw std::runtime_error("state: BVH deeper than the traversal stack");
    {
        // a stack entry keeps the record id in its low bits and the entry distance in the rest
        const uint64_t n_rec = bvh_.nodes.size() + static_cast<uint64_t>(bvh_.tri_order.size());
        id_bits_ = 1;
        while ((uint64_t(1) << id_bits_) < n_rec)
            id_bits_++;
        if (id_bits_ > 24)
            throw std::runtime_error("arg: scene too large (the BVH is limited to 2^24 nodes and triangles)");
    }
    if (static_cast<int>(sc.mats.size()) > MAX_MATS)
        throw std::runtime_error("arg: at most 65534 materials are supported");
    if (sc.depth > 65535)
        throw std::runtime_error("arg: Depth above 65535 is not supported");
    delete geo_;
    geo_ = new Geometry();
    // intersector order: faces (all-triangles intersector) or the triangle references of the BVH's leaves
    const int n_entries = brute_ ? sc.n_tris() : static_cast<int>(bvh_.tri_order.size());
    std::vector<int32_t> face(std::max(n_entries, 1), 0);
    for (int k = 0; k < n_entries; k++)
        face[k] = brute_ ? k : bvh_.tri_order[k];
    upload(geo_->tri_face, face, s);
    std::vector<int32_t> tmat(std::max(n_entries, 1), -1);
    for (int k = 0; k < n_entries; k++)
        tmat[k] = sc.mat_ids[face[k]];
    upload(geo_->tri_mat, tmat, s);
    upload(geo_->verts, sc.verts, s);
    upload(geo_->normals, sc.normals, s);
    upload(geo_->mat_ids, sc.mat_ids, s);
    upload(geo_->nodes, pack_nodes(bvh_, sc.n_tris()), s);
    upload(geo_->tri_pq, pr.rec, s);
    upload(geo_->rec_tri, pr.rec_tri, s);
    upload(geo_->tri_rec, pr.tri_rec, s);
    {
        // potential blockers per record towards light 0 (used when it is the scene's only light and square)
        std::vector<uint32_t> blk(32, 0xFFFFFFFFu);
        if (brute_ && sc.lights.size() == 1 && sc.lights[0].type == 0)
        {
            double corners[4][3];
            square_light_corners(sc.lights[0], corners);
            blk = blocker_masks(sc.verts.data(), pr, corners);
        }
        upload(geo_->rec_block, blk, s);
    }
    geo_->n_recs = pr.n_recs;
    geo_->n_nodes = static_cast<int>(bvh_.nodes.size());
    geo_->id_bits = id_bits_;
    geo_->stack_need = std::max(bvh_.stack_need, 1);
    {
        // the root's child boxes hold the whole scene (padded): bound of every |coordinate| a slab test meets
        float cm = 0.f;
        if (!bvh_.nodes.empty())
            for (int k = 0; k < 3; k++)
                for (int j = 0; j < 4; j++)
                    if (bvh_.nodes[0].child[j] != BVH_EMPTY)
                        cm = std::max(cm, std::max(std::fabs(bvh_.nodes[0].lo[k][j]), std::fabs(bvh_.nodes[0].hi[k][j])));
        geo_->cmax = cm;
    }
    geo_->smax = pr.smax;
    f64_->build_tables(sc, bvh_, brute_, s);
    f32_->build_tables(sc, bvh_, brute_, s);
    HIP_CHECK(hipStreamSynchronize(s));
    have_scene_ = true;
}

template <class R> static DScene<R> make_dscene(const Scene &sc, const Context::Geometry &g, const void *tri_pre,
                                                 const void *mats, const void *lights, const void *tri_nrm, const void *bvh, bool brute)
{
    DScene<R> S;
    S.n_tris = sc.n_tris();
    S.n_mats = static_cast<int>(sc.mats.size());
    S.n_lights = static_cast<int>(sc.lights.size());
    S.n_square = sc.n_square_lights();
    S.brute = brute;
    S.enable_indirect = sc.enable_indirect;
    S.enable_bary = sc.enable_bary_normal && sc.has_normals;
    S.n_recs = g.n_recs;
    S.pq_smax = g.smax;
    S.bvh_cmax = g.cmax;
    S.pad_[0] = S.pad_[1] = 0.f;
    S.tri_pre = static_cast<const R *>(tri_pre);
    S.tri_pq = static_cast<const float *>(g.tri_pq.p);
    S.rec_tri = static_cast<const int32_t *>(g.rec_tri.p);
    S.tri_rec = static_cast<const int32_t *>(g.tri_rec.p);
    S.rec_block = static_cast<const uint32_t *>(g.rec_block.p);
    S.tri_face = static_cast<const int32_t *>(g.tri_face.p);
    S.verts = static_cast<const float *>(g.verts.p);
    S.normals = static_cast<const float *>(g.normals.p);
    S.mat_ids = static_cast<const int32_t *>(g.mat_ids.p);
    S.mats = static_cast<const R *>(mats);
    S.lights = static_cast<const R *>(lights);
    S.tri_nrm = static_cast<const R *>(tri_nrm);
    S.tri_mat = static_cast<const int32_t *>(g.tri_mat.p);
    S.bvh_nodes = static_cast<const uint4 *>(g.nodes.p);
    S.bvh_tris = static_cast<const uint4 *>(bvh);
    S.n_nodes = g.n_nodes, S.id_bits = g.id_bits;
    S.stack_levels = g.stack_need; // the operator kernels keep the whole stack in LDS
    S.sq_light = -1, S.lds_nodes = 0;
    for (size_t l = 0; l < sc.lights.size() && S.sq_light < 0; l++)
        if (sc.lights[l].type == 0)
            S.sq_light = static_cast<int>(l);
    S.spill_stride = 0;
    S.spill = nullptr;
    return S;
}

static DCamera make_camera(const Scene &sc)
{
    DCamera c;
    for (int i = 0; i < 16; i++)
        c.m[i] = sc.raster2world[i];
    for (int i = 0; i < 3; i++)
        c.pos[i] = sc.cam_pos[i];
    c.width = sc.width, c.height = sc.height;
    return c;
}

void Context::require_scene() const
{
    if (!have_scene_)
        throw std::runtime_error("state: ptx_build_accel has not been called on this context");
}

void Context::generate_rays(double *ro, double *rd)
{
    require_scene();
    HIP_CHECK(hipSetDevice(device_));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const int64_t n = static_cast<int64_t>(scene_.width) * scene_.height;
    DeviceBuffer d_o, d_d;
    d_o.alloc(n * 24), d_d.alloc(n * 24);
    hipLaunchKernelGGL(k_camera_rays, dim3(blocks_for(n)), dim3(BLOCK), 0, s, make_camera(scene_), static_cast<double *>(d_o.p),
                       static_cast<double *>(d_d.p));
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(ro, d_o.p, n * 24, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipMemcpyAsync(rd, d_d.p, n * 24, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

// Occupancy and LDS split of a kernel that traverses the BVH.  The registers (and the static LDS)
// allow `by_regs` workgroups per CU.  The dynamic LDS of a workgroup holds its traversal stacks
// (1 KB per level; deeper levels spill to global memory) and, behind them, a copy of the first
// nodes of the tree (the builder numbers the busiest nodes first).  The occupancy is what the
// registers allow as long as that leaves a useful amount of LDS per workgroup; lds_stack_cap and
// lds_nodes_opt (options for tests and tuning) force the split.
struct BvhLdsPlan
{
    int per_cu = 1, levels = 0, nodes = 0;
    size_t dyn_bytes = 0;
};
static BvhLdsPlan plan_bvh_lds(const void *kernel, int want_per_cu, int stack_need, int lds_stack_cap, int n_nodes, int lds_nodes_opt)
{
    std::lock_guard<std::mutex> lock(g_lds_limit_mutex);
    BvhLdsPlan r;
    hipFuncAttributes fa;
    HIP_CHECK(hipFuncGetAttributes(&fa, kernel));
    int dev = 0, lds_cu = 0, by_regs = 1;
    HIP_CHECK(hipGetDevice(&dev));
    HIP_CHECK(hipDeviceGetAttribute(&lds_cu, hipDeviceAttributeMaxSharedMemoryPerMultiprocessor, dev));
    HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&by_regs, kernel, BLOCK, 0));
    if (want_per_cu > 0)
        by_regs = std::min(by_regs, want_per_cu);
    // below about 24 levels in LDS (16 next to cached nodes) the spills stop being the exception
    const int cap = lds_stack_cap > 0 ? std::min(lds_stack_cap, stack_need) : stack_need;
    const int useful = std::min(cap, lds_nodes_opt != 0 ? 16 : 24);
    const int level_bytes = BLOCK * 4;
    int avail = 0;
    for (int b = std::max(by_regs, 1); b >= 1; b--)
    {
        // LDS is handed out in granules of 1280 bytes on this device, which the calculator
        // does not round to (measured: 3 x 54272 bytes do not fit 160 KB, 3 x 53760 do)
        const int granule = 1280;
        int dyn = (lds_cu / b) / granule * granule - static_cast<int>(fa.sharedSizeBytes);
        dyn = std::min(dyn, cap * level_bytes + std::max(n_nodes, 0) * 64) / 64 * 64;
        int got = 0;
        while (dyn >= useful * level_bytes)
        {
            raise_lds_limit(kernel, dyn);
            HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&got, kernel, BLOCK, static_cast<size_t>(dyn)));
            if (got >= b)
                break;
            got = 0;
            dyn -= granule;
        }
        if (dyn >= useful * level_bytes && got >= b)
        {
            avail = dyn, r.per_cu = b;
            break;
        }
    }
    if (avail <= 0)
        avail = std::min(cap, std::max(1, (lds_cu - static_cast<int>(fa.sharedSizeBytes)) / level_bytes)) * level_bytes; // one workgroup per CU
    int levels = std::min(cap, avail / level_bytes);
    int nodes = 0;
    if (lds_nodes_opt != 0)
    {
        // with nodes in LDS the stacks keep 16 levels unless the option says otherwise
        const int keep = std::min(levels, lds_stack_cap > 0 ? cap : 16);
        nodes = std::min((avail - keep * level_bytes) / 64, n_nodes);
        if (lds_nodes_opt > 0)
            nodes = std::min(nodes, lds_nodes_opt);
        if (nodes >= 16)
            levels = keep;
        else
            nodes = 0;
    }
    r.levels = levels, r.nodes = nodes;
    r.dyn_bytes = static_cast<size_t>(levels) * level_bytes + static_cast<size_t>(nodes) * 64;
    raise_lds_limit(kernel, static_cast<int>(r.dyn_bytes));
    return r;
}

template <class R> static auto wf_anyhit_kernel()
{
    if constexpr (sizeof(R) == 8)
        return k_wf_anyhit_f64<0>;
    else
        return k_wf_anyhit_f32<0>;
}
template <class R> static auto wf_traverse_kernel()
{
    if constexpr (sizeof(R) == 8)
        return k_wf_traverse_f64<0>;
    else
        return k_wf_traverse_f32<0>;
}

// closest hits of n rays through the wavefront tracer's traversal kernel (timing aid)
template <class R> void Context::raycast_wavefront_impl(Impl<R> &im, const double *ro, const double *rd, int64_t n, int32_t *ids, double *pts)
{
    hipStream_t s = static_cast<hipStream_t>(stream_);
    auto S = make_dscene<R>(scene_, *geo_, im.tri_pre.p, im.mats.p, im.lights.p, im.tri_nrm.p, im.bvh.p, brute_);
    auto kernel = wf_traverse_kernel<R>();
    const BvhLdsPlan lp = plan_bvh_lds(reinterpret_cast<const void *>(kernel), trace_blocks_per_cu, geo_->stack_need, lds_stack_levels, S.n_nodes, lds_nodes);
    const int grid = lp.per_cu * n_cus_;
    DeviceBuffer d_o, d_d, d_ids, d_pts, d_q, d_spill, d_stats;
    d_o.alloc(n * 24), d_d.alloc(n * 24), d_ids.alloc(n * 4);
    if (pts)
        d_pts.alloc(n * 24);
    d_q.alloc(wf_queue_bytes<R>(n));
    const int spill_levels = std::max(geo_->stack_need - lp.levels, 0);
    d_spill.alloc(static_cast<size_t>(spill_levels) * grid * BLOCK * 4);
    d_stats.alloc(static_cast<size_t>(grid) * (BLOCK / 64) * WAVE_STATS * 8);
    HIP_CHECK(hipMemsetAsync(d_stats.p, 0, d_stats.bytes, s));
    HIP_CHECK(hipMemcpyAsync(d_o.p, ro, n * 24, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d_d.p, rd, n * 24, hipMemcpyHostToDevice, s));
    WfQueue<R> Q = wf_queue_at<R>(static_cast<char *>(d_q.p), n);
    S.stack_levels = lp.levels, S.lds_nodes = lp.nodes;
    S.spill = static_cast<uint32_t *>(d_spill.p), S.spill_stride = grid * BLOCK;
    if (std::getenv("PTX_DEBUG"))
        std::fprintf(stderr, "[ptx] wavefront traversal: %d workgroups per CU, %zu bytes of dynamic LDS (%d of %d stack levels, %d nodes)\n", lp.per_cu,
                     lp.dyn_bytes, lp.levels, geo_->stack_need, lp.nodes);
    hipLaunchKernelGGL((k_wf_rays_in<R>), dim3(blocks_for(n)), dim3(BLOCK), 0, s, static_cast<const double *>(d_o.p), static_cast<const double *>(d_d.p), n, Q);
    hipLaunchKernelGGL(kernel, dim3(grid), dim3(BLOCK), lp.dyn_bytes, s, S, Q, wf_refill, static_cast<unsigned long long *>(d_stats.p));
    hipLaunchKernelGGL((k_wf_hits_out<R>), dim3(blocks_for(n)), dim3(BLOCK), 0, s, S, Q, n, static_cast<int32_t *>(d_ids.p),
                       pts ? static_cast<double *>(d_pts.p) : nullptr);
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(ids, d_ids.p, n * 4, hipMemcpyDeviceToHost, s));
    if (pts)
        HIP_CHECK(hipMemcpyAsync(pts, d_pts.p, n * 24, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

void Context::raycast(const double *ro, const double *rd, int64_t n, int32_t *ids, double *pts, bool from_camera, int mode)
{
    require_scene();
    HIP_CHECK(hipSetDevice(device_));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (from_camera)
        n = static_cast<int64_t>(scene_.width) * scene_.height;
    if (n <= 0)
        return;
    if (raycast_wavefront && !brute_ && !from_camera)
    {
        if (mode == PTX_MODE_F64_)
            raycast_wavefront_impl<double>(*f64_, ro, rd, n, ids, pts);
        else
            raycast_wavefront_impl<float>(*f32_, ro, rd, n, ids, pts);
        return;
    }
    DeviceBuffer d_o, d_d, d_ids, d_pts;
    if (!from_camera)
    {
        d_o.alloc(n * 24), d_d.alloc(n * 24);
        HIP_CHECK(hipMemcpyAsync(d_o.p, ro, n * 24, hipMemcpyHostToDevice, s));
        HIP_CHECK(hipMemcpyAsync(d_d.p, rd, n * 24, hipMemcpyHostToDevice, s));
    }
    d_ids.alloc(n * 4);
    if (pts)
        d_pts.alloc(n * 24);
    DCamera cam = make_camera(scene_);
    const dim3 grid(blocks_for(n)), block(BLOCK);
    const double *po = static_cast<const double *>(d_o.p), *pd = static_cast<const double *>(d_d.p);
    int32_t *pi = static_cast<int32_t *>(d_ids.p);
    double *pp = pts ? static_cast<double *>(d_pts.p) : nullptr;
    if (mode == PTX_MODE_F64_)
    {
        auto S = make_dscene<double>(scene_, *geo_, f64_->tri_pre.p, f64_->mats.p, f64_->lights.p, f64_->tri_nrm.p, f64_->bvh.p, brute_);
        if (brute_)
            hipLaunchKernelGGL((k_raycast<double, true>), grid, block, 0, s, S, cam, int(from_camera), po, pd, n, pi, pp);
        else
        {
            allow_dynamic_lds(k_raycast<double, false>, stack_bytes());
            hipLaunchKernelGGL((k_raycast<double, false>), grid, block, stack_bytes(), s, S, cam, int(from_camera), po, pd, n, pi, pp);
        }
    }
    else
    {
        auto S = make_dscene<float>(scene_, *geo_, f32_->tri_pre.p, f32_->mats.p, f32_->lights.p, f32_->tri_nrm.p, f32_->bvh.p, brute_);
        if (brute_)
            hipLaunchKernelGGL((k_raycast<float, true>), grid, block, 0, s, S, cam, int(from_camera), po, pd, n, pi, pp);
        else
        {
            allow_dynamic_lds(k_raycast<float, false>, stack_bytes());
            hipLaunchKernelGGL((k_raycast<float, false>), grid, block, stack_bytes(), s, S, cam, int(from_camera), po, pd, n, pi, pp);
        }
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(ids, d_ids.p, n * 4, hipMemcpyDeviceToHost, s));
    if (pts)
        HIP_CHECK(hipMemcpyAsync(pts, d_pts.p, n * 24, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

void Context::vistest(const double *p1, const double *p2, int64_t n, int32_t *vis)
{
    require_scene();
    HIP_CHECK(hipSetDevice(device_));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    if (n <= 0)
        return;
    DeviceBuffer d1, d2, dv;
    d1.alloc(n * 24), d2.alloc(n * 24), dv.alloc(n * 4);
    HIP_CHECK(hipMemcpyAsync(d1.p, p1, n * 24, hipMemcpyHostToDevice, s));
    HIP_CHECK(hipMemcpyAsync(d2.p, p2, n * 24, hipMemcpyHostToDevice, s));
    auto S = make_dscene<double>(scene_, *geo_, f64_->tri_pre.p, f64_->mats.p, f64_->lights.p, f64_->tri_nrm.p, f64_->bvh.p, brute_);
    const dim3 grid(blocks_for(n)), block(BLOCK);
    if (brute_)
        hipLaunchKernelGGL((k_vistest<double, true>), grid, block, 0, s, S, static_cast<const double *>(d1.p),
                           static_cast<const double *>(d2.p), n, static_cast<int32_t *>(dv.p));
    else
    {
        allow_dynamic_lds(k_vistest<double, false>, stack_bytes());
        hipLaunchKernelGGL((k_vistest<double, false>), grid, block, stack_bytes(), s, S, static_cast<const double *>(d1.p),
                           static_cast<const double *>(d2.p), n, static_cast<int32_t *>(dv.p));
    }
    HIP_CHECK(hipGetLastError());
    HIP_CHECK(hipMemcpyAsync(vis, dv.p, n * 4, hipMemcpyDeviceToHost, s));
    HIP_CHECK(hipStreamSynchronize(s));
}

// ------------------------------------------------------------------------------------------------
// render loop
// ------------------------------------------------------------------------------------------------
// the instantiation of the tracer for a scene, with its launch geometry
template <class R> struct TracePlan
{
    void (*kernel)(DScene<R>, DCamera, DBatch, DState<R>, int) = nullptr;
    std::string name; // which instantiation, as profilers print it (k_trace_f64<true, true, 1>, ...)
    int blocks_per_cu = 1;
    size_t dyn_lds = 0; // bytes of traversal stacks per workgroup (BVH kernels)
    int stack_levels = 0; // levels of them
    int lds_nodes = 0;    // node records kept in LDS behind the stack levels
    bool rings = false;   // the ring tracer (needs DState::rings)
    bool ring_tests = false; // ... compiled with visibility tests (a longer ray ring)
    bool wavefront = false;  // the wavefront tracer: `traverse` instead of `kernel`, shading in kernels of its own
    bool wf_fast = false;    // ... running the physically based estimator (visibility tests through `anyhit`)
    bool wf_sq = false;      // ... running the reference's estimator on a scene with square lights (their tests likewise)
    void (*traverse)(DScene<R>, WfQueue<R>, int, unsigned long long *) = nullptr;
    void (*anyhit)(DScene<R>, WfQueue<R>, int, unsigned long long *) = nullptr;
};

template <class R, bool BRUTE>
static TracePlan<R> plan_trace(const DScene<R> &S, int blocks_per_cu, bool shadow_queue, int stack_need, int lds_stack_cap, bool rings = false,
                               bool fast = false, int lds_nodes_opt = -1, bool wavefront = false)
{
    TracePlan<R> p;
    const bool tab = S.n_mats <= LDS_MATS && S.n_lights <= LDS_LIGHTS;
    // how visibility tests are compiled (see trace_body): none / parked in the LDS queue / in place.
    // The reference's estimator tests square lights only, the physically based one every light.
    const int n_tested = fast ? S.n_lights : S.n_square;
    const int lights = n_tested == 0 ? 0 : ((BRUTE && S.n_lights == 1 && shadow_queue) ? 1 : 2);
    const bool rings_wanted = rings; // the option, before the limits of the ring tracer itself
    if (fast && S.n_lights > RING_MAX_TESTS)
        rings = false; // the state word of a ring entry counts up to 12 visibility tests per bounce
    auto pick = [&](auto tab_c) {
        constexpr bool TAB = decltype(tab_c)::value;
        if (fast)
        {
            if constexpr (sizeof(R) == 8)
            {
                if constexpr (BRUTE)
                    p.kernel = lights == 1 ? k_trace_fast_f64<true, TAB, 1> : k_trace_fast_f64<true, TAB, 2>;
                else
                    p.kernel = k_trace_fast_f64<false, TAB, 2>;
            }
            else
            {
                if constexpr (BRUTE)
                    p.kernel = lights == 1 ? k_trace_fast_f32<true, TAB, 1> : k_trace_fast_f32<true, TAB, 2>;
                else
                    p.kernel = k_trace_fast_f32<false, TAB, 2>;
            }
        }
        else if constexpr (sizeof(R) == 8)
        {
            if constexpr (BRUTE)
                p.kernel = lights == 1 ? k_trace_f64<true, TAB, 1> : k_trace_f64<true, TAB, 2>;
            else
                p.kernel = lights == 0 ? k_trace_f64<false, TAB, 0> : k_trace_f64<false, TAB, 2>;
        }
        else
        {
            if constexpr (BRUTE)
                p.kernel = lights == 1 ? k_trace_f32<true, TAB, 1> : k_trace_f32<true, TAB, 2>;
            else
                p.kernel = lights == 0 ? k_trace_f32<false, TAB, 0> : k_trace_f32<false, TAB, 2>;
        }
    };
    if (tab)
        pick(std::true_type());
    else
        pick(std::false_type());
    {
        char nm[96];
        std::snprintf(nm, sizeof(nm), "k_trace_%s%s<%s, %s, %d>", fast ? "fast_" : "", sizeof(R) == 8 ? "f64" : "f32", BRUTE ? "true" : "false",
                      tab ? "true" : "false", (fast && lights != 1) ? 2 : lights);
        p.name = nm;
    }
    if (!BRUTE && fast && rings)
    {
        p.rings = true, p.ring_tests = true;
        if constexpr (sizeof(R) == 8)
            p.kernel = tab ? k_trace_rings_fast_f64<true> : k_trace_rings_fast_f64<false>;
        else
            p.kernel = tab ? k_trace_rings_fast_f32<true> : k_trace_rings_fast_f32<false>;
        char nm[96];
        std::snprintf(nm, sizeof(nm), "k_trace_rings_fast_%s<%s>", sizeof(R) == 8 ? "f64" : "f32", tab ? "true" : "false");
        p.name = nm;
    }
    else if (!BRUTE && S.n_square <= RING_MAX_TESTS && (S.n_square == 0 || S.n_lights <= 32) && rings)
    {
        // the tracer that keeps more paths in flight than lanes (its state word counts up to 12
        // visibility tests per bounce, and a shading pass names the lights to test in a 32-bit set)
        p.rings = true;
        const bool sq = S.n_square >= 1;
        p.ring_tests = sq;
        if constexpr (sizeof(R) == 8)
            p.kernel = tab ? (sq ? k_trace_rings_f64<true, true> : k_trace_rings_f64<true, false>)
                           : (sq ? k_trace_rings_f64<false, true> : k_trace_rings_f64<false, false>);
        else
            p.kernel = tab ? (sq ? k_trace_rings_f32<true, true> : k_trace_rings_f32<true, false>)
                           : (sq ? k_trace_rings_f32<false, true> : k_trace_rings_f32<false, false>);
        char nm[96];
        std::snprintf(nm, sizeof(nm), "k_trace_rings_%s<%s, %s>", sizeof(R) == 8 ? "f64" : "f32", tab ? "true" : "false", sq ? "true" : "false");
        p.name = nm;
    }
    // BVH scenes whose lights need no visibility test (the reference's estimator without square
    // lights): traversal and shading in separate kernels, segments through two ray queues
    // -- and the physically based estimator with up to 32 lights, whose visibility tests are a third
    // queue answered by an any-hit launch per bounce (a test slot per light and segment: the batch
    // shrinks with the number of lights; with 20 lights still 2.3 times the lane tracer's rate)
    // -- and the reference's estimator on scenes with square lights and at most 12 lights in all, the same
    // way (every light takes a slot there, tested or not: with 20 lights the ring tracer is 20 % faster)
    const bool wf_ok = fast ? S.n_lights <= WF_MAX_LIGHTS : (S.n_square == 0 || S.n_lights <= RING_MAX_TESTS);
    if (!BRUTE && wavefront && rings_wanted && wf_ok)
    {
        p.wavefront = true, p.rings = false, p.ring_tests = false;
        p.wf_fast = fast;
        p.wf_sq = !fast && S.n_square > 0;
        p.traverse = wf_traverse_kernel<R>();
        p.anyhit = wf_anyhit_kernel<R>();
        p.name = std::string(sizeof(R) == 8 ? "k_wf_traverse_f64<0>" : "k_wf_traverse_f32<0>") + (fast ? " + k_wf_shade_fast" : p.wf_sq ? " + k_wf_anyhit" : "");
    }
    // BVH kernels: occupancy and the split of the dynamic LDS between stack levels and node records
    int per_cu = blocks_per_cu;
    p.stack_levels = 0;
    if (!BRUTE)
    {
        const void *kfn = p.wavefront ? reinterpret_cast<const void *>(p.traverse) : reinterpret_cast<const void *>(p.kernel);
        const BvhLdsPlan lp = plan_bvh_lds(kfn, blocks_per_cu, stack_need, lds_stack_cap, S.n_nodes, lds_nodes_opt);
        per_cu = lp.per_cu;
        p.stack_levels = lp.levels;
        p.lds_nodes = lp.nodes;
        p.dyn_lds = lp.dyn_bytes;
        if (p.wf_fast || p.wf_sq) // same registers, same LDS split
            {
                std::lock_guard<std::mutex> lock(g_lds_limit_mutex);
                raise_lds_limit(reinterpret_cast<const void *>(p.anyhit), static_cast<int>(p.dyn_lds));
            }
    }
    else if (per_cu <= 0)
        HIP_CHECK(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, p.kernel, BLOCK, 0));
    if (std::getenv("PTX_DEBUG"))
        std::fprintf(stderr, "[ptx] trace kernel: %d workgroups per CU, %zu bytes of dynamic LDS (%d of %d stack levels, %d of %d nodes)\n", per_cu,
                     p.dyn_lds, p.stack_levels, stack_need, p.lds_nodes, S.n_nodes);
    p.blocks_per_cu = std::max(per_cu, 1);
    return p;
}

// grid_blocks: the persistent grid the state region was laid out for; a batch with fewer paths
// than that grid has lanes starts fewer workgroups
template <class R>
static void launch_trace(const TracePlan<R> &plan, const DScene<R> &S, const DCamera &cam, const DBatch &B, DState<R> &st, hipStream_t s,
                         int chunk, int grid_blocks)
{
    const int64_t blocks = std::max<int64_t>(1, std::min<int64_t>(grid_blocks, (B.n_paths + BLOCK - 1) / BLOCK));
    st.n_waves = static_cast<int>(blocks) * (BLOCK / 64);
    hipLaunchKernelGGL(plan.kernel, dim3(static_cast<unsigned>(blocks)), dim3(BLOCK), plan.dyn_lds, s, S, cam, B, st, chunk);
    HIP_CHECK(hipGetLastError());
}

// One batch through the wavefront tracer: camera rays, then per bounce one persistent traversal
// launch and one shading launch (which reads the count the launch before it produced: nothing
// comes back to the host), then the fold of the records.  grid_blocks: workgroups of the
// traversal kernel (the spill columns and the statistics rows were laid out for them).
template <class R>
static void launch_wavefront(const TracePlan<R> &plan, const DScene<R> &S, const DCamera &cam, const DBatch &B, Context::Impl<R> &im, hipStream_t s,
                             int grid_blocks, int refill, unsigned long long *pinned_word, bool may_wait)
{
    DState<R> &st = im.st;
    st.n_waves = grid_blocks * (BLOCK / 64);
    const unsigned per_path = blocks_for(B.n_paths);
    HIP_CHECK(hipMemsetAsync(st.wave_stats, 0, static_cast<size_t>(st.n_waves) * WAVE_STATS * sizeof(unsigned long long), s));
    hipLaunchKernelGGL((k_wf_camera<R>), dim3(blocks_for(B.n_pix)), dim3(BLOCK), 0, s, cam, B, im.wfq[0]);
    for (int b = 0; b < B.depth; b++)
    {
        const WfQueue<R> &in = im.wfq[b & 1], &out = im.wfq[(b + 1) & 1];
        // Deep scenes: most paths end long before Depth.  From bounce 8 on, every eighth bounce reads
        // the count of the queue it is about to traverse back (4 bytes into pinned memory, one wait)
        // and stops when no ray is left; scenes of ordinary depth never get here.
        if (may_wait && b >= 8 && (b & 7) == 0)
        {
            *pinned_word = 1ull;
            HIP_CHECK(hipMemcpyAsync(pinned_word, in.count, sizeof(uint32_t), hipMemcpyDeviceToHost, s));
            HIP_CHECK(hipStreamSynchronize(s));
            if (static_cast<uint32_t>(*pinned_word) == 0u)
                break;
        }
        if (b == 0)
            hipLaunchKernelGGL((k_wf_counts<R>), dim3(1), dim3(1), 0, s, in, static_cast<uint32_t>(B.n_pix), out);
        else
            HIP_CHECK(hipMemsetAsync(out.count, 0, 8, s)); // count and pool of the queue this bounce fills
        hipLaunchKernelGGL(plan.traverse, dim3(static_cast<unsigned>(grid_blocks)), dim3(BLOCK), plan.dyn_lds, s, S, in, refill, st.wave_stats);
        const dim3 shade_grid(static_cast<unsigned>((B.n_paths + WF_SHADE_BLOCK - 1) / WF_SHADE_BLOCK));
        if (plan.wf_fast)
        {
            // shading sends the visibility tests of the bounce to the test queue, an any-hit launch answers
            // them, the gather fills in the light-sampling half of the bounce's records
            const WfQueue<R> &tq = im.wft;
            HIP_CHECK(hipMemsetAsync(tq.count, 0, WF_QUEUE_HEADER_WORDS * sizeof(uint32_t), s)); // pool and the lengths of the per-light lists
            if (b == 0)
                hipLaunchKernelGGL((k_wf_shade_fast<R, true>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, tq, im.wfs, b);
            else
                hipLaunchKernelGGL((k_wf_shade_fast<R, false>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, tq, im.wfs, b);
            if (S.n_lights > 0)
            {
                hipLaunchKernelGGL(plan.anyhit, dim3(static_cast<unsigned>(grid_blocks)), dim3(BLOCK), plan.dyn_lds, s, S, tq, refill, st.wave_stats);
                if (b == 0)
                    hipLaunchKernelGGL((k_wf_gather<R, true>), dim3(per_path), dim3(BLOCK), 0, s, B, in, tq, im.wfs, b, S.n_lights);
                else
                    hipLaunchKernelGGL((k_wf_gather<R, false>), dim3(per_path), dim3(BLOCK), 0, s, B, in, tq, im.wfs, b, S.n_lights);
            }
        }
        else if (plan.wf_sq)
        {
            // the reference's estimator with square lights: the same three steps
            const WfQueue<R> &tq = im.wft;
            HIP_CHECK(hipMemsetAsync(tq.count, 0, WF_QUEUE_HEADER_WORDS * sizeof(uint32_t), s)); // pool and the lengths of the per-light lists
            if (b == 0)
                hipLaunchKernelGGL((k_wf_shade<R, true, true>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, tq, im.wfs, b);
            else
                hipLaunchKernelGGL((k_wf_shade<R, false, true>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, tq, im.wfs, b);
            hipLaunchKernelGGL(plan.anyhit, dim3(static_cast<unsigned>(grid_blocks)), dim3(BLOCK), plan.dyn_lds, s, S, tq, refill, st.wave_stats);
            if (b == 0)
                hipLaunchKernelGGL((k_wf_gather<R, true>), dim3(per_path), dim3(BLOCK), 0, s, B, in, tq, im.wfs, b, S.n_lights);
            else
                hipLaunchKernelGGL((k_wf_gather<R, false>), dim3(per_path), dim3(BLOCK), 0, s, B, in, tq, im.wfs, b, S.n_lights);
        }
        else if (b == 0)
            hipLaunchKernelGGL((k_wf_shade<R, true, false>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, im.wfq[0], im.wfs, b);
        else
            hipLaunchKernelGGL((k_wf_shade<R, false, false>), shade_grid, dim3(WF_SHADE_BLOCK), 0, s, S, B, in, out, im.wfq[0], im.wfs, b);
    }
    if (plan.wf_fast)
        hipLaunchKernelGGL((k_wf_resolve<R, true>), dim3(per_path), dim3(BLOCK), 0, s, S, B, im.wfs);
    else
        hipLaunchKernelGGL((k_wf_resolve<R, false>), dim3(per_path), dim3(BLOCK), 0, s, S, B, im.wfs);
    HIP_CHECK(hipGetLastError());
}

// accumulate + statistics of one traced batch
template <class R>
static void launch_finish(const DScene<R> &S, const DBatch &B, const DState<R> &st, double *sum, hipStream_t s, RenderTimers *tm)
{
    const dim3 block(BLOCK);
    if (tm)
        tm->mark(s, RenderTimers::ACCUMULATE);
    hipLaunchKernelGGL((k_accumulate<R>), dim3(blocks_for(B.n_pix)), block, 0, s, B, st, sum);
    if (tm)
        tm->mark(s, RenderTimers::OTHER);
    hipLaunchKernelGGL((k_reduce_stats<R>), dim3(1), block, 0, s, B, st);
    if (tm)
        tm->mark(s, RenderTimers::END);
    HIP_CHECK(hipGetLastError());
}

void RenderTimers::mark(hipStream_t s, int cls)
{
    hipEvent_t e;
    if (used < events.size())
        e = events[used];
    else
    {
        HIP_CHECK(hipEventCreate(&e));
        events.push_back(e);
    }
    used++;
    HIP_CHECK(hipEventRecord(e, s));
    classes.push_back(cls);
}

void RenderTimers::collect(double ms[4])
{
    for (int i = 0; i < 4; i++)
        ms[i] = 0;
    for (size_t i = 0; i + 1 < used; i++)
    {
        if (classes[i] == END)
            continue;
        float t = 0;
        if (hipEventElapsedTime(&t, events[i], events[i + 1]) == hipSuccess)
            ms[classes[i]] += t;
    }
    used = 0;
    classes.clear();
}

RenderTimers::~RenderTimers()
{
    for (auto e : events)
        (void)hipEventDestroy(e);
}

template <class R> void Context::render_impl(Impl<R> &im, const RenderOpts &o, double *d_sum, RenderStats &stats, bool static_state)
{
    hipStream_t s = static_cast<hipStream_t>(stream_);
    const Scene &sc = scene_;
    const int depth = o.depth > 0 ? o.depth : sc.depth;
    const int spp = o.spp > 0 ? o.spp : sc.spp;
    const int spp_total = o.spp_total > 0 ? o.spp_total : spp;
    if (depth > 65535)
        throw std::runtime_error("arg: depth above 65535 is not supported");
    if (o.estimator != PTX_ESTIMATOR_REFERENCE_ && o.estimator != PTX_ESTIMATOR_FAST_)
        throw std::runtime_error("arg: unknown estimator");

    // rows / columns inside DrawRegion (inclusive ends), clipped to the image
    int r0 = std::max(sc.region[0], 0), r1 = std::min(sc.region[2], sc.height - 1) + 1;
    int c0 = std::max(sc.region[1], 0), c1 = std::min(sc.region[3], sc.width - 1) + 1;
    if (o.row_end > o.row_begin)
    {
        r0 = std::max(r0, o.row_begin);
        r1 = std::min(r1, o.row_end);
    }
    if (r1 <= r0 || c1 <= c0)
        return;
    // interleaved bands: local row lr <-> image row
    const int stride = std::max(o.row_stride, 1), band = std::max(o.band, 1), phase = o.row_phase;
    int n_local_rows;
    if (stride <= 1)
        n_local_rows = r1 - r0;
    else
    {
        n_local_rows = 0;
        for (int r = 0; r < r1 - r0; r++)
            n_local_rows += ((r / band) % stride) == phase;
    }
    if (n_local_rows <= 0)
        return;
    const int n_cols = c1 - c0;

    auto S = make_dscene<R>(sc, *geo_, im.tri_pre.p, im.mats.p, im.lights.p, im.tri_nrm.p, im.bvh.p, brute_);
    DCamera cam = make_camera(sc);
    RenderTimers *tm = (o.profile && !static_state && !o.enqueue_only) ? &timers_ : nullptr;
    // The grid of the tracer is just large enough to fill the device: its waves are persistent.
    const bool fast = o.estimator == PTX_ESTIMATOR_FAST_;
    TracePlan<R> plan = brute_ ? plan_trace<R, true>(S, trace_blocks_per_cu, shadow_queue, 0, 0, false, fast)
                              : plan_trace<R, false>(S, trace_blocks_per_cu, shadow_queue, geo_->stack_need, lds_stack_levels, bvh_rings, fast, lds_nodes, bvh_wavefront);
    if (plan.wavefront)
    {
        // The wavefront tracer keeps depth records per path of a batch: at a depth where even its
        // smallest batch does not fit the memory there is (or the static region), the scene goes to the
        // ring / lane tracers, whose record slots belong to the grid and shrink with it.
        size_t room = STATIC_STATE_BYTES;
        if (!static_state)
        {
            size_t free_b = 0, total_b = 0;
            HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
            room = static_cast<size_t>(0.9 * static_cast<double>(free_b + im.state_mem.bytes));
        }
        StateNeeds least;
        least.paths = std::max<int64_t>(c1 - c0, int64_t(1) << 18), least.depth = depth;
        least.waves = plan.blocks_per_cu * n_cus_ * (BLOCK / 64), least.slots = 0, least.wavefront = true;
        least.wf_tests = (plan.wf_fast || plan.wf_sq) ? std::max(S.n_lights, 1) : 0;
        least.spill_levels = std::max(geo_->stack_need - plan.stack_levels, 0);
        uint32_t *unused = nullptr;
        if (im.layout_state(nullptr, least, &unused) > room)
            plan = plan_trace<R, false>(S, trace_blocks_per_cu, shadow_queue, geo_->stack_need, lds_stack_levels, bvh_rings, fast, lds_nodes, false);
    }
    last_kernel_ = plan.name;

    // ---- size the batch and the grid so that the state fits the memory there is --------------
    // The record slots belong to the waves of the grid (a path holds one from its start to its
    // resolve pass), so the state does not grow with the batch except for 3 scalars per path.
    // (the wavefront tracer runs one traversal launch per bounce of the whole batch and gains from
    // larger ones -- room: +3 % at 2^26 paths, +8 % at 2^27 -- at 400 to 900 bytes of state per path: twice the default)
    int64_t batch = o.batch_paths > 0 ? o.batch_paths : default_batch_paths * ((plan.wavefront && !plan.wf_fast && !plan.wf_sq) ? 2 : 1); // (its test queue makes the other estimator's state 2 to 5 times larger: no doubling there)
    batch = std::max<int64_t>(batch, n_cols);
    int grid_blocks = static_cast<int>(std::min<int64_t>(static_cast<int64_t>(plan.blocks_per_cu) * n_cus_, MAX_TRACE_WAVES / (BLOCK / 64)));
    size_t budget = STATIC_STATE_BYTES;
    if (!static_state)
    {
        size_t free_b = 0, total_b = 0;
        HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
        budget = static_cast<size_t>(0.9 * static_cast<double>(free_b + im.state_mem.bytes));
    }
    StateNeeds need;
    int rows_per_chunk = 1, samples_per_batch = 1;
    int64_t cap = 0;
    uint32_t *spill = nullptr;
    for (;;)
    {
        rows_per_chunk = static_cast<int>(std::min<int64_t>(n_local_rows, std::max<int64_t>(1, batch / n_cols)));
        const int64_t pix_per_chunk = static_cast<int64_t>(rows_per_chunk) * n_cols;
        samples_per_batch = static_cast<int>(std::min<int64_t>(spp, std::max<int64_t>(1, batch / pix_per_chunk)));
        cap = pix_per_chunk * samples_per_batch;
        if (cap >= (int64_t(1) << 31))
            throw std::runtime_error("arg: batch_paths must stay below 2^31");
        need.paths = cap, need.depth = depth;
        need.waves = grid_blocks * (BLOCK / 64);
        need.slots = plan.rings ? RING_SLOTS : LANE_SLOTS;
        need.wavefront = plan.wavefront;
        need.wf_tests = (plan.wf_fast || plan.wf_sq) ? std::max(S.n_lights, 1) : 0;
        need.ring_bytes = plan.rings ? ring_bytes<R>(fast, plan.ring_tests) : 0;
        need.spill_levels = brute_ ? 0 : std::max(geo_->stack_need - plan.stack_levels, 0);
        need.ring_batch = ring_batch;
        // (the wavefront tracer's test queue is indexed with 32 bits: batch x lights must stay below 2^31)
        const bool queue_ok = need.wf_tests == 0 || cap * need.wf_tests < (int64_t(1) << 31);
        if (queue_ok && im.layout_state(nullptr, need, &spill) <= budget)
            break;
        // too large: first smaller batches (they cost a few per cent at most), then a smaller grid
        if (batch > std::max<int64_t>(n_cols, int64_t(1) << 18))
            batch = std::max<int64_t>(std::max<int64_t>(n_cols, int64_t(1) << 18), batch / 2);
        else if (grid_blocks > 8)
            grid_blocks = grid_blocks * 3 / 4;
        else
            throw std::runtime_error(static_state
                                         ? "arg: this depth and image width do not fit the static state region of ptx_render_static (256 MiB); use ptx_render"
                                         : "device: not enough device memory for the path state at this depth");
    }
    char *base;
    if (static_state)
        base = reinterpret_cast<char *>(static_state_); // the device address of the region, never the symbol's name
    else
    {
        im.reserve_state(im.layout_state(nullptr, need, &spill));
        base = static_cast<char *>(im.state_mem.p);
    }
    im.layout_state(base, need, &spill);
    if (!brute_)
    {
        // stack levels the LDS does not hold: one column per lane of the launch in global memory
        S.stack_levels = plan.stack_levels;
        S.lds_nodes = plan.lds_nodes;
        S.spill = spill;
        S.spill_stride = need.waves * 64;
    }
    const int chunk = std::max(64, chunk_paths);
    HIP_CHECK(hipMemsetAsync(im.st.stats, 0, sizeof(unsigned long long) * 8, s));
    HIP_CHECK(hipMemsetAsync(im.st.pool, 0, sizeof(unsigned int), s));

    if (!o.enqueue_only)
        HIP_CHECK(hipEventRecord(ev_begin_, s));
    uint64_t passes = 0;
    for (int lr0 = 0; lr0 < n_local_rows; lr0 += rows_per_chunk)
    {
        const int nr = std::min(rows_per_chunk, n_local_rows - lr0);
        for (int s0 = 0; s0 < spp; s0 += samples_per_batch)
        {
            DBatch B;
            B.row0 = lr0, B.n_rows = nr;
            B.col0 = c0, B.n_cols = n_cols;
            B.band = band, B.row_stride = stride, B.row_phase = phase, B.region_row0 = r0;
            B.image_width = sc.width;
            B.n_samples = std::min(samples_per_batch, spp - s0);
            B.sample0 = o.sample_offset + s0;
            B.spp_total = spp_total, B.depth = depth;
            B.seed = o.seed;
            B.n_pix = static_cast<int64_t>(nr) * n_cols;
            B.n_paths = B.n_pix * B.n_samples;
            if (tm)
                tm->mark(s, RenderTimers::TRACE);
            if (plan.wavefront)
                launch_wavefront<R>(plan, S, cam, B, im, s, grid_blocks, wf_refill, host_stats_ + 7, !o.enqueue_only);
            else
                launch_trace<R>(plan, S, cam, B, im.st, s, chunk, grid_blocks);
            launch_finish<R>(S, B, im.st, d_sum, s, tm);
            passes++;
        }
    }
    if (o.enqueue_only)
    {
        // nothing is waited for and nothing read back: the caller owns the stream (it may be capturing it)
        stats.passes = passes;
        stats.state_bytes = im.layout_state(nullptr, need, &spill);
        return;
    }
    HIP_CHECK(hipEventRecord(ev_end_, s));
    HIP_CHECK(hipMemcpyAsync(host_stats_, im.st.stats, sizeof(unsigned long long) * 8, hipMemcpyDeviceToHost, s)); // pinned: a plain DMA
    HIP_CHECK(hipStreamSynchronize(s));
    float ms = 0;
    HIP_CHECK(hipEventElapsedTime(&ms, ev_begin_, ev_end_));
    const unsigned long long *h_stats = host_stats_;
    stats.segments = h_stats[0];
    stats.shadow_rays = h_stats[1];
    stats.paths = h_stats[2];
    stats.wave_iterations = h_stats[3];
    stats.records = h_stats[4];
    stats.bvh_nodes = h_stats[5];
    stats.bvh_tris = h_stats[6];
    stats.passes = passes;
    stats.ms_total = ms;
    stats.state_bytes = im.layout_state(nullptr, need, &spill);
    if (tm)
        tm->collect(stats.ms_class);
}

void Context::render(const RenderOpts &o, double *sum, bool out_is_device, RenderStats &stats, bool static_state)
{
    require_scene();
    HIP_CHECK(hipSetDevice(device_));
    hipStream_t s = static_cast<hipStream_t>(stream_);
    stats = RenderStats();
    const int64_t n = static_cast<int64_t>(scene_.width) * scene_.height;
    if (static_state && !out_is_device)
        throw std::runtime_error("arg: ptx_render_static writes to device memory only");
    if (o.enqueue_only && !out_is_device)
        throw std::runtime_error("arg: enqueue_only needs a device output buffer (a copy to the host would have to be waited for)");
    DeviceBuffer tmp;
    double *d_sum = sum;
    if (!out_is_device)
    {
        tmp.alloc(n * 24);
        d_sum = static_cast<double *>(tmp.p);
        if (o.accumulate)
            HIP_CHECK(hipMemcpyAsync(d_sum, sum, n * 24, hipMemcpyHostToDevice, s));
    }
    if (!o.accumulate)
        HIP_CHECK(hipMemsetAsync(d_sum, 0, n * 24, s));
    if (scene_.enable_path_tracing)
    {
        if (o.mode == PTX_MODE_F64_)
            render_impl(*f64_, o, d_sum, stats, static_state);
        else if (o.mode == PTX_MODE_F32_)
            render_impl(*f32_, o, d_sum, stats, static_state);
        else
            throw std::runtime_error("arg: unknown render mode");
    }
    if (!out_is_device)
    {
        HIP_CHECK(hipMemcpyAsync(sum, d_sum, n * 24, hipMemcpyDeviceToHost, s));
        HIP_CHECK(hipStreamSynchronize(s));
    }
}

} // namespace ptx
