// One frame over several GPUs of one node from a single process: one host thread and one context
// per device, the frame cut by sample ranges or by interleaved row bands, and ONE exchange at the
// end onto the first device.  Sample ranges: the per-pixel radiance sums of all contexts are reduced
// (ncclReduce, f64 sum, over xGMI -- or peer copies added in context order): W * H * 24 bytes per
// context.  Row bands: the supports are disjoint, so nothing is summed and only a context's own rows
// travel, packed into one message of W * H * 24 / n bytes (ncclSend / ncclRecv in one group -- or one
// peer copy each), and the first device puts them in place.  Frame buffers, packed rows, the landing
// area and the host threads belong to the contexts and are reused from frame to frame.
//
// What is being sharded is the row loop of cPathTracer::RayTracing (src/tracing/cPathTracer.cpp:
// 526-529 of the reference: OpenMP over rows inside one process); the per-pixel value is
// sum_s final_s / spp followed by one clamp (cPathTracer.cpp:692, 701-702), linear in the samples
// before the clamp, so both cuts are exact up to the order of the sums, and random numbers are
// keyed by the global (pixel, sample), so every cut renders the same samples.
#include "../../include/ptx.h"
#include "context.hpp"

#include <dlfcn.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

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

// RCCL is looked up at the first sharded render, not at load time: a process that already holds
// an RCCL (PyTorch brings its own) keeps using that one, and a single-GPU user never loads it
struct Rccl
{
    void *handle = nullptr;
    decltype(&ncclCommInitAll) CommInitAll = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclReduce) Reduce = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    void load()
    {
        if (handle)
            return;
        for (const char *name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"})
            if ((handle = dlopen(name, RTLD_NOW | RTLD_LOCAL)))
                break;
        if (!handle)
            throw std::runtime_error("device: librccl.so not found (the multi-GPU exchange needs RCCL; exchange = peer copies does not)");
        auto sym = [&](const char *n) {
            void *p = dlsym(handle, n);
            if (!p)
                throw std::runtime_error(std::string("device: RCCL lacks ") + n);
            return p;
        };
        CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
        CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
        Reduce = reinterpret_cast<decltype(Reduce)>(sym("ncclReduce"));
        Send = reinterpret_cast<decltype(Send)>(sym("ncclSend"));
        Recv = reinterpret_cast<decltype(Recv)>(sym("ncclRecv"));
        GroupStart = reinterpret_cast<decltype(GroupStart)>(sym("ncclGroupStart"));
        GroupEnd = reinterpret_cast<decltype(GroupEnd)>(sym("ncclGroupEnd"));
        GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    }
    void check(ncclResult_t r, const char *what)
    {
        if (r != ncclSuccess)
            throw std::runtime_error(std::string("device: ") + what + ": " + GetErrorString(r));
    }
};
Rccl g_rccl;
std::mutex g_rccl_mutex;
// communicators of the device list last used (a frame loop reuses them)
std::vector<int> g_comm_devs;
std::vector<ncclComm_t> g_comms;

__global__ void __launch_bounds__(256) k_add_sums(double *__restrict__ dst, const double *__restrict__ src, int64_t n)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n)
        dst[i] += src[i];
}

// Rows of the interleaved bands of one owner: band b (rows [b * band, b * band + band)) belongs to
// owner b mod n.  Packed layout: the owner's rows in ascending order, one after the other.
// pack: frame -> packed (TO_PACKED) or packed -> frame; row_vals = 3 * width doubles per row.
template <bool TO_PACKED>
__global__ void __launch_bounds__(256) k_band_rows(double *__restrict__ frame, double *__restrict__ packed, int height, int64_t row_vals, int band,
                                                   int n_owners, int owner, int64_t packed_rows)
{
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= packed_rows * row_vals)
        return;
    const int64_t pr = i / row_vals, c = i % row_vals;
    // packed row pr: the (pr / band)-th band of this owner, row pr % band of it
    const int64_t b = (pr / band) * n_owners + owner;
    const int64_t row = b * band + pr % band;
    if (row >= height)
        return;
    if (TO_PACKED)
        packed[i] = frame[row * row_vals + c];
    else
        frame[row * row_vals + c] = packed[i];
}

// rows of the frame that owner `i` of `n` renders
int64_t rows_of_owner(int height, int band, int n, int i)
{
    int64_t rows = 0;
    for (int64_t b = i; b * band < height; b += n)
        rows += std::min<int64_t>(band, height - b * band);
    return rows;
}

struct Buffer
{
    void *p = nullptr;
    int dev = 0;
    size_t bytes = 0;
    // grows, never shrinks: a frame loop allocates on its first frame only
    void ensure(int device, size_t need)
    {
        if (p && bytes >= need && dev == device)
            return;
        release();
        dev = device;
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMalloc(&p, need));
        bytes = need;
    }
    void release()
    {
        if (p)
        {
            (void)hipSetDevice(dev);
            (void)hipFree(p);
        }
        p = nullptr, bytes = 0;
    }
    ~Buffer() { release(); }
};

} // namespace

// What a context keeps between sharded frames: its frame of sums, its packed bands, on the first
// device the landing area of the others' data, and the host thread that drives its GPU.  Made at the
// context's first sharded frame, freed with the context.
struct Context::ShardSlot
{
    Buffer frame, packed, stage;
    std::thread worker;
    std::mutex m;
    std::condition_variable cv;
    std::function<void()> job;
    bool has_job = false, done = true, stop = false;
    std::string error;

    ShardSlot()
    {
        worker = std::thread([this]() {
            std::unique_lock<std::mutex> lk(m);
            for (;;)
            {
                cv.wait(lk, [this]() { return has_job || stop; });
                if (stop)
                    return;
                std::function<void()> j = std::move(job);
                has_job = false;
                lk.unlock();
                std::string e;
                try
                {
                    j();
                }
                catch (const std::exception &ex)
                {
                    e = ex.what();
                }
                lk.lock();
                error = e;
                done = true;
                cv.notify_all();
            }
        });
    }
    void start(std::function<void()> j)
    {
        std::lock_guard<std::mutex> lk(m);
        job = std::move(j);
        has_job = true, done = false;
        error.clear();
        cv.notify_all();
    }
    std::string wait()
    {
        std::unique_lock<std::mutex> lk(m);
        cv.wait(lk, [this]() { return done; });
        return error;
    }
    ~ShardSlot()
    {
        {
            std::lock_guard<std::mutex> lk(m);
            stop = true;
            cv.notify_all();
        }
        if (worker.joinable())
            worker.join();
    }
};

void Context::free_shard_slot(ShardSlot *s) { delete s; }

Context::ShardSlot &Context::shard_slot()
{
    if (!shard_)
        shard_ = new ShardSlot();
    return *shard_;
}

void render_sharded(Context *const *ctxs, int n, const RenderOpts &base, int scheme, int exchange, double *rgb_sum, RenderStats *stats)
{
    if (n < 1)
        throw std::runtime_error("arg: at least one context");
    for (int i = 0; i < n; i++)
    {
        if (!ctxs[i] || !ctxs[i]->has_scene())
            throw std::runtime_error("state: ptx_build_accel has not been called on every context");
        if (ctxs[i]->scene().width != ctxs[0]->scene().width || ctxs[i]->scene().height != ctxs[0]->scene().height ||
            ctxs[i]->scene().spp != ctxs[0]->scene().spp)
            throw std::runtime_error("arg: the contexts hold scenes of different sizes");
        for (int j = 0; j < i; j++)
            if (ctxs[i] == ctxs[j])
                throw std::runtime_error("arg: the same context twice");
    }
    if (exchange == PTX_EXCHANGE_RCCL)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++)
                if (ctxs[i]->device() == ctxs[j]->device())
                    throw std::runtime_error("arg: an RCCL communicator takes every device once (two contexts share a GPU: use peer copies)");
    const Scene &sc = ctxs[0]->scene();
    const int64_t row_vals = static_cast<int64_t>(sc.width) * 3;
    const int64_t n_val = row_vals * sc.height;
    const size_t bytes = static_cast<size_t>(n_val) * sizeof(double);
    const int spp = base.spp > 0 ? base.spp : sc.spp; // samples of the whole frame
    const int band = base.band > 0 ? base.band : 8;
    const bool bands = scheme == PTX_SHARD_BANDS;
    std::vector<Context::ShardSlot *> slot(n);
    for (int i = 0; i < n; i++)
        slot[i] = &ctxs[i]->shard_slot();
    // bands: every context owns the rows of its bands and only those travel -- packed, one message of
    // rows_i * W * 24 bytes per context; sample ranges: whole frames are summed
    std::vector<int64_t> rows(n, 0), row_off(n, 0);
    if (bands)
        for (int i = 0; i < n; i++)
        {
            rows[i] = rows_of_owner(sc.height, band, n, i);
            row_off[i] = i ? row_off[i - 1] + rows[i - 1] : 0;
        }
    for (int i = 0; i < n; i++)
        slot[i]->start([&, i]() {
            Context::ShardSlot &me = *slot[i];
            me.frame.ensure(ctxs[i]->device(), bytes);
            RenderOpts o = base;
            o.accumulate = false;
            o.spp_total = base.spp_total > 0 ? base.spp_total : spp;
            o.row_stride = 1, o.row_phase = 0, o.band = 1; // the caller's interleave, if any, is not this call's
            if (!bands)
            {
                // contiguous sample ranges that tile [0, spp)
                const int q = spp / n, r = spp % n;
                o.spp = q + (i < r ? 1 : 0);
                o.sample_offset = base.sample_offset + i * q + (i < r ? i : r);
            }
            else
            {
                o.spp = spp;
                o.row_stride = n, o.row_phase = i, o.band = band;
            }
            hipStream_t st_i = static_cast<hipStream_t>(ctxs[i]->stream());
            RenderStats st;
            if (o.spp > 0)
                ctxs[i]->render(o, static_cast<double *>(me.frame.p), true, st);
            else
                HIP_CHECK(hipMemsetAsync(me.frame.p, 0, bytes, st_i));
            if (bands && i > 0 && rows[i] > 0)
            {
                me.packed.ensure(ctxs[i]->device(), static_cast<size_t>(rows[i] * row_vals) * sizeof(double));
                const int64_t nv = rows[i] * row_vals;
                hipLaunchKernelGGL(k_band_rows<true>, dim3(static_cast<unsigned>((nv + 255) / 256)), dim3(256), 0, st_i, static_cast<double *>(me.frame.p),
                                   static_cast<double *>(me.packed.p), sc.height, row_vals, band, n, i, rows[i]);
                HIP_CHECK(hipGetLastError());
            }
            HIP_CHECK(hipStreamSynchronize(st_i));
            if (stats)
                stats[i] = st;
        });
    std::string first_error;
    for (int i = 0; i < n; i++)
    {
        const std::string e = slot[i]->wait();
        if (first_error.empty() && !e.empty())
            first_error = e;
    }
    if (!first_error.empty())
        throw std::runtime_error(first_error);

    // ---- the one exchange: -> device of context 0 ------------------------------------------------
    hipStream_t s0 = static_cast<hipStream_t>(ctxs[0]->stream());
    Context::ShardSlot &root = *slot[0];
    auto unpack_on_root = [&]() {
        // the others' bands from the landing area into the root's frame (its own rows are there already)
        for (int i = 1; i < n; i++)
            if (rows[i] > 0)
            {
                const int64_t nv = rows[i] * row_vals;
                hipLaunchKernelGGL(k_band_rows<false>, dim3(static_cast<unsigned>((nv + 255) / 256)), dim3(256), 0, s0, static_cast<double *>(root.frame.p),
                                   static_cast<double *>(root.stage.p) + row_off[i] * row_vals, sc.height, row_vals, band, n, i, rows[i]);
                HIP_CHECK(hipGetLastError());
            }
    };
    if (exchange == PTX_EXCHANGE_RCCL) // also for a single context: a communicator of one, so that the same calls run
    {
        std::lock_guard<std::mutex> lock(g_rccl_mutex);
        g_rccl.load();
        std::vector<int> devs(n);
        for (int i = 0; i < n; i++)
            devs[i] = ctxs[i]->device();
        if (devs != g_comm_devs)
        {
            for (ncclComm_t c : g_comms)
                g_rccl.CommDestroy(c);
            g_comms.assign(n, nullptr);
            g_comm_devs.clear();
            g_rccl.check(g_rccl.CommInitAll(g_comms.data(), n, devs.data()), "ncclCommInitAll");
            g_comm_devs = devs;
        }
        if (bands && n > 1)
        {
            // gather of disjoint bands: every context sends its packed rows to rank 0, which receives
            // them side by side; one group, point to point over xGMI (seven peers, seven links)
            HIP_CHECK(hipSetDevice(devs[0]));
            root.stage.ensure(devs[0], bytes);
            g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
            for (int i = 1; i < n; i++)
            {
                if (rows[i] <= 0)
                    continue;
                const size_t count = static_cast<size_t>(rows[i] * row_vals);
                HIP_CHECK(hipSetDevice(devs[0]));
                g_rccl.check(g_rccl.Recv(static_cast<double *>(root.stage.p) + row_off[i] * row_vals, count, ncclDouble, i, g_comms[0], s0), "ncclRecv");
                HIP_CHECK(hipSetDevice(devs[i]));
                g_rccl.check(g_rccl.Send(slot[i]->packed.p, count, ncclDouble, 0, g_comms[i], static_cast<hipStream_t>(ctxs[i]->stream())), "ncclSend");
            }
            g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
            HIP_CHECK(hipSetDevice(devs[0]));
            unpack_on_root();
        }
        else
        {
            g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
            for (int i = 0; i < n; i++)
            {
                HIP_CHECK(hipSetDevice(devs[i]));
                g_rccl.check(g_rccl.Reduce(slot[i]->frame.p, slot[i]->frame.p, static_cast<size_t>(n_val), ncclDouble, ncclSum, 0, g_comms[i],
                                           static_cast<hipStream_t>(ctxs[i]->stream())),
                             "ncclReduce");
            }
            g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        }
        for (int i = 0; i < n; i++)
        {
            HIP_CHECK(hipSetDevice(devs[i]));
            HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(ctxs[i]->stream())));
        }
    }
    else if (n > 1)
    {
        HIP_CHECK(hipSetDevice(ctxs[0]->device()));
        root.stage.ensure(ctxs[0]->device(), bytes);
        if (bands)
        {
            // peer copies of the packed bands into the landing area, then one unpack each
            for (int i = 1; i < n; i++)
                if (rows[i] > 0)
                    HIP_CHECK(hipMemcpyPeerAsync(static_cast<double *>(root.stage.p) + row_off[i] * row_vals, ctxs[0]->device(), slot[i]->packed.p,
                                                 ctxs[i]->device(), static_cast<size_t>(rows[i] * row_vals) * sizeof(double), s0));
            unpack_on_root();
        }
        else
            // whole frames into the landing area one after the other, added in context order (a fixed sum order)
            for (int i = 1; i < n; i++)
            {
                HIP_CHECK(hipMemcpyPeerAsync(root.stage.p, ctxs[0]->device(), slot[i]->frame.p, ctxs[i]->device(), bytes, s0));
                hipLaunchKernelGGL(k_add_sums, dim3(static_cast<unsigned>((n_val + 255) / 256)), dim3(256), 0, s0,
                                   static_cast<double *>(root.frame.p), static_cast<const double *>(root.stage.p), n_val);
                HIP_CHECK(hipGetLastError());
            }
        HIP_CHECK(hipStreamSynchronize(s0));
    }
    HIP_CHECK(hipSetDevice(ctxs[0]->device()));
    HIP_CHECK(hipMemcpyAsync(rgb_sum, root.frame.p, bytes, hipMemcpyDeviceToHost, s0));
    HIP_CHECK(hipStreamSynchronize(s0));
}

} // namespace ptx
