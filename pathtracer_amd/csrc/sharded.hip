// One frame over several GPUs of one node from a single process: one host thread and one context
// per device, the frame cut by sample ranges or by interleaved row bands, and ONE exchange at the
// end -- the reduce of the per-pixel radiance sums onto the first device, by RCCL (ncclReduce, f64
// sum, over xGMI) or by peer copies added in device order.
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

struct Buffer
{
    void *p = nullptr;
    int dev = 0;
    void alloc(int device, size_t bytes)
    {
        dev = device;
        HIP_CHECK(hipSetDevice(device));
        HIP_CHECK(hipMalloc(&p, bytes));
    }
    ~Buffer()
    {
        if (p)
        {
            (void)hipSetDevice(dev);
            (void)hipFree(p);
        }
    }
};

} // namespace

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
    }
    if (exchange == PTX_EXCHANGE_RCCL)
        for (int i = 0; i < n; i++)
            for (int j = 0; j < i; j++)
                if (ctxs[i]->device() == ctxs[j]->device())
                    throw std::runtime_error("arg: an RCCL communicator takes every device once (two contexts share a GPU: use peer copies)");
    const Scene &sc = ctxs[0]->scene();
    const int64_t n_val = static_cast<int64_t>(sc.width) * sc.height * 3;
    const size_t bytes = static_cast<size_t>(n_val) * sizeof(double);
    const int spp = base.spp > 0 ? base.spp : sc.spp; // samples of the whole frame
    std::vector<Buffer> buf(n);
    std::vector<std::string> err(n);
    std::vector<std::thread> th;
    for (int i = 0; i < n; i++)
        th.emplace_back([&, i]() {
            try
            {
                buf[i].alloc(ctxs[i]->device(), bytes);
                RenderOpts o = base;
                o.accumulate = false;
                o.spp_total = base.spp_total > 0 ? base.spp_total : spp;
                if (scheme == PTX_SHARD_SAMPLES)
                {
                    // contiguous sample ranges that tile [0, spp)
                    const int q = spp / n, r = spp % n;
                    o.spp = q + (i < r ? 1 : 0);
                    o.sample_offset = base.sample_offset + i * q + (i < r ? i : r);
                }
                else
                {
                    o.spp = spp;
                    o.row_stride = n, o.row_phase = i, o.band = base.band > 0 ? base.band : 8;
                }
                RenderStats st;
                if (o.spp > 0)
                    ctxs[i]->render(o, static_cast<double *>(buf[i].p), true, st);
                else
                    HIP_CHECK(hipMemsetAsync(buf[i].p, 0, bytes, static_cast<hipStream_t>(ctxs[i]->stream())));
                HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(ctxs[i]->stream())));
                if (stats)
                    stats[i] = st;
            }
            catch (const std::exception &e)
            {
                err[i] = e.what();
            }
        });
    for (auto &t : th)
        t.join();
    for (int i = 0; i < n; i++)
        if (!err[i].empty())
            throw std::runtime_error(err[i]);

    // ---- the one exchange: sums of all devices -> device of context 0 ---------------------------
    hipStream_t s0 = static_cast<hipStream_t>(ctxs[0]->stream());
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
        g_rccl.check(g_rccl.GroupStart(), "ncclGroupStart");
        for (int i = 0; i < n; i++)
        {
            HIP_CHECK(hipSetDevice(devs[i]));
            g_rccl.check(g_rccl.Reduce(buf[i].p, buf[i].p, static_cast<size_t>(n_val), ncclDouble, ncclSum, 0, g_comms[i],
                                       static_cast<hipStream_t>(ctxs[i]->stream())),
                         "ncclReduce");
        }
        g_rccl.check(g_rccl.GroupEnd(), "ncclGroupEnd");
        for (int i = 0; i < n; i++)
        {
            HIP_CHECK(hipSetDevice(devs[i]));
            HIP_CHECK(hipStreamSynchronize(static_cast<hipStream_t>(ctxs[i]->stream())));
        }
    }
    else if (n > 1)
    {
        // peer copies into a staging buffer on device 0, added in device order (a fixed sum order)
        Buffer stage;
        stage.alloc(ctxs[0]->device(), bytes);
        for (int i = 1; i < n; i++)
        {
            HIP_CHECK(hipMemcpyPeerAsync(stage.p, ctxs[0]->device(), buf[i].p, ctxs[i]->device(), bytes, s0));
            hipLaunchKernelGGL(k_add_sums, dim3(static_cast<unsigned>((n_val + 255) / 256)), dim3(256), 0, s0,
                               static_cast<double *>(buf[0].p), static_cast<const double *>(stage.p), n_val);
            HIP_CHECK(hipGetLastError());
        }
        HIP_CHECK(hipStreamSynchronize(s0));
    }
    HIP_CHECK(hipSetDevice(ctxs[0]->device()));
    HIP_CHECK(hipMemcpyAsync(rgb_sum, buf[0].p, bytes, hipMemcpyDeviceToHost, s0));
    HIP_CHECK(hipStreamSynchronize(s0));
}

} // namespace ptx
