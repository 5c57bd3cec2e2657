// One rendering context per GPU: device copies of the scene and the per-batch path state.
#pragma once
#include "bvh.hpp"
#include "scene.hpp"

#include <hip/hip_runtime.h>
#include <cstdint>
#include <string>
#include <vector>

namespace ptx
{

constexpr int PTX_MODE_F64_ = 0;
constexpr int PTX_MODE_F32_ = 1;
constexpr int PTX_ESTIMATOR_REFERENCE_ = 0; // the reference's estimator: what parity is defined against
constexpr int PTX_ESTIMATOR_FAST_ = 1;      // the physically based one (MIS, lobe sampling, roulette, shadowed sphere lights)

struct RenderOpts
{
    int mode = PTX_MODE_F64_;
    int spp = 0, sample_offset = 0, spp_total = 0;
    int row_begin = 0, row_end = 0;
    int row_stride = 1, row_phase = 0, band = 1;
    int depth = 0;
    uint64_t seed = 13;
    int64_t batch_paths = 0;
    bool accumulate = false;
    bool profile = false;
    int estimator = PTX_ESTIMATOR_REFERENCE_;
    bool enqueue_only = false; // no wait, no read-back, no statistics: the call only puts work on the stream
};

struct RenderStats
{
    uint64_t paths = 0, segments = 0, shadow_rays = 0, passes = 0, wave_iterations = 0, records = 0, bvh_nodes = 0, bvh_tris = 0;
    double ms_total = 0;
    double ms_class[4] = {0, 0, 0, 0}; // trace, (unused), accumulate, other
    uint64_t state_bytes = 0;          // device memory of the call's path state (records, results, rings, ...)
};

// per-kernel-class timing with HIP events on the render stream (profiling mode only)
struct RenderTimers
{
    enum
    {
        TRACE = 0,
        RESOLVE = 1,
        ACCUMULATE = 2,
        OTHER = 3,
        END = 4
    };
    std::vector<hipEvent_t> events;
    std::vector<int> classes;
    size_t used = 0;
    void mark(hipStream_t s, int cls);
    void collect(double ms[4]);
    ~RenderTimers();
};

class Context
{
public:
    explicit Context(int device);
    ~Context();
    Context(const Context &) = delete;
    Context &operator=(const Context &) = delete;

    void set_stream(void *hip_stream);
    void build_accel(const Scene &sc);
    void generate_rays(double *rays_o, double *rays_d);
    void raycast(const double *ro, const double *rd, int64_t n, int32_t *ids, double *pts, bool from_camera, int mode);
    void vistest(const double *p1, const double *p2, int64_t n, int32_t *vis);
    // static_state: the path state lives in the library's static device region instead of
    // allocated memory, and the call allocates nothing (ptx_render_static)
    void render(const RenderOpts &o, double *sum, bool out_is_device, RenderStats &stats, bool static_state = false);
    // name of the tracer kernel the last render launched, as profilers print it
    const std::string &last_kernel() const { return last_kernel_; }

    const Bvh &bvh() const { return bvh_; }
    bool brute_force() const { return brute_; }
    // LDS the traversal stacks of one workgroup of a BVH kernel take (0: all-triangles intersector)
    size_t stack_bytes() const { return brute_ ? 0 : static_cast<size_t>(bvh_.stack_need > 1 ? bvh_.stack_need : 1) * 256 * 4; }
    const Scene &scene() const { return scene_; }
    bool has_scene() const { return have_scene_; }
    int device() const { return device_; }
    void *stream() const { return stream_; }

    // scenes up to this many triangles are intersected by the all-triangles loop
    int brute_force_limit = 64;
    int64_t default_batch_paths = int64_t(1) << 25;
    // path ids a wave of the tracer takes from the pool at a time (>= 64)
    int chunk_paths = 64;
    // workgroups of the tracer per compute unit (<= 0: what the occupancy calculator reports)
    int trace_blocks_per_cu = 0;
    // single-square-light scenes: park visibility tests in a per-wave LDS queue (full-wave tests)
    bool shadow_queue = true;
    // BVH tracer: at most this many traversal stack levels in LDS, deeper ones spill to global memory
    // (0: as many as fit at full occupancy)
    int lds_stack_levels = 0;
    // BVH tracer: at most this many node records (the first ones: the builder numbers the busiest
    // nodes first) in LDS behind the stack levels (-1: what fits next to 16 stack levels; 0: none)
    int lds_nodes = -1;
    int bvh_split = 0; // reference splitting: at most this many extra triangle references per 100 triangles (0: none, the default: fewer triangle tests, more node visits, no net gain measured)
    // BVH scenes with at most 12 square lights: the tracer that queues hits and rays per wave (ring tracer)
    // instead of the one that keeps every path on its lane
    bool bvh_rings = true;
    // ring tracer: lanes that hand over hits / take rays together (1..64)
    int ring_batch = 40;
    // BVH scenes without square lights, the reference's estimator: traversal and shading in separate
    // kernels (five waves per SIMD traverse instead of three) with the segments in two ray queues
    bool bvh_wavefront = true;
    // wavefront tracer: a wave of the traversal kernel fetches new rays when this many lanes are idle (1..64)
    int wf_refill = 24;
    // timing aid: ptx_raycast through the wavefront tracer's traversal kernel (unit-length directions)
    bool raycast_wavefront = false;

    struct Geometry;
    template <class R> struct Impl;
    // what the context keeps between frames rendered over several GPUs (sharded.hip): device buffers
    // and the host thread that drives this GPU; made on first use, freed with the context
    struct ShardSlot;
    ShardSlot &shard_slot();

private:
    void require_scene() const;
    template <class R> void raycast_wavefront_impl(Impl<R> &im, const double *ro, const double *rd, int64_t n, int32_t *ids, double *pts);
    template <class R> void render_impl(Impl<R> &im, const RenderOpts &o, double *d_sum, RenderStats &stats, bool static_state);

    int device_;
    int n_cus_ = 1;
    void *stream_ = nullptr;
    Scene scene_;
    Bvh bvh_;
    int id_bits_ = 1;
    bool brute_ = true, have_scene_ = false;
    Geometry *geo_ = nullptr;
    Impl<double> *f64_ = nullptr;
    Impl<float> *f32_ = nullptr;
    RenderTimers timers_;
    unsigned char *static_state_ = nullptr;     // device address of this GPU's copy of the static state region
    hipEvent_t ev_begin_ = nullptr, ev_end_ = nullptr;
    unsigned long long *host_stats_ = nullptr;  // pinned
    std::string last_kernel_;
    ShardSlot *shard_ = nullptr;
    static void free_shard_slot(ShardSlot *);
};

// One frame over n contexts (one per GPU, one host thread each) and the final reduce of the sums
// onto context 0's GPU; scheme / exchange: PTX_SHARD_* / PTX_EXCHANGE_* of ptx.h (sharded.hip)
void render_sharded(Context *const *ctxs, int n, const RenderOpts &base, int scheme, int exchange, double *rgb_sum, RenderStats *stats);

} // namespace ptx
