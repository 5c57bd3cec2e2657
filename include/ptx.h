/*
 * ptx.h -- C ABI of the MI355X path-tracing core (libptx_amd.so).
 *
 * The reference (Rullec/PathTracer) has no plugin/FFI interface; the boundary this library sits
 * behind is the C++ class surface between its scene layer and its tracer.  Every entry point below
 * cites the reference interface it replaces (paths relative to the reference root).
 *
 * Conventions: plain C, opaque handles, int status returns (0 = PTX_OK), no exceptions cross the
 * boundary, caller-owned output buffers, one context per host thread / GPU.  After a non-zero
 * return ptx_last_error() describes the failure (thread-local).
 */
#ifndef PTX_H
#define PTX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PTX_OK 0
#define PTX_ERR_ARG 1      /* bad argument / null pointer                                       */
#define PTX_ERR_IO 2       /* file missing or unreadable                                        */
#define PTX_ERR_PARSE 3    /* malformed conf / OBJ / MTL                                        */
#define PTX_ERR_DEVICE 4   /* HIP error (no device, allocation, launch)                         */
#define PTX_ERR_STATE 5    /* call order violated (e.g. render before build_accel)              */

/* PTX_MODE_F64: double everywhere.  Camera rays and the ray / triangle test keep the reference's
 * operation order with no FMA contraction (cMathUtil.cpp:141-163: face ids and hit points are
 * bit-exact); the shading helpers (samplers, BxDFs, light terms, the clamp) use algebraically
 * equal, cheaper forms -- rsqrt, sincospi, a Rodrigues rotation, fused multiply-adds -- whose
 * results differ from a literal evaluation by ulps: per-pixel sums agree with it to 1e-9, except on
 * paths that meet an exact geometric tie, where an ulp can pick the other of two coincident faces.
 * PTX_MODE_F32: float state and tests, statistically equivalent images. */
#define PTX_MODE_F64 0
#define PTX_MODE_F32 1

/* Which estimator ptx_render runs.  REFERENCE is cPathTracer::RayTracePrimaryRay as written
 * (src/tracing/cPathTracer.cpp:555-704: every light sampled without a weight, the reference's
 * direction sampler, unshadowed sphere lights, the per-bounce clamp) -- what parity is defined
 * against.  FAST has no counterpart in the reference: the same scene description rendered by a
 * physically based forward path tracer -- light sampling weighted against BRDF sampling by the
 * balance heuristic, cosine / Phong-lobe importance sampling, Russian roulette from bounce 3 on,
 * sphere lights sampled on their cone WITH a visibility test, no per-bounce clamp.  The conf
 * lights emit but do not occlude (sampled directions that cross one collect it); faces of an
 * emissive material end a path and count when the camera sees them, directly or through glass.
 * The two estimators converge to different images. */
#define PTX_ESTIMATOR_REFERENCE 0
#define PTX_ESTIMATOR_FAST 1

typedef struct ptx_scene ptx_scene;
typedef struct ptx_context ptx_context;

/* Overrides of conf values (the benchmark configs change W/H/spp relative to the shipped confs).
 * A field <= 0 (or NULL) keeps the conf's value. */
typedef struct ptx_overrides
{
    int32_t width, height, spp, depth;
    const char *model;     /* replaces Scene.ModelName                                           */
    int32_t full_region;   /* != 0: DrawRegion := [0,0,H,W]                                      */
} ptx_overrides;

typedef struct ptx_scene_info
{
    int32_t n_tris, n_mats, n_lights, n_square_lights;
    int32_t width, height, depth, spp;
    int32_t enable_indirect, enable_bary_normal, enable_path_tracing;
    int32_t region[4];     /* stRow, stCol, edRow, edCol (inclusive)                             */
    int32_t accel_type;    /* 0 None, 1 AABB, 2 OCTREE (accepted for compatibility; the core always
                              uses its own BVH / brute-force intersector)                        */
    double fov_deg, near_plane, world_scale;
} ptx_scene_info;

/* ABI version of this header.  Version 2 put struct_size in front of ptx_render_opts and ptx_stats:
 * the caller writes sizeof(the struct it was compiled against) there, the library reads / writes
 * that many bytes and treats fields beyond them as zero, so that fields can be appended without
 * breaking callers built against an older header.  A struct_size of 0 is refused (PTX_ERR_ARG): it
 * marks a caller built against version 1, whose structs had no such field.  ptx_abi_version() returns
 * the version the loaded library was built with. */
#define PTX_ABI_VERSION 2

typedef struct ptx_render_opts
{
    uint32_t struct_size;  /* sizeof(ptx_render_opts) as the caller knows it                     */
    int32_t mode;          /* PTX_MODE_F64 | PTX_MODE_F32                                        */
    int32_t spp;           /* samples rendered by this call (<=0: scene spp)                     */
    int32_t sample_offset; /* global index of the first sample (sample-range sharding)           */
    int32_t spp_total;     /* divisor of the per-sample average (<=0: spp)                       */
    int32_t row_begin, row_end; /* rows [row_begin,row_end) rendered by this call; 0,0 = all     */
    int32_t row_stride, row_phase, band; /* interleaved bands: a row r is rendered iff
                              ((r / band) % row_stride) == row_phase; stride<=1 disables          */
    int32_t depth;         /* <=0: scene depth                                                   */
    uint64_t seed;         /* RNG seed (the reference's srand(13) -> default 13)                 */
    int64_t batch_paths;   /* paths in flight per pass (<=0: library default)                    */
    int32_t accumulate;    /* != 0: add to the sums already in the output buffer                 */
    int32_t profile;       /* != 0: time every kernel with HIP events (fills ms_*)                 */
    int32_t estimator;     /* PTX_ESTIMATOR_REFERENCE | PTX_ESTIMATOR_FAST                         */
    int32_t enqueue_only;  /* != 0: the call puts its work on the context's stream and returns without
                              waiting for it or reading anything back: no statistics (only passes and
                              state_bytes are filled), no timing, every bounce the depth allows is
                              launched (deep scenes otherwise stop early once a queue count read back
                              from the device says no ray is left).  Device output only.  With
                              ptx_render_static the call then contains nothing a stream capture
                              (hipStreamBeginCapture .. hipStreamEndCapture) refuses.               */
} ptx_render_opts;

typedef struct ptx_stats
{
    uint32_t struct_size;   /* sizeof(ptx_stats) as the caller knows it, set BEFORE the call     */
    uint32_t reserved_;
    uint64_t paths;         /* pixel-samples traced                                              */
    uint64_t segments;      /* closest-hit rays (one per bounce of every path)                   */
    uint64_t shadow_rays;   /* visibility tests issued                                           */
    uint64_t passes;        /* batches (tracer launches)                                         */
    uint64_t wave_iterations; /* 64-lane bounce iterations of the tracer: segments /
                               (64 * wave_iterations) is its active-lane fraction               */
    uint64_t records;       /* (direct, coefficient) records written = shaded segments           */
    uint64_t bvh_nodes;     /* BVH node records visited by all rays (0: all-triangles intersector) */
    uint64_t bvh_tris;      /* BVH triangle records visited (exact tests run by the traversal)   */
    double ms_total;        /* device time of the call, HIP events on the context's stream       */
    double ms_trace, ms_accumulate, ms_other; /* per-kernel device time (profiling
                               mode only, otherwise 0)                                           */
    uint64_t state_bytes;   /* device memory the call's path state took (record slots of the
                               tracer's waves, per-path results of a batch, rings, stack spill)  */
} ptx_stats;

/* ---- scene front end ----------------------------------------------------------------------
 * Replaces cRTScene::cRTScene + cRTScene::Init's mesh load (src/scene/cRTScene.cpp:6-39),
 * cMeshLoader::LoadObj (src/geometry/cMeshLoader.cpp:48-201), cBaseCamera::ParseConf / Reload
 * (src/render/camera/cBaseCamera.cpp:57-90,132-156), cPathTracer::ParseConfig
 * (src/tracing/cPathTracer.cpp:215-259) and BuildLight (src/tracing/cLight.cpp:266-358).
 * root_dir: directory the conf's relative paths are resolved against (the reference is run from
 * its repository root, src/main.cpp:30-32); NULL = current directory. */
int ptx_scene_load_conf(const char *conf_path, const char *root_dir, const ptx_overrides *ov,
                        ptx_scene **out);

/* Scene from caller arrays (no files).  verts/normals: [n_tris][3][3] float; mats: [n_mats][12]
 * double (Ka Kd Ks Ni Ns illum); lights: [n_lights][16] double (type Li[3] then 4 corners or
 * centre+radius); view16: row-major view matrix; replaces the same reference code as above. */
int ptx_scene_from_arrays(const float *verts, const float *normals, const int32_t *mat_ids,
                          int32_t n_tris, const double *mats, int32_t n_mats,
                          const double *lights, int32_t n_lights, const double *view16,
                          const double *cam_pos3, double fov_deg, double near_plane,
                          int32_t width, int32_t height, int32_t depth, int32_t spp,
                          int32_t enable_indirect, int32_t enable_bary_normal, ptx_scene **out);
void ptx_scene_destroy(ptx_scene *);

/* Introspection (what cObjMesh::GetFaceList / GetMaterial / cBaseCamera::GetViewMat expose,
 * src/geometry/cObjMesh.h, src/render/camera/cBaseCamera.hpp:26-31). Null outputs are skipped. */
int ptx_scene_get_info(const ptx_scene *, ptx_scene_info *out);
int ptx_scene_copy_geometry(const ptx_scene *, float *verts, float *normals, int32_t *mat_ids);
/* OBJ vertex index (0-based) of every face corner, [n_tris][3]: tFace::mVertexIdList,
 * src/geometry/cBaseMesh.h:76 (what the reference's .edge tables are built from). */
int ptx_scene_copy_vertex_ids(const ptx_scene *, int32_t *vertex_ids);
int ptx_scene_copy_tables(const ptx_scene *, double *mats12, double *lights16);
int ptx_scene_copy_camera(const ptx_scene *, double *view16, double *cam_pos3, double *raster2world16);
const char *ptx_scene_result_path(const ptx_scene *);
/* Acceleration data of the all-triangles intersector, for inspection and tests (no counterpart in
 * the reference): number of plane records and of coplanar triangle pairs among them, the
 * candidate-bit -> triangle table [64] (-1: none), and per record the set of records (bit mask)
 * that can block a segment between it and the light -- all ones unless the scene has exactly one
 * light and it is a square light.  Null outputs are skipped. */
int ptx_scene_plane_records(const ptx_scene *, int32_t *n_records, int32_t *n_pairs,
                            int32_t *rec_tri64, uint32_t *blockers32);
/* Builds the 4-wide BVH of the scene on the host (the structure that replaces the reference's grid
 * and octree, src/geometry/cAccelStruct.cpp:95-482, for scenes too large for the all-triangles
 * intersector) and checks it, for inspection and tests: info6 = number of nodes, number of
 * children, depth, stack bound, triangle references, violations (a child box outside its parent's,
 * a vertex outside its triangle's box, a triangle referenced other than once, a path through the
 * tree that can exceed the stack bound).  median != 0: median splits instead of SAH. */
int ptx_scene_bvh_check(const ptx_scene *, int32_t median, int64_t *info6);

/* ---- context / accel -------------------------------------------------------------------------
 * ptx_context_create + ptx_build_accel replace cPathTracer::Init (src/tracing/cPathTracer.cpp:20-57)
 * and cAccelStruct::Init (src/geometry/cAccelStruct.cpp:16-26, 95-179, 377-482).
 * hip_stream: a hipStream_t (NULL = the device's default stream). */
int ptx_context_create(int32_t hip_device, ptx_context **out);
int ptx_context_set_stream(ptx_context *, void *hip_stream);
void ptx_context_destroy(ptx_context *);
/* Tunables: "brute_force_limit" (scenes with at most this many triangles use the all-triangles
 * intersector instead of the BVH; default and maximum 64), "default_batch_paths" (paths per
 * tracer launch; default 2^25, less when the device's free memory asks for it), "chunk_paths" (path ids a wave of the tracer takes from the pool at
 * a time, 64..65536), "trace_blocks_per_cu" (workgroups of the tracer per compute unit; 0 = what
 * the occupancy calculator reports), "shadow_queue" (1/0: park the visibility tests of
 * one-square-light scenes in the per-wave LDS queue, or run them in place), "lds_stack_levels"
 * (BVH scenes: at most this many traversal stack levels per lane in LDS, deeper ones in a global
 * array; 0 = as many as fit at full occupancy), "bvh_rings" (1/0: BVH scenes with at most 12 lights that need
 * visibility tests are traced with per-wave hit and ray rings -- more paths in flight than lanes,
 * shading on full waves -- or with one path per lane), "ring_batch" (1..64: lanes of a wave of that
 * tracer that hand over hits / take rays together; default 40), "bvh_wavefront" (1/0, default 1: BVH
 * scenes whose lights need no visibility test under the reference's estimator, and BVH scenes with
 * up to 12 lights otherwise (square lights, the physically based estimator), are traced by separate traversal, shading and
 * resolve kernels with the segments in device-side ray queues, or by the ring tracer; its state
 * grows with the batch: about 400 bytes per path at depth 5 in double precision, plus 84 per light
 * and path where lights need visibility tests), "wf_refill" (1..64: idle lanes at which a wave of that traversal kernel takes
 * new rays; default 24), "lds_nodes" (BVH scenes: at most this many of the tree's first -- busiest
 * -- node records in LDS behind the stack levels; -1 = what fits next to 16 stack levels, 0 = none),
 * "bvh_split" (0..400, default 0: reference splitting in the BVH builder, at most this many extra
 * triangle references per 100 triangles), "raycast_wavefront" (1/0, a timing aid: ptx_raycast of unit-length directions through the
 * wavefront tracer's traversal kernel).  Results do not depend on any of
 * them.  Set before
 * ptx_build_accel. */
int ptx_context_set_option(ptx_context *, const char *name, int64_t value);
int ptx_build_accel(ptx_context *, const ptx_scene *);
/* BVH statistics of the last build: n_nodes (4-wide), n_leaves (= triangles), max_depth, stack_need (most
 * entries a traversal can hold on its stack), brute_force flag */
int ptx_accel_info(const ptx_context *, int32_t *info5);

/* ---- hot path ----------------------------------------------------------------------------------
 * ptx_generate_rays  replaces cPathTracer::GenerateRay (src/tracing/cPathTracer.cpp:129-213):
 *                    origins/directions [H*W][3] double, HOST buffers.
 * ptx_first_hit      replaces cAccelStruct::RayCast for the camera rays
 *                    (src/geometry/cAccelStruct.cpp:28-54): face id per pixel, -1 = miss; HOST
 *                    buffers; hit_pts may be NULL.
 * ptx_render         replaces cPathTracer::Process / RayTracing / RayTracePrimaryRay
 *                    (src/tracing/cPathTracer.cpp:65-86, 517-704): writes the per-pixel sums
 *                    sum_s final_s / spp_total BEFORE the last max-clamp, [H*W][3] double.
 *                    out_is_device != 0: rgb_sum is device memory on the context's GPU.
 * ptx_raycast / ptx_vistest  batched operator forms of cAccelStruct::RayCast / VisTest
 *                    (src/geometry/cAccelStruct.hpp:33-35) on HOST arrays of n rays / segments. */
int ptx_generate_rays(ptx_context *, double *rays_o, double *rays_d);
int ptx_first_hit(ptx_context *, int32_t *face_ids, double *hit_pts);
int ptx_render(ptx_context *, const ptx_render_opts *, double *rgb_sum, int32_t out_is_device,
               ptx_stats *stats);
int ptx_raycast(ptx_context *, const double *rays_o, const double *rays_d, int64_t n,
                int32_t *face_ids, double *hit_pts);
int ptx_vistest(ptx_context *, const double *p1, const double *p2, int64_t n, int32_t *visible);
/* Name of the tracer kernel the last ptx_render / ptx_render_static of this context launched, as
 * a profiler prints it ("k_trace_f64<true, true, 1>", "k_trace_rings_f64<true, false>", ...); ""
 * before the first render.  No counterpart in the reference; measurement tooling uses it to tie
 * counter profiles to the kernel that actually ran. */
const char *ptx_context_last_kernel(const ptx_context *);

/* ptx_render_static: ptx_render for callers that cannot allocate -- tight serving loops, harnesses
 * that forbid hipMalloc, code under HIP stream capture (there with opts->enqueue_only set: without
 * it the call waits for its stream to read the statistics back, which a capture refuses).  It has
 * no counterpart in the reference.
 * Same kernels, same results as ptx_render with out_is_device != 0 (rgb_sum is DEVICE memory,
 * [H*W][3] double, on the context's GPU), but the path state (record slots, per-path results,
 * statistics, rings, stack spill) lives in a region of 256 MiB that the library holds in every
 * GPU's memory from load to unload, and the call makes no hipMalloc / hipFree or other allocating
 * call (events and the pinned landing place of the statistics belong to the context).  Batches
 * and the tracer's grid are cut down to what fits the region, so large frames take more launches
 * than under ptx_render; a depth and image width whose smallest state does not fit is rejected with
 * PTX_ERR_ARG.  opts->profile is ignored.  All contexts of one GPU share that GPU's region and it
 * keeps the last call's contents: the call clears what must start at zero, and calls that overlap
 * on one GPU (two host threads, two streams) race -- serialise them. */
int ptx_render_static(ptx_context *, const ptx_render_opts *, double *rgb_sum_device, ptx_stats *stats);

/* ---- one frame over several GPUs of one node ---------------------------------------------------
 * ptx_render_sharded shards the row loop of cPathTracer::RayTracing (src/tracing/cPathTracer.cpp:
 * 526-529: OpenMP over rows inside one process) across n contexts, one per GPU, each driven by a
 * host thread of its own; every context must hold the same scene (ptx_build_accel).
 *   scheme   PTX_SHARD_SAMPLES: context i renders a contiguous range of the opts->spp samples of
 *            every pixel; PTX_SHARD_BANDS: its interleaved bands of opts->band rows (0 = 8) of
 *            the whole frame, band b going to context b mod n.
 *   exchange the one data exchange, after the render: the per-pixel sums of all contexts are
 *            reduced onto the GPU of context 0 by PTX_EXCHANGE_RCCL (ncclReduce, f64 sum, over
 *            xGMI; librccl.so is looked up at the first call; every context on a GPU of its own)
 *            or by PTX_EXCHANGE_PEER (peer copies added in context order: a fixed sum order;
 *            contexts may share a GPU).
 * rgb_sum: HOST buffer [H*W][3], the same pre-clamp sums as ptx_render writes (bands: bit for bit
 * those of a single ptx_render; sample ranges: up to the order of the additions).  stats: n
 * entries, one per context (each with its struct_size set; the stride is that of the first), or
 * NULL.  opts->row_stride/row_phase/accumulate are ignored.  Bands travel packed: only a context's
 * own rows, W * H * 24 / n bytes per context (ncclSend / ncclRecv in one group, or one peer copy
 * each); sample ranges reduce whole frames.  The contexts keep their frame buffers and host threads
 * from call to call: a frame loop allocates on its first frame only. */
#define PTX_SHARD_SAMPLES 0
#define PTX_SHARD_BANDS 1
#define PTX_EXCHANGE_RCCL 0
#define PTX_EXCHANGE_PEER 1
int ptx_render_sharded(ptx_context *const *contexts, int32_t n, const ptx_render_opts *opts, int32_t scheme,
                       int32_t exchange, double *rgb_sum, ptx_stats *stats);

/* ---- image output --------------------------------------------------------------------------
 * ptx_finalize_image: the last per-pixel max-clamp (src/tracing/cPathTracer.cpp:701-702), host.
 * ptx_write_ppm: cPathTracer::OutputImage (src/tracing/cPathTracer.cpp:261-284): "P3\nW H\n255\n"
 * then int(c*255) triplets separated by single spaces on one line. */
int ptx_finalize_image(const double *rgb_sum, int64_t n_pixels, double *rgb_out);
int ptx_write_ppm(const char *path, const double *rgb, int32_t width, int32_t height);

const char *ptx_last_error(void);
const char *ptx_version(void);
/* PTX_ABI_VERSION of the loaded library (no counterpart in the reference; callers compare it with
 * the header they were built against before they pass structs). */
int32_t ptx_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* PTX_H */
