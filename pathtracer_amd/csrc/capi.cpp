// extern "C" boundary of libptx_amd.so: translates exceptions into status codes, owns the handles.
#include "../../include/ptx.h"
#include "context.hpp"
#include "plane_records.hpp"
#include "scene.hpp"
#include "sum_order.hpp"

#include <cmath>
#include <cstring>
#include <memory>
#include <new>
#include <stdexcept>
#include <algorithm>
#include <string>
#include <vector>

struct ptx_scene
{
    ptx::Scene sc;
};
struct ptx_context
{
    ptx::Context *ctx = nullptr;
};

namespace
{
thread_local std::string g_last_error;

int fail(const std::exception &e)
{
    g_last_error = e.what();
    const std::string &m = g_last_error;
    if (m.rfind("io:", 0) == 0)
        return PTX_ERR_IO;
    if (m.rfind("parse:", 0) == 0 || m.rfind("json:", 0) == 0)
        return PTX_ERR_PARSE;
    if (m.rfind("device:", 0) == 0)
        return PTX_ERR_DEVICE;
    if (m.rfind("state:", 0) == 0)
        return PTX_ERR_STATE;
    return PTX_ERR_ARG;
}
int fail_arg(const char *what)
{
    g_last_error = std::string("arg: ") + what;
    return PTX_ERR_ARG;
}
} // namespace

#define PTX_TRY try {
#define PTX_CATCH                                                                                        \
    }                                                                                                    \
    catch (const std::bad_alloc &)                                                                       \
    {                                                                                                    \
        g_last_error = "device: out of host memory";                                                     \
        return PTX_ERR_DEVICE;                                                                           \
    }                                                                                                    \
    catch (const std::exception &e) { return fail(e); }                                                  \
    catch (...)                                                                                          \
    {                                                                                                    \
        g_last_error = "unknown error";                                                                  \
        return PTX_ERR_ARG;                                                                              \
    }                                                                                                    \
    return PTX_OK;

extern "C" {

const char *ptx_last_error(void) { return g_last_error.c_str(); }
#define PTX_STR_(x) #x
#define PTX_STR(x) PTX_STR_(x)
const char *ptx_version(void) { return "pathtracer_amd 0.2 (gfx950, sum order " PTX_STR(PTX_SUM) ")"; }

int ptx_scene_load_conf(const char *conf_path, const char *root_dir, const ptx_overrides *ov, ptx_scene **out)
{
    if (!conf_path || !out)
        return fail_arg("ptx_scene_load_conf: null argument");
    *out = nullptr;
    PTX_TRY
    ptx::Overrides o;
    if (ov)
    {
        o.width = ov->width, o.height = ov->height, o.spp = ov->spp, o.depth = ov->depth;
        if (ov->model)
            o.model = ov->model;
        o.full_region = ov->full_region != 0;
    }
    std::unique_ptr<ptx_scene> s(new ptx_scene());
    ptx::load_conf(conf_path, root_dir ? root_dir : "", o, s->sc);
    *out = s.release();
    PTX_CATCH
}

int ptx_scene_from_arrays(const float *verts, const float *normals, const int32_t *mat_ids, int32_t n_tris,
                          const double *mats, int32_t n_mats, const double *lights, int32_t n_lights,
                          const double *view16, const double *cam_pos3, double fov_deg, double near_plane, int32_t width,
                          int32_t height, int32_t depth, int32_t spp, int32_t enable_indirect, int32_t enable_bary_normal,
                          ptx_scene **out)
{
    if (!out || n_tris < 0 || n_mats < 0 || n_lights < 0 || (n_tris > 0 && (!verts || !mat_ids)) || !view16 || !cam_pos3)
        return fail_arg("ptx_scene_from_arrays: null or negative argument");
    if ((n_mats > 0 && !mats) || (n_lights > 0 && !lights))
        return fail_arg("ptx_scene_from_arrays: null material or light table");
    if (width <= 0 || height <= 0 || depth <= 0 || spp <= 0)
        return fail_arg("ptx_scene_from_arrays: width, height, depth and spp must be positive");
    *out = nullptr;
    PTX_TRY
    std::unique_ptr<ptx_scene> s(new ptx_scene());
    ptx::Scene &sc = s->sc;
    sc.verts.assign(verts, verts + static_cast<size_t>(n_tris) * 9);
    for (float x : sc.verts)
        if (!std::isfinite(x))
            throw std::runtime_error("arg: vertex coordinate is not finite");
    sc.has_normals = normals != nullptr;
    if (normals)
        sc.normals.assign(normals, normals + static_cast<size_t>(n_tris) * 9);
    else
        sc.normals.assign(static_cast<size_t>(n_tris) * 9, 0.f);
    sc.mat_ids.assign(mat_ids, mat_ids + n_tris);
    sc.vert_ids.assign(static_cast<size_t>(n_tris) * 3, -1);
    for (int i = 0; i < n_tris; i++)
        if (sc.mat_ids[i] >= n_mats || sc.mat_ids[i] < -1)
            throw std::runtime_error("arg: material id out of range (-1 = no material)");
    for (int i = 0; i < n_mats; i++)
    {
        ptx::Material m;
        const double *p = mats + static_cast<size_t>(i) * 12;
        for (int c = 0; c < 3; c++)
            m.Ka[c] = p[c], m.Kd[c] = p[3 + c], m.Ks[c] = p[6 + c];
        m.Ni = p[9], m.Ns = p[10], m.illum = p[11];
        sc.mats.push_back(m);
    }
    for (int i = 0; i < n_lights; i++)
    {
        ptx::Light L;
        const double *p = lights + static_cast<size_t>(i) * 16;
        L.type = p[0] == 0.0 ? 0 : 1;
        for (int c = 0; c < 3; c++)
            L.radiance[c] = p[1 + c];
        if (L.type == 0)
        {
            for (int k = 0; k < 4; k++)
                for (int c = 0; c < 3; c++)
                    L.p[k][c] = p[4 + 3 * k + c];
        }
        else
        {
            for (int c = 0; c < 3; c++)
                L.center[c] = p[4 + c];
            L.radius = p[7];
        }
        sc.lights.push_back(L);
    }
    std::memcpy(sc.view, view16, sizeof(double) * 16);
    for (int c = 0; c < 3; c++)
        sc.cam_pos[c] = cam_pos3[c];
    sc.fov_deg = fov_deg, sc.near_plane = near_plane;
    sc.width = width, sc.height = height, sc.depth = depth, sc.spp = spp;
    sc.enable_indirect = enable_indirect != 0, sc.enable_bary_normal = enable_bary_normal != 0;
    sc.region[0] = 0, sc.region[1] = 0, sc.region[2] = height, sc.region[3] = width;
    sc.update_camera();
    *out = s.release();
    PTX_CATCH
}

void ptx_scene_destroy(ptx_scene *s) { delete s; }

int ptx_scene_get_info(const ptx_scene *s, ptx_scene_info *o)
{
    if (!s || !o)
        return fail_arg("ptx_scene_get_info: null argument");
    const ptx::Scene &sc = s->sc;
    o->n_tris = sc.n_tris(), o->n_mats = static_cast<int32_t>(sc.mats.size());
    o->n_lights = static_cast<int32_t>(sc.lights.size()), o->n_square_lights = sc.n_square_lights();
    o->width = sc.width, o->height = sc.height, o->depth = sc.depth, o->spp = sc.spp;
    o->enable_indirect = sc.enable_indirect, o->enable_bary_normal = sc.enable_bary_normal;
    o->enable_path_tracing = sc.enable_path_tracing;
    for (int i = 0; i < 4; i++)
        o->region[i] = sc.region[i];
    o->accel_type = sc.accel_type;
    o->fov_deg = sc.fov_deg, o->near_plane = sc.near_plane, o->world_scale = sc.world_scale;
    return PTX_OK;
}

int ptx_scene_copy_geometry(const ptx_scene *s, float *verts, float *normals, int32_t *mat_ids)
{
    if (!s)
        return fail_arg("ptx_scene_copy_geometry: null scene");
    const ptx::Scene &sc = s->sc;
    if (verts)
        std::memcpy(verts, sc.verts.data(), sc.verts.size() * sizeof(float));
    if (normals)
        std::memcpy(normals, sc.normals.data(), sc.normals.size() * sizeof(float));
    if (mat_ids)
        std::memcpy(mat_ids, sc.mat_ids.data(), sc.mat_ids.size() * sizeof(int32_t));
    return PTX_OK;
}

int ptx_scene_copy_vertex_ids(const ptx_scene *s, int32_t *vertex_ids)
{
    if (!s || !vertex_ids)
        return fail_arg("ptx_scene_copy_vertex_ids: null argument");
    std::memcpy(vertex_ids, s->sc.vert_ids.data(), s->sc.vert_ids.size() * sizeof(int32_t));
    return PTX_OK;
}

int ptx_scene_copy_tables(const ptx_scene *s, double *mats12, double *lights16)
{
    if (!s)
        return fail_arg("ptx_scene_copy_tables: null scene");
    const ptx::Scene &sc = s->sc;
    if (mats12)
        for (size_t i = 0; i < sc.mats.size(); i++)
        {
            double *p = mats12 + i * 12;
            const ptx::Material &m = sc.mats[i];
            for (int c = 0; c < 3; c++)
                p[c] = m.Ka[c], p[3 + c] = m.Kd[c], p[6 + c] = m.Ks[c];
            p[9] = m.Ni, p[10] = m.Ns, p[11] = m.illum;
        }
    if (lights16)
        for (size_t i = 0; i < sc.lights.size(); i++)
        {
            double *p = lights16 + i * 16;
            const ptx::Light &L = sc.lights[i];
            std::memset(p, 0, sizeof(double) * 16);
            p[0] = L.type;
            for (int c = 0; c < 3; c++)
                p[1 + c] = L.radiance[c];
            if (L.type == 0)
            {
                for (int k = 0; k < 4; k++)
                    for (int c = 0; c < 3; c++)
                        p[4 + 3 * k + c] = L.p[k][c];
            }
            else
            {
                for (int c = 0; c < 3; c++)
                    p[4 + c] = L.center[c];
                p[7] = L.radius;
            }
        }
    return PTX_OK;
}

int ptx_scene_copy_camera(const ptx_scene *s, double *view16, double *cam_pos3, double *raster2world16)
{
    if (!s)
        return fail_arg("ptx_scene_copy_camera: null scene");
    if (view16)
        std::memcpy(view16, s->sc.view, sizeof(double) * 16);
    if (cam_pos3)
        std::memcpy(cam_pos3, s->sc.cam_pos, sizeof(double) * 3);
    if (raster2world16)
        std::memcpy(raster2world16, s->sc.raster2world, sizeof(double) * 16);
    return PTX_OK;
}

const char *ptx_scene_result_path(const ptx_scene *s) { return s ? s->sc.result_path.c_str() : ""; }

int ptx_scene_plane_records(const ptx_scene *s, int32_t *n_records, int32_t *n_pairs, int32_t *rec_tri64, uint32_t *blockers32)
{
    if (!s)
        return fail_arg("ptx_scene_plane_records: null scene");
    PTX_TRY
    ptx::PlaneRecords pr;
    ptx::build_plane_records(s->sc.verts.data(), s->sc.n_tris(), pr);
    if (n_records)
        *n_records = pr.n_recs;
    if (n_pairs)
        *n_pairs = pr.n_pairs;
    if (rec_tri64)
        for (int k = 0; k < 64; k++)
            rec_tri64[k] = pr.rec_tri[k];
    if (blockers32)
    {
        std::vector<uint32_t> blk(32, 0xFFFFFFFFu);
        if (pr.n_recs <= 32 && s->sc.lights.size() == 1 && s->sc.lights[0].type == 0)
        {
            double corners[4][3];
            ptx::square_light_corners(s->sc.lights[0], corners);
            blk = ptx::blocker_masks(s->sc.verts.data(), pr, corners);
        }
        for (int k = 0; k < 32; k++)
            blockers32[k] = blk[k];
    }
    return PTX_OK;
    PTX_CATCH
}

int ptx_scene_bvh_check(const ptx_scene *s, int32_t median, int64_t *info6)
{
    if (!s || !info6)
        return fail_arg("ptx_scene_bvh_check: null argument");
    PTX_TRY
    const ptx::Scene &sc = s->sc;
    ptx::Bvh b;
    // median: 0 = SAH, 1 = median splits, 2 = SAH with reference splitting (budget BVH_SPLIT_BUDGET)
    ptx::build_bvh(sc.verts.data(), sc.n_tris(), b, median == 1, median == 2 ? ptx::BVH_SPLIT_BUDGET : 0.0);
    const int F = sc.n_tris(), K = static_cast<int>(b.tri_order.size());
    std::vector<int> seen(std::max(K, 1), 0);
    // the boxes stored with the references of every face: [face] -> list of (lo, hi)
    struct RefBox { float lo[3], hi[3]; };
    std::vector<std::vector<RefBox>> boxes(std::max(F, 1));
    int64_t children = 0, violations = 0;
    // walk the tree: every child box must lie inside its parent's box for that child and the stack
    // bound must hold on every path
    struct Item { int32_t node; float lo[3], hi[3]; int depth_need; bool has_box; };
    std::vector<Item> todo;
    Item root{};
    root.node = 0, root.has_box = false, root.depth_need = 0;
    todo.push_back(root);
    int64_t worst_need = 0;
    while (!todo.empty())
    {
        Item it = todo.back();
        todo.pop_back();
        const ptx::BvhNode &nd = b.nodes[it.node];
        int n = 0;
        for (int i = 0; i < 4; i++)
            n += nd.child[i] != ptx::BVH_EMPTY;
        const int here = it.depth_need + std::max(n - 1, 0);
        worst_need = std::max<int64_t>(worst_need, here);
        for (int i = 0; i < 4; i++)
        {
            const int32_t ch = nd.child[i];
            if (ch == ptx::BVH_EMPTY)
                continue;
            children++;
            if (it.has_box)
                for (int k = 0; k < 3; k++)
                    if (nd.lo[k][i] < it.lo[k] || nd.hi[k][i] > it.hi[k])
                        violations++;
            if (ch >= 0)
            {
                Item c{};
                c.node = ch, c.has_box = true, c.depth_need = here;
                for (int k = 0; k < 3; k++)
                    c.lo[k] = nd.lo[k][i], c.hi[k] = nd.hi[k][i];
                todo.push_back(c);
            }
            else
            {
                const int k_ref = ~ch;
                if (k_ref < 0 || k_ref >= K || b.tri_order[k_ref] < 0 || b.tri_order[k_ref] >= F)
                {
                    violations++;
                    continue;
                }
                seen[k_ref]++;
                RefBox rb;
                for (int k = 0; k < 3; k++)
                    rb.lo[k] = nd.lo[k][i], rb.hi[k] = nd.hi[k][i];
                boxes[b.tri_order[k_ref]].push_back(rb);
            }
        }
    }
    int64_t refs = 0;
    for (int k = 0; k < K; k++)
    {
        refs += seen[k];
        if (seen[k] != 1)
            violations++; // every reference hangs in the tree exactly once
    }
    // every face is covered by the boxes of its references: its corners, and a grid of points over it,
    // each lie strictly inside at least one of them (a face that was not cut has one box holding all of it)
    for (int f = 0; f < F; f++)
    {
        const float *v = &sc.verts[static_cast<size_t>(f) * 9];
        if (boxes[f].empty())
        {
            violations++;
            continue;
        }
        const int G = boxes[f].size() == 1 ? 1 : 12;
        for (int i = 0; i <= G; i++)
            for (int j = 0; i + j <= G; j++)
            {
                const double wa = double(i) / G, wb = double(j) / G, wc = 1.0 - wa - wb;
                double p[3];
                for (int k = 0; k < 3; k++)
                    p[k] = wa * v[k] + wb * v[3 + k] + wc * v[6 + k];
                bool inside = false;
                for (const RefBox &rb : boxes[f])
                {
                    bool in = true;
                    for (int k = 0; k < 3; k++)
                        in = in && p[k] > rb.lo[k] && p[k] < rb.hi[k];
                    inside = inside || in;
                }
                if (!inside)
                    violations++;
            }
        if (G == 1)
            for (int j = 0; j < 3; j++)
                for (int k = 0; k < 3; k++)
                    if (!(v[j * 3 + k] > boxes[f][0].lo[k] && v[j * 3 + k] < boxes[f][0].hi[k]))
                        violations++;
    }
    if (worst_need > b.stack_need)
        violations++;
    info6[0] = static_cast<int64_t>(b.nodes.size());
    info6[1] = children;
    info6[2] = b.max_depth;
    info6[3] = b.stack_need;
    info6[4] = refs;
    info6[5] = violations;
    return PTX_OK;
    PTX_CATCH
}

int ptx_context_create(int32_t hip_device, ptx_context **out)
{
    if (!out)
        return fail_arg("ptx_context_create: null argument");
    *out = nullptr;
    PTX_TRY
    auto *c = new ptx_context();
    try
    {
        c->ctx = new ptx::Context(hip_device);
    }
    catch (...)
    {
        delete c;
        throw;
    }
    *out = c;
    PTX_CATCH
}

int ptx_context_set_stream(ptx_context *c, void *hip_stream)
{
    if (!c || !c->ctx)
        return fail_arg("ptx_context_set_stream: null context");
    c->ctx->set_stream(hip_stream);
    return PTX_OK;
}

int ptx_context_set_option(ptx_context *c, const char *name, int64_t value)
{
    if (!c || !c->ctx || !name)
        return fail_arg("ptx_context_set_option: null argument");
    const std::string n = name;
    if (n == "brute_force_limit")
        c->ctx->brute_force_limit = static_cast<int>(value);
    else if (n == "default_batch_paths" && value > 0)
        c->ctx->default_batch_paths = value;
    else if (n == "lds_stack_levels" && value >= 0 && value <= 96)
        c->ctx->lds_stack_levels = static_cast<int>(value);
    else if (n == "lds_nodes" && value >= -1 && value <= (1 << 20))
        c->ctx->lds_nodes = static_cast<int>(value);
    else if (n == "bvh_split" && value >= 0 && value <= 400)
        c->ctx->bvh_split = static_cast<int>(value);
    else if (n == "bvh_rings" && (value == 0 || value == 1))
        c->ctx->bvh_rings = value != 0;
    else if (n == "ring_batch" && value >= 1 && value <= 64)
        c->ctx->ring_batch = static_cast<int>(value);
    else if (n == "bvh_wavefront" && (value == 0 || value == 1))
        c->ctx->bvh_wavefront = value != 0;
    else if (n == "wf_refill" && value >= 1 && value <= 64)
        c->ctx->wf_refill = static_cast<int>(value);
    else if (n == "raycast_wavefront" && (value == 0 || value == 1))
        c->ctx->raycast_wavefront = value != 0;
    else if (n == "shadow_queue" && (value == 0 || value == 1))
        c->ctx->shadow_queue = value != 0;
    else if (n == "trace_blocks_per_cu" && value >= 0 && value <= 64)
        c->ctx->trace_blocks_per_cu = static_cast<int>(value);
    else if (n == "chunk_paths" && value >= 64 && value <= (1 << 16))
        c->ctx->chunk_paths = static_cast<int>(value);
    else
        return fail_arg("ptx_context_set_option: unknown option or bad value");
    return PTX_OK;
}

void ptx_context_destroy(ptx_context *c)
{
    if (c)
    {
        delete c->ctx;
        delete c;
    }
}

int ptx_build_accel(ptx_context *c, const ptx_scene *s)
{
    if (!c || !c->ctx || !s)
        return fail_arg("ptx_build_accel: null argument");
    PTX_TRY
    c->ctx->build_accel(s->sc);
    PTX_CATCH
}

int ptx_accel_info(const ptx_context *c, int32_t *info5)
{
    if (!c || !c->ctx || !info5)
        return fail_arg("ptx_accel_info: null argument");
    if (!c->ctx->has_scene())
    {
        g_last_error = "state: ptx_build_accel has not been called on this context";
        return PTX_ERR_STATE;
    }
    const ptx::Bvh &b = c->ctx->bvh();
    info5[0] = static_cast<int32_t>(b.nodes.size());
    info5[1] = b.n_leaves, info5[2] = b.max_depth, info5[3] = b.stack_need;
    info5[4] = c->ctx->brute_force() ? 1 : 0;
    return PTX_OK;
}

int ptx_generate_rays(ptx_context *c, double *rays_o, double *rays_d)
{
    if (!c || !c->ctx || !rays_o || !rays_d)
        return fail_arg("ptx_generate_rays: null argument");
    PTX_TRY
    c->ctx->generate_rays(rays_o, rays_d);
    PTX_CATCH
}

int ptx_first_hit(ptx_context *c, int32_t *face_ids, double *hit_pts)
{
    if (!c || !c->ctx || !face_ids)
        return fail_arg("ptx_first_hit: null argument");
    PTX_TRY
    c->ctx->raycast(nullptr, nullptr, 0, face_ids, hit_pts, true, ptx::PTX_MODE_F64_);
    PTX_CATCH
}

int ptx_raycast(ptx_context *c, const double *rays_o, const double *rays_d, int64_t n, int32_t *face_ids, double *hit_pts)
{
    if (!c || !c->ctx || n < 0 || (n > 0 && (!rays_o || !rays_d || !face_ids)))
        return fail_arg("ptx_raycast: null argument");
    PTX_TRY
    c->ctx->raycast(rays_o, rays_d, n, face_ids, hit_pts, false, ptx::PTX_MODE_F64_);
    PTX_CATCH
}

int ptx_vistest(ptx_context *c, const double *p1, const double *p2, int64_t n, int32_t *visible)
{
    if (!c || !c->ctx || n < 0 || (n > 0 && (!p1 || !p2 || !visible)))
        return fail_arg("ptx_vistest: null argument");
    PTX_TRY
    c->ctx->vistest(p1, p2, n, visible);
    PTX_CATCH
}

// The caller's struct may be shorter (built against an older header) or longer (a newer one) than
// this library's: struct_size bytes are the caller's, fields beyond them read as zero.
static ptx::RenderOpts to_opts(const ptx_render_opts *o)
{
    ptx::RenderOpts ro;
    if (o)
    {
        if (o->struct_size < sizeof(uint32_t) + sizeof(int32_t))
            throw std::runtime_error("arg: ptx_render_opts.struct_size is not set (ABI version 2: write sizeof(ptx_render_opts) there)");
        ptx_render_opts in;
        std::memset(&in, 0, sizeof(in));
        std::memcpy(&in, o, std::min<size_t>(o->struct_size, sizeof(in)));
        ro.mode = in.mode;
        ro.spp = in.spp, ro.sample_offset = in.sample_offset, ro.spp_total = in.spp_total;
        ro.row_begin = in.row_begin, ro.row_end = in.row_end;
        ro.row_stride = in.row_stride, ro.row_phase = in.row_phase, ro.band = in.band;
        ro.depth = in.depth;
        ro.seed = in.seed;
        ro.batch_paths = in.batch_paths;
        ro.accumulate = in.accumulate != 0;
        ro.profile = in.profile != 0;
        ro.estimator = in.estimator;
        ro.enqueue_only = in.enqueue_only != 0;
    }
    return ro;
}

static void check_stats_arg(const ptx_stats *stats)
{
    if (stats && stats->struct_size < 2 * sizeof(uint32_t))
        throw std::runtime_error("arg: ptx_stats.struct_size is not set (ABI version 2: write sizeof(ptx_stats) there before the call)");
}

static void from_stats(const ptx::RenderStats &st, ptx_stats *stats)
{
    ptx_stats out;
    std::memset(&out, 0, sizeof(out));
    out.struct_size = stats->struct_size;
    out.paths = st.paths, out.segments = st.segments, out.shadow_rays = st.shadow_rays;
    out.passes = st.passes;
    out.wave_iterations = st.wave_iterations;
    out.records = st.records;
    out.bvh_nodes = st.bvh_nodes, out.bvh_tris = st.bvh_tris;
    out.ms_total = st.ms_total;
    out.ms_trace = st.ms_class[0], out.ms_accumulate = st.ms_class[2];
    out.ms_other = st.ms_class[3];
    out.state_bytes = st.state_bytes;
    std::memcpy(stats, &out, std::min<size_t>(stats->struct_size, sizeof(out)));
}

int ptx_render(ptx_context *c, const ptx_render_opts *o, double *rgb_sum, int32_t out_is_device, ptx_stats *stats)
{
    if (!c || !c->ctx || !rgb_sum)
        return fail_arg("ptx_render: null argument");
    PTX_TRY
    check_stats_arg(stats);
    ptx::RenderStats st;
    c->ctx->render(to_opts(o), rgb_sum, out_is_device != 0, st);
    if (stats)
        from_stats(st, stats);
    PTX_CATCH
}

int ptx_render_static(ptx_context *c, const ptx_render_opts *o, double *rgb_sum_device, ptx_stats *stats)
{
    if (!c || !c->ctx || !rgb_sum_device)
        return fail_arg("ptx_render_static: null argument");
    PTX_TRY
    check_stats_arg(stats);
    ptx::RenderStats st;
    c->ctx->render(to_opts(o), rgb_sum_device, true, st, /*static_state=*/true);
    if (stats)
        from_stats(st, stats);
    PTX_CATCH
}

int32_t ptx_abi_version(void) { return PTX_ABI_VERSION; }

const char *ptx_context_last_kernel(const ptx_context *c) { return (c && c->ctx) ? c->ctx->last_kernel().c_str() : ""; }

int ptx_render_sharded(ptx_context *const *contexts, int32_t n, const ptx_render_opts *o, int32_t scheme, int32_t exchange,
                       double *rgb_sum, ptx_stats *stats)
{
    if (!contexts || n < 1 || !rgb_sum)
        return fail_arg("ptx_render_sharded: null argument or no context");
    if ((scheme != PTX_SHARD_SAMPLES && scheme != PTX_SHARD_BANDS) || (exchange != PTX_EXCHANGE_RCCL && exchange != PTX_EXCHANGE_PEER))
        return fail_arg("ptx_render_sharded: unknown scheme or exchange");
    PTX_TRY
    std::vector<ptx::Context *> cs(n);
    for (int i = 0; i < n; i++)
    {
        if (!contexts[i] || !contexts[i]->ctx)
            throw std::runtime_error("arg: ptx_render_sharded: null context");
        cs[i] = contexts[i]->ctx;
    }
    check_stats_arg(stats);
    std::vector<ptx::RenderStats> st(n);
    ptx::render_sharded(cs.data(), n, to_opts(o), scheme, exchange, rgb_sum, st.data());
    if (stats)
    {
        // the caller's array has the stride of the caller's struct
        const size_t stride = stats->struct_size;
        for (int i = 0; i < n; i++)
        {
            ptx_stats *dst = reinterpret_cast<ptx_stats *>(reinterpret_cast<char *>(stats) + static_cast<size_t>(i) * stride);
            dst->struct_size = static_cast<uint32_t>(stride);
            from_stats(st[i], dst);
        }
    }
    PTX_CATCH
}

int ptx_finalize_image(const double *rgb_sum, int64_t n_pixels, double *rgb_out)
{
    if (!rgb_sum || !rgb_out || n_pixels < 0)
        return fail_arg("ptx_finalize_image: null argument");
    ptx::finalize_image(rgb_sum, n_pixels, rgb_out);
    return PTX_OK;
}

int ptx_write_ppm(const char *path, const double *rgb, int32_t width, int32_t height)
{
    if (!path || !rgb || width <= 0 || height <= 0)
        return fail_arg("ptx_write_ppm: bad argument");
    PTX_TRY
    ptx::write_ppm(path, rgb, width, height);
    PTX_CATCH
}

} // extern "C"
