/*
 * pt_oracle.c -- CPU restatement of the Rullec/PathTracer hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This file is the checker, never the product: only tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py may load it.  It restates, in plain C and in double precision, the
 * arithmetic of the reference's integrator, intersector, BxDFs and lights, function by function.
 * Every function cites the reference file:line it follows (paths relative to the reference root).
 *
 * PARITY STATUS: "parity unpinned" at the last-ulp level.  The reference cannot be compiled here
 * (Eigen, glm, tinyobjloader are absent) and ships no tests or known-answer vectors for this path,
 * so the operation order of Eigen's 3x3 inverse / reductions is restated from knowledge of the
 * library (cofactors of column 0, det along column 0, one reciprocal) with left-to-right sums.
 * The only reference-owned artefacts that pin anything are the .edge caches (fan triangulation and
 * face numbering), checked in tests/test_scene_io.py.
 *
 * One deliberate substitution: the reference draws from the libc drand48() stream shared by all
 * OpenMP threads (not reproducible).  Here every draw is a pure function of
 * (seed, pixel, sample, bounce, dimension) -- see orc_rng() -- so that the GPU core can consume
 * exactly the same numbers.  The draw *sites* and their order are the reference's.
 *
 * Build: gcc -O2 -fopenmp -ffp-contract=off -fPIC -shared (see oracle/Makefile).
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#ifndef M_PI
#define M_PI 3.14159265358979323846
#endif

/* Order of the additions in the 3- and 4-term sums of the hot path (dot products, determinant,
 * row x vector, squared norms).  The reference leaves it to Eigen, whose source is not in its
 * checkout; the plausible orders are a build-time switch so that their effect can be counted
 * (tools/sum_order_report.py):
 *   ORC_SUM 0  left to right            (a + b) + c          ((a + b) + c) + d      -- the default
 *   ORC_SUM 1  last two first           a + (b + c)          (a + b) + (c + d)      -- Eigen's unrolled scalar redux
 *   ORC_SUM 2  as 2-wide packets        a + (b + c)          (a + c) + (b + d)      -- Eigen's vectorised redux of a
 *                                                                                       4-vector (points carry w = 0 or 1) */
#ifndef ORC_SUM
#define ORC_SUM 0
#endif
#if ORC_SUM == 0
#define SUM3(a, b, c) (((a) + (b)) + (c))
#define SUM4(a, b, c, d) ((((a) + (b)) + (c)) + (d))
#define SQ4(x, y, z) (((x) + (y)) + (z)) /* squared norm of a 4-vector whose w is 0 */
#elif ORC_SUM == 1
#define SUM3(a, b, c) ((a) + ((b) + (c)))
#define SUM4(a, b, c, d) (((a) + (b)) + ((c) + (d)))
#define SQ4(x, y, z) (((x) + (y)) + (z)) /* (x + y) + (z + 0) */
#else
#define SUM3(a, b, c) ((a) + ((b) + (c)))
#define SUM4(a, b, c, d) (((a) + (c)) + ((b) + (d)))
#define SQ4(x, y, z) (((x) + (z)) + (y)) /* (x + z) + (y + 0) */
#endif
int orc_sum_order(void) { return ORC_SUM; }

/* ------------------------------------------------------------------------------------------ */
/* scene description handed over by the Python front end (flat arrays, all double / int32)      */
/* ------------------------------------------------------------------------------------------ */
typedef struct
{
    int32_t n_tris;
    const double *verts;   /* [n_tris][3 verts][3]   positions, float-representable doubles     */
    const double *normals; /* [n_tris][3 verts][3]   vertex normals                              */
    const int32_t *mat_ids; /* [n_tris]  index into mats, -1 = no material                       */
    int32_t n_mats;
    const double *mats; /* [n_mats][12]: Ka[3] Kd[3] Ks[3] Ni Ns illum                          */
    int32_t n_lights;
    const double *lights; /* [n_lights][16]: type(0 square,1 sphere) Li[3]; square: P0..P3 (12);
                             sphere: C[3] R (rest unused)                                         */
    /* camera */
    int32_t width, height;
    double fov_deg, near_;  /* (double)(float) values, cBaseCamera.hpp:44-47                     */
    double view[16];        /* row-major 4x4, entries are (double)(float) of glm::lookAt          */
    double cam_pos[3];      /* CameraPos in double (cPathTracer.cpp:184)                          */
    /* tracer */
    int32_t depth, spp;
    int32_t enable_indirect, enable_bary_normal;
    int32_t region[4]; /* stRow, stCol, edRow, edCol, inclusive (cPathTracer.cpp:532-537)        */
    uint64_t seed;
    int32_t sample_offset; /* global index of the first sample rendered (sample-range sharding)   */
    int32_t spp_total;     /* divisor of the per-sample average (== spp when not sharded)         */
    int32_t use_bvh;       /* 0: brute force over all faces (reference accel "None")             */
    int32_t estimator;     /* 0: the reference's estimator; 1: the physically based one ("fast")  */
    int32_t fast_strategy; /* fast estimator, for its own tests: 0 both sampling strategies with
                              the balance heuristic; 1 light sampling alone; 2 BRDF sampling alone
                              (the three agree in the mean wherever no delta lobe is involved)     */
} orc_scene;

typedef struct
{
    uint64_t paths, segments, closest_rays, shadow_rays, tri_tests;
} orc_stats;

/* optional BVH (built by orc_bvh_build, used when scene->use_bvh) -- same hit semantics as the
 * brute-force loops, it only skips faces whose padded box the ray misses.                      */
typedef struct
{
    double lo[3], hi[3];
    int32_t left, right; /* internal: children; leaf: left = -1 - first, right = count */
} orc_bvh_node;
static orc_bvh_node *g_bvh_nodes = NULL;
static int32_t *g_bvh_order = NULL;
static int32_t g_bvh_n = 0;

/* ------------------------------------------------------------------------------------------ */
/* RNG: counter based, replaces drand48() (cMathUtil.cpp:236,275,318-319; cLight.cpp:75;        */
/* cBxDF.cpp:193).  53-bit uniform in [0,1).                                                    */
/* ------------------------------------------------------------------------------------------ */
static inline uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t orc_rng_bits(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim)
{
    uint64_t h = mix64(seed + 0x9E3779B97F4A7C15ULL * ((uint64_t)pixel + 1));
    /* dimension split round the bounce field (bits 8-23) so that dimensions from 256 on (scenes
     * with 64 or more lights) do not run into it */
    h = mix64(h ^ (((uint64_t)sample << 32) | ((uint64_t)(dim >> 8) << 24) | ((uint64_t)bounce << 8) |
                   (uint64_t)(dim & 255u)));
    return h;
}

double orc_rng(uint64_t seed, uint32_t pixel, uint32_t sample, uint32_t bounce, uint32_t dim)
{
    return (double)(orc_rng_bits(seed, pixel, sample, bounce, dim) >> 11) * (1.0 / 9007199254740992.0);
}

/* ------------------------------------------------------------------------------------------ */
/* small vector helpers (3 components; the reference's 4th component is 0/1 and never matters)  */
/* ------------------------------------------------------------------------------------------ */
static inline double dot3(const double *a, const double *b)
{
    return SUM3(a[0] * b[0], a[1] * b[1], a[2] * b[2]);
}
static inline void cross3(const double *a, const double *b, double *o)
{
    double x = a[1] * b[2] - a[2] * b[1];
    double y = a[2] * b[0] - a[0] * b[2];
    double z = a[0] * b[1] - a[1] * b[0];
    o[0] = x, o[1] = y, o[2] = z;
}
static inline double norm3(const double *a) { return sqrt(SQ4(a[0] * a[0], a[1] * a[1], a[2] * a[2])); }
/* Eigen normalized(): divides by the norm when squaredNorm > 0, else returns the vector as is */
static inline void normalize3(const double *a, double *o)
{
    double z = SQ4(a[0] * a[0], a[1] * a[1], a[2] * a[2]);
    if (z > 0)
    {
        double n = sqrt(z);
        o[0] = a[0] / n, o[1] = a[1] / n, o[2] = a[2] / n;
    }
    else
    {
        o[0] = a[0], o[1] = a[1], o[2] = a[2];
    }
}
static inline int has_nan3(const double *a) { return isnan(a[0]) || isnan(a[1]) || isnan(a[2]); }

/* 3x3 inverse, Eigen Matrix3d::inverse() restated: cofactors of column 0, det along column 0,
 * one reciprocal.  m is row-major m[r*3+c].  Used by cMathUtil.cpp:151,215 and cGeoUtil.cpp:220. */
static void inverse3(const double *m, double *inv)
{
    double c00 = m[4] * m[8] - m[5] * m[7];
    double c10 = m[7] * m[2] - m[8] * m[1];
    double c20 = m[1] * m[5] - m[2] * m[4];
    double det = SUM3(c00 * m[0], c10 * m[3], c20 * m[6]);
    double id = 1.0 / det;
    inv[0] = c00 * id, inv[1] = c10 * id, inv[2] = c20 * id;
    inv[3] = (m[5] * m[6] - m[3] * m[8]) * id;
    inv[4] = (m[8] * m[0] - m[6] * m[2]) * id;
    inv[5] = (m[2] * m[3] - m[0] * m[5]) * id;
    inv[6] = (m[3] * m[7] - m[4] * m[6]) * id;
    inv[7] = (m[6] * m[1] - m[7] * m[0]) * id;
    inv[8] = (m[0] * m[4] - m[1] * m[3]) * id;
}

/* ------------------------------------------------------------------------------------------ */
/* cMathUtil::RayCast / RayCastT (cMathUtil.cpp:141-163, 205-227)                               */
/* returns 1 and (beta,gamma,t) when accepted                                                   */
/* ------------------------------------------------------------------------------------------ */
static inline int tri_solve(const double *o, const double *d, const double *p1, const double *p2,
                            const double *p3, double *t_out)
{
    const double eps = 1e-10;
    double m[9], inv[9], r[3];
    for (int i = 0; i < 3; i++)
    {
        m[i * 3 + 0] = p1[i] - p2[i];
        m[i * 3 + 1] = p1[i] - p3[i];
        m[i * 3 + 2] = d[i];
        r[i] = p1[i] - o[i];
    }
    inverse3(m, inv);
    double beta = SUM3(inv[0] * r[0], inv[1] * r[1], inv[2] * r[2]);
    double gamma = SUM3(inv[3] * r[0], inv[4] * r[1], inv[5] * r[2]);
    double t = SUM3(inv[6] * r[0], inv[7] * r[1], inv[8] * r[2]);
    double alpha = 1 - beta - gamma;
    if (0 - eps < alpha && alpha < 1 + eps && 0 - eps < beta && beta < 1 + eps && 0 - eps < gamma &&
        gamma < 1 + eps && t > 0 - eps)
    {
        *t_out = t;
        return 1;
    }
    return 0;
}

/* cMathUtil::RayCast: writes the hit point o + t*d, returns accept flag */
int orc_tri_raycast(const double *o, const double *d, const double *p1, const double *p2,
                    const double *p3, double *pt)
{
    double t;
    if (!tri_solve(o, d, p1, p2, p3, &t))
    {
        pt[0] = pt[1] = pt[2] = NAN;
        return 0;
    }
    for (int i = 0; i < 3; i++)
        pt[i] = o[i] + t * d[i];
    return 1;
}

/* cMathUtil::RayCastT: returns t or NaN */
double orc_tri_raycast_t(const double *o, const double *d, const double *p1, const double *p2,
                         const double *p3)
{
    double t;
    if (!tri_solve(o, d, p1, p2, p3, &t))
        return NAN;
    return t;
}

/* ------------------------------------------------------------------------------------------ */
/* optional BVH: median-split over face centroids, boxes padded so that every face the          */
/* brute-force loop could accept is still visited.                                              */
/* ------------------------------------------------------------------------------------------ */
static const double *g_sort_verts;
static int g_sort_axis;
static int cmp_centroid(const void *a, const void *b)
{
    int ia = *(const int32_t *)a, ib = *(const int32_t *)b;
    const double *va = g_sort_verts + (size_t)ia * 9, *vb = g_sort_verts + (size_t)ib * 9;
    double ca = va[g_sort_axis] + va[3 + g_sort_axis] + va[6 + g_sort_axis];
    double cb = vb[g_sort_axis] + vb[3 + g_sort_axis] + vb[6 + g_sort_axis];
    if (ca < cb)
        return -1;
    if (ca > cb)
        return 1;
    return ia - ib;
}

static int32_t bvh_build_rec(const double *verts, int32_t first, int32_t count)
{
    int32_t me = g_bvh_n++;
    orc_bvh_node *nd = &g_bvh_nodes[me];
    for (int k = 0; k < 3; k++)
        nd->lo[k] = INFINITY, nd->hi[k] = -INFINITY;
    for (int32_t i = first; i < first + count; i++)
    {
        const double *v = verts + (size_t)g_bvh_order[i] * 9;
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++)
            {
                if (v[j * 3 + k] < nd->lo[k])
                    nd->lo[k] = v[j * 3 + k];
                if (v[j * 3 + k] > nd->hi[k])
                    nd->hi[k] = v[j * 3 + k];
            }
    }
    double ext = 0;
    for (int k = 0; k < 3; k++)
        if (nd->hi[k] - nd->lo[k] > ext)
            ext = nd->hi[k] - nd->lo[k];
    if (count <= 4)
    {
        nd->left = -1 - first;
        nd->right = count;
    }
    else
    {
        int axis = 0;
        for (int k = 1; k < 3; k++)
            if (nd->hi[k] - nd->lo[k] > nd->hi[axis] - nd->lo[axis])
                axis = k;
        g_sort_verts = verts;
        g_sort_axis = axis;
        qsort(g_bvh_order + first, (size_t)count, sizeof(int32_t), cmp_centroid);
        int32_t half = count / 2;
        int32_t l = bvh_build_rec(verts, first, half);
        int32_t r = bvh_build_rec(verts, first + half, count - half);
        nd = &g_bvh_nodes[me];
        nd->left = l;
        nd->right = r;
    }
    /* pad: barycentric slack 1e-10 of the face extent plus rounding of the hit reconstruction */
    double pad = 1e-7 * (ext + 1.0);
    for (int k = 0; k < 3; k++)
        nd->lo[k] -= pad, nd->hi[k] += pad;
    return me;
}

void orc_bvh_build(const orc_scene *sc)
{
    free(g_bvh_nodes);
    free(g_bvh_order);
    g_bvh_nodes = (orc_bvh_node *)malloc(sizeof(orc_bvh_node) * (size_t)(2 * sc->n_tris + 1));
    g_bvh_order = (int32_t *)malloc(sizeof(int32_t) * (size_t)(sc->n_tris + 1));
    for (int32_t i = 0; i < sc->n_tris; i++)
        g_bvh_order[i] = i;
    g_bvh_n = 0;
    if (sc->n_tris > 0)
        bvh_build_rec(sc->verts, 0, sc->n_tris);
}

/* conservative ray/box test in double: returns 1 when the ray (t >= -slack) may touch the box */
static inline int ray_box(const orc_bvh_node *nd, const double *o, const double *d, double tmax)
{
    double t0 = -1e-6, t1 = tmax;
    for (int k = 0; k < 3; k++)
    {
        if (d[k] == 0.0 || isnan(d[k]))
        {
            if (o[k] < nd->lo[k] || o[k] > nd->hi[k])
                return 0;
        }
        else
        {
            double a = (nd->lo[k] - o[k]) / d[k], b = (nd->hi[k] - o[k]) / d[k];
            if (a > b)
            {
                double s = a;
                a = b;
                b = s;
            }
            /* widen by a relative ulp-scale slack */
            a -= fabs(a) * 1e-12 + 1e-12;
            b += fabs(b) * 1e-12 + 1e-12;
            if (a > t0)
                t0 = a;
            if (b < t1)
                t1 = b;
            if (t0 > t1)
                return 0;
        }
    }
    return 1;
}

/* ------------------------------------------------------------------------------------------ */
/* cAccelStruct::RayCast (cAccelStruct.cpp:28-54): closest hit, faces in ascending id, strict    */
/* '<' so ties keep the lowest id; hits with dist <= 1e-6 ignored.  Returns face id or -1.       */
/* ------------------------------------------------------------------------------------------ */
/* Tie probe (test diagnostics, no counterpart in the reference): when `second` is given it
 * receives the smallest accepted distance (> 1e-6) of a face OTHER than the winner -- infinity
 * when there is none.  second - |pt - o| is the margin by which the winner won; a margin of a few
 * ulps means the choice hinges on the last bit of the ray (coplanar overlapping faces, shared
 * edges).  The BVH variant culls boxes beyond (1 + 1e-9) of the best distance, so it reports
 * runners-up inside that band exactly and may report a larger value (or infinity) otherwise. */
/* Tie choice (test diagnostics too): a search whose winner and runner-up lie within `gap` of each
 * other is a tie -- which of the two a renderer reports hinges on the last ulp of the ray.  Ties are
 * numbered along a path in the order they are met; bit k of `mask` set makes tie k return the
 * runner-up (its face, its point) instead of the winner, so that a test can re-trace the path every
 * way its ties can fall and name the one another renderer took. */
typedef struct
{
    uint32_t mask;
    int32_t n;  /* ties met so far */
    double gap;
} orc_tie;

typedef struct
{
    double dist, pt[3];
    int32_t face;
} orc_cand;

/* the reference's order: smaller distance wins, equal distances keep the lower face id (strict '<'
 * over ascending ids, cAccelStruct.cpp:39-51) */
static inline int cand_before(double n, int32_t f, const orc_cand *c)
{
    return n < c->dist || (n == c->dist && f < c->face);
}

static inline void cand_offer(orc_cand *best, orc_cand *snd, double n, int32_t f, const double *p)
{
    orc_cand c = {n, {p[0], p[1], p[2]}, f};
    if (best->face < 0 || cand_before(n, f, best))
    {
        *snd = *best;
        *best = c;
    }
    else if (snd->face < 0 || cand_before(n, f, snd))
        *snd = c;
}

static int32_t cand_result(orc_cand *best, const orc_cand *snd, double *pt, double *second, orc_tie *tie)
{
    if (second)
        *second = (best->face < 0 || snd->face < 0) ? INFINITY : snd->dist;
    if (tie && best->face >= 0 && snd->face >= 0 && snd->dist - best->dist <= tie->gap)
    {
        int32_t k = tie->n++;
        if (k < 32 && ((tie->mask >> k) & 1u))
            *best = *snd;
    }
    if (best->face >= 0)
        pt[0] = best->pt[0], pt[1] = best->pt[1], pt[2] = best->pt[2];
    return best->face;
}

static int32_t closest_brute(const orc_scene *sc, const double *o, const double *d, double *pt,
                             orc_stats *st, double *second, orc_tie *tie)
{
    orc_cand best = {1.7976931348623157e308, {NAN, NAN, NAN}, -1}, snd = best;
    pt[0] = pt[1] = pt[2] = NAN;
    for (int32_t f = 0; f < sc->n_tris; f++)
    {
        const double *v = sc->verts + (size_t)f * 9;
        double p[3];
        if (st)
            st->tri_tests++;
        if (!orc_tri_raycast(o, d, v, v + 3, v + 6, p))
            continue;
        if (has_nan3(p))
            continue;
        double e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
        double n = norm3(e);
        /* ascending ids: cand_before() is the reference's  n < dist  here */
        if (n > 1e-6)
            cand_offer(&best, &snd, n, f, p);
    }
    return cand_result(&best, &snd, pt, second, tie);
}

static int32_t closest_bvh(const orc_scene *sc, const double *o, const double *d, double *pt,
                           orc_stats *st, double *second, orc_tie *tie)
{
    orc_cand best = {1.7976931348623157e308, {NAN, NAN, NAN}, -1}, snd = best;
    pt[0] = pt[1] = pt[2] = NAN;
    int32_t stack[128];
    int sp = 0;
    stack[sp++] = 0;
    /* the winner is chosen by distance |p - o|, the boxes are culled by ray parameter: 1 / |d|
     * converts (the direction may have any length, tRay::Init does not normalise) */
    const double per_len = 1.0 / norm3(d);
    while (sp)
    {
        const orc_bvh_node *nd = &g_bvh_nodes[stack[--sp]];
        if (!ray_box(nd, o, d, best.dist * per_len * (1 + 1e-9) + 1e-9))
            continue;
        if (nd->left < 0)
        {
            int32_t first = -1 - nd->left;
            for (int32_t i = first; i < first + nd->right; i++)
            {
                int32_t f = g_bvh_order[i];
                const double *v = sc->verts + (size_t)f * 9;
                double p[3];
                if (st)
                    st->tri_tests++;
                if (!orc_tri_raycast(o, d, v, v + 3, v + 6, p) || has_nan3(p))
                    continue;
                double e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
                double n = norm3(e);
                /* same winner as the ascending-id strict-'<' loop: smaller dist, ties -> lower id */
                if (n > 1e-6)
                    cand_offer(&best, &snd, n, f, p);
            }
        }
        else if (sp + 2 <= 128)
        {
            stack[sp++] = nd->left;
            stack[sp++] = nd->right;
        }
    }
    return cand_result(&best, &snd, pt, second, tie);
}

int32_t orc_closest_hit(const orc_scene *sc, const double *o, const double *d, double *pt)
{
    if (sc->use_bvh && g_bvh_nodes)
        return closest_bvh(sc, o, d, pt, NULL, NULL, NULL);
    return closest_brute(sc, o, d, pt, NULL, NULL, NULL);
}

/* closest hit plus the runner-up's distance (the tie probe above) */
int32_t orc_closest_hit_probe(const orc_scene *sc, const double *o, const double *d, double *pt, double *second)
{
    if (sc->use_bvh && g_bvh_nodes)
        return closest_bvh(sc, o, d, pt, NULL, second, NULL);
    return closest_brute(sc, o, d, pt, NULL, second, NULL);
}

static int32_t closest(const orc_scene *sc, const double *o, const double *d, double *pt,
                       orc_stats *st, double *second, orc_tie *tie)
{
    if (st)
        st->closest_rays++;
    if (sc->use_bvh && g_bvh_nodes)
        return closest_bvh(sc, o, d, pt, st, second, tie);
    return closest_brute(sc, o, d, pt, st, second, tie);
}

/* ------------------------------------------------------------------------------------------ */
/* cAccelStruct::VisTest (cAccelStruct.cpp:56-87): p1 -> p2 blocked iff some face has            */
/* 1e-6 < t < |p1-p2| - 1e-6.  p1 == p2 gives a NaN direction, hence "visible".                  */
/* ------------------------------------------------------------------------------------------ */
static int vistest_impl(const orc_scene *sc, const double *p1, const double *p2, orc_stats *st)
{
    double e[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double dist = norm3(e);
    double cur_min = dist;
    double g[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]}, dir[3];
    /* Eigen normalized() on a zero vector leaves it zero -> singular system -> NaN t everywhere */
    normalize3(g, dir);
    if (sc->use_bvh && g_bvh_nodes)
    {
        int32_t stack[128];
        int sp = 0;
        stack[sp++] = 0;
        while (sp)
        {
            const orc_bvh_node *nd = &g_bvh_nodes[stack[--sp]];
            if (!ray_box(nd, p1, dir, dist))
                continue;
            if (nd->left < 0)
            {
                int32_t first = -1 - nd->left;
                for (int32_t i = first; i < first + nd->right; i++)
                {
                    const double *v = sc->verts + (size_t)g_bvh_order[i] * 9;
                    if (st)
                        st->tri_tests++;
                    double t = orc_tri_raycast_t(p1, dir, v, v + 3, v + 6);
                    if (!isnan(t) && t < dist - 1e-6 && t > 1e-6)
                        return 0;
                }
            }
            else if (sp + 2 <= 128)
            {
                stack[sp++] = nd->left;
                stack[sp++] = nd->right;
            }
        }
        return 1;
    }
    for (int32_t f = 0; f < sc->n_tris; f++)
    {
        const double *v = sc->verts + (size_t)f * 9;
        if (st)
            st->tri_tests++;
        double t = orc_tri_raycast_t(p1, dir, v, v + 3, v + 6);
        if (!isnan(t) && t < cur_min - 1e-6 && t > 1e-6)
        {
            cur_min = t;
            if (cur_min < dist)
                return 0;
        }
    }
    return 1;
}

int orc_vistest(const orc_scene *sc, const double *p1, const double *p2)
{
    return vistest_impl(sc, p1, p2, NULL);
}

/* ------------------------------------------------------------------------------------------ */
/* cGeoUtil (cGeoUtil.cpp:103-110, 117-127, 131-172, 174-231)                                    */
/* ------------------------------------------------------------------------------------------ */
void orc_normal_from_3pts(const double *p1, const double *p2, const double *p3, double *n)
{
    double a[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double b[3] = {p3[0] - p2[0], p3[1] - p2[1], p3[2] - p2[2]};
    double c[3];
    cross3(a, b, c);
    normalize3(c, n);
}

/* Reflect(normal, incoming) = 2 (n . -i) n + i */
void orc_reflect(const double *n, const double *in, double *out)
{
    double mi[3] = {-in[0], -in[1], -in[2]};
    double k = 2 * dot3(n, mi);
    for (int i = 0; i < 3; i++)
        out[i] = k * n[i] + in[i];
}

/* Refract(normal, incoming, ni): flips the normal (and inverts ni) when n.i > 0;
 * out = ni * (n x -(n x i)) - n * sqrt(1 - ni^2 |n x i|^2)   (NaN on total internal reflection) */
void orc_refract(const double *normal_, const double *in, double ni, double *out)
{
    double n[3] = {normal_[0], normal_[1], normal_[2]};
    if (dot3(n, in) > 0)
    {
        n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
        ni = 1.0 / ni;
    }
    double nxi[3], mnxi[3], c[3];
    cross3(n, in, nxi);
    mnxi[0] = -nxi[0], mnxi[1] = -nxi[1], mnxi[2] = -nxi[2];
    cross3(n, mnxi, c);
    double s = sqrt(1 - ni * ni * dot3(nxi, nxi));
    for (int i = 0; i < 3; i++)
        out[i] = ni * c[i] - n[i] * s;
}

/* CalcBarycentricCoordinate: w = [a b c]^-1 p (absolute positions as columns, rows that are all
 * zero regularised by 1e-6), then L2-normalised. */
void orc_barycentric(const double *p_, const double *a, const double *b, const double *c, double *w)
{
    double A[9], inv[9], p[3] = {p_[0], p_[1], p_[2]};
    for (int i = 0; i < 3; i++)
    {
        A[i * 3 + 0] = a[i];
        A[i * 3 + 1] = b[i];
        A[i * 3 + 2] = c[i];
    }
    for (int i = 0; i < 3; i++)
    {
        if (norm3(A + i * 3) < 1e-10)
        {
            A[i * 3 + 0] += 1e-6, A[i * 3 + 1] += 1e-6, A[i * 3 + 2] += 1e-6;
            p[i] += 1e-6;
        }
    }
    inverse3(A, inv);
    double r[3];
    for (int i = 0; i < 3; i++)
        r[i] = SUM3(inv[i * 3 + 0] * p[0], inv[i * 3 + 1] * p[1], inv[i * 3 + 2] * p[2]);
    normalize3(r, w);
}

/* ------------------------------------------------------------------------------------------ */
/* quaternion helpers (cMathUtil.cpp:40-55, 340-345; Eigen Quaternion * vector)                  */
/* q = (w, x, y, z)                                                                              */
/* ------------------------------------------------------------------------------------------ */
static void axis_angle_to_quat(const double *angvel, double *q)
{
    double th = norm3(angvel);
    if (th < 1e-7)
    {
        q[0] = 1, q[1] = q[2] = q[3] = 0;
        return;
    }
    double h = th / 2, c = cos(h), s = sin(h), u[3];
    normalize3(angvel, u);
    q[0] = c, q[1] = u[0] * s, q[2] = u[1] * s, q[3] = u[2] * s;
}
static void quat_rot(const double *q, const double *v, double *o)
{
    double uv[3], c2[3];
    cross3(q + 1, v, uv);
    uv[0] += uv[0], uv[1] += uv[1], uv[2] += uv[2];
    cross3(q + 1, uv, c2);
    for (int i = 0; i < 3; i++)
        o[i] = v[i] + q[0] * uv[i] + c2[i];
}

/* cMathUtil::SampleHemiSphereUniform (cMathUtil.cpp:230-267) with explicit xi inputs */
/* rotates `res`, a vector given about +Y, onto the frame whose +Y is n: the rotation of
 * cMathUtil::SampleHemiSphereUniform (cMathUtil.cpp:242-265) */
static void rot_from_y(const double *n, const double *res, double *out);

void orc_sample_hemisphere(const double *n, double xi1, double xi2, double *out)
{
    double theta = acos(1 - xi1), phi = 2 * M_PI * xi2;
    double res[3] = {cos(theta) * cos(phi), sin(theta), -cos(theta) * sin(phi)};
    rot_from_y(n, res, out);
}

static void rot_from_y(const double *n, const double *res, double *out)
{
    double y[3] = {0, 1, 0}, nmy[3] = {n[0] - y[0], n[1] - y[1], n[2] - y[2]}, ax0[3], axis[3];
    cross3(y, nmy, ax0);
    normalize3(ax0, axis);
    double angle = acos(dot3(y, n));
    if (!(fabs(norm3(axis) - 1) < 1e-10))
    {
        axis[0] = 1, axis[1] = 0, axis[2] = 0;
        if (fabs(n[1] - 1) < 1e-10)
            angle = 0;
        else if (fabs(n[1] + 1) < 1e-10)
            angle = M_PI;
    }
    double av[3] = {axis[0] * angle, axis[1] * angle, axis[2] * angle}, q[4];
    axis_angle_to_quat(av, q);
    quat_rot(q, res, out);
}

/* ------------------------------------------------------------------------------------------ */
/* lights (cLight.cpp:13-109, 188-255).  Returns pdf; writes the wi ray (origin, direction).     */
/* ------------------------------------------------------------------------------------------ */
double orc_square_light_sample(const double *L /*16*/, const double *ref, double xi1, double xi2,
                               double *ro, double *rd)
{
    const double *P0 = L + 4, *P1 = L + 7, *P2 = L + 10;
    double N[3];
    orc_normal_from_3pts(P0, P1, P2, N);
    double e21[3] = {P2[0] - P1[0], P2[1] - P1[1], P2[2] - P1[2]};
    double e10[3] = {P1[0] - P0[0], P1[1] - P0[1], P1[2] - P0[2]};
    double area = norm3(e21) * norm3(e10);
    double s[3];
    for (int i = 0; i < 3; i++)
    {
        s[i] = P0[i] + e10[i] * xi1;
        s[i] += e21[i] * xi2;
    }
    double g[3] = {ref[0] - s[0], ref[1] - s[1], ref[2] - s[2]}, dir[3];
    normalize3(g, dir);
    double c = dot3(dir, N);
    if (c <= 0)
    {
        ro[0] = ro[1] = ro[2] = 0;
        rd[0] = rd[1] = rd[2] = 0;
        return 0;
    }
    for (int i = 0; i < 3; i++)
        ro[i] = s[i], rd[i] = dir[i];
    double h[3] = {s[0] - ref[0], s[1] - ref[1], s[2] - ref[2]};
    double r = norm3(h);
    return r * r / (area * c) * (1.0 / (2 * M_PI));
}

double orc_sphere_light_sample(const double *L /*16*/, const double *ref, double xi1, double xi2,
                               double xi3, double *ro, double *rd)
{
    const double *C = L + 4;
    double R = L[7];
    double e[3] = {ref[0] - C[0], ref[1] - C[1], ref[2] - C[2]};
    double dc = norm3(e);
    double pdf;
    if (dc > R)
    {
        /* the cone sample itself is discarded by the reference (cLight.cpp:216 builds the ray from
         * the centre direction); only its pdf survives (cMathUtil.cpp:325) */
        double max_theta = asin(R / dc);
        double g[3] = {C[0] - ref[0], C[1] - ref[1], C[2] - ref[2]}, to_light[3];
        normalize3(g, to_light);
        pdf = 1.0 / (2 * M_PI * (1 - cos(max_theta)));
        for (int i = 0; i < 3; i++)
            ro[i] = ref[i], rd[i] = -to_light[i];
        (void)xi1, (void)xi2, (void)xi3;
    }
    else
    {
        double v[3] = {xi1 - 0.5, xi2 - 0.5, xi3 - 0.5}, u[3], lp[3], g[3];
        normalize3(v, u);
        pdf = 1.0 / (4 * M_PI);
        for (int i = 0; i < 3; i++)
            lp[i] = u[i] * R + C[i];
        for (int i = 0; i < 3; i++)
            g[i] = ref[i] - lp[i];
        normalize3(g, rd);
        for (int i = 0; i < 3; i++)
            ro[i] = ref[i];
    }
    return pdf;
}

/* ------------------------------------------------------------------------------------------ */
/* BxDFs (cBxDF.cpp:69-95, 98-111, 113-146, 156-227).  mat = Ka Kd Ks Ni Ns illum                */
/* ------------------------------------------------------------------------------------------ */
int orc_is_bsdf(const double *mat) { return !(fabs(mat[9] - 1.0) < 1e-5); } /* IsSame eps=1e-5 */

void orc_brdf_eval(const double *mat, const double *wi, const double *wo, const double *n,
                   double *f)
{
    double r[3];
    orc_reflect(n, wi, r);
    double ca = dot3(r, wo), ns = mat[10];
    double pw = pow(ca, ns);
    for (int i = 0; i < 3; i++)
        f[i] = mat[3 + i] / M_PI + mat[6 + i] * (ns + 2) / (2 * M_PI) * pw;
}

void orc_bsdf_eval(const double *mat, const double *wi, const double *wo, const double *n,
                   double *f)
{
    double ideal[3];
    orc_refract(n, wi, 1.0 / mat[9], ideal);
    if (has_nan3(ideal))
        orc_reflect(n, wi, ideal);
    double c = dot3(ideal, wo);
    if (c < 0)
        c = 0;
    double v = 1.0 / (4 * M_PI) * pow(c, 1);
    f[0] = f[1] = f[2] = v;
}

void orc_bxdf_eval(const double *mat, const double *wi, const double *wo, const double *n,
                   double *f)
{
    if (orc_is_bsdf(mat))
        orc_bsdf_eval(mat, wi, wo, n, f);
    else
        orc_brdf_eval(mat, wi, wo, n, f);
}

/* Sample_f: xi[0..1] are the draws in reference order; writes wi (pointing away from the surface)
 * and the indirect coefficient */
void orc_bxdf_sample(const double *mat, const double *n, const double *wo, const double *xi,
                     double *wi, double *coef)
{
    if (!orc_is_bsdf(mat))
    {
        double pdf = 1.0 / (2 * M_PI);
        orc_sample_hemisphere(n, xi[0], xi[1], wi);
        double mwi[3] = {-wi[0], -wi[1], -wi[2]}, f[3];
        orc_brdf_eval(mat, mwi, wo, n, f);
        double c = dot3(n, wi);
        for (int i = 0; i < 3; i++)
            coef[i] = f[i] * c / pdf;
    }
    else
    {
        double ni = mat[9];
        double R0 = pow((1.0 - ni) / (1.0 + ni), 2);
        double mwo[3] = {-wo[0], -wo[1], -wo[2]};
        double R = R0 + (1 - R0) * pow(1 - fabs(dot3(n, mwo)), 5);
        if (xi[0] < R)
            orc_reflect(n, mwo, wi);
        else
        {
            orc_refract(n, mwo, 1.0 / 1.5, wi);
            if (has_nan3(wi))
                orc_reflect(n, mwo, wi);
        }
        double pdf = 1.0 / (4 * M_PI);
        double v = 1.0 * pdf * 1.0;
        coef[0] = coef[1] = coef[2] = v / pdf;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cBxDF::Sample_Li (cBxDF.cpp:16-52): NEE over all lights                                       */
/* ------------------------------------------------------------------------------------------ */
static void nee(const orc_scene *sc, const double *mat, const double *n, const double *p,
                const double *ray_d, uint32_t pixel, uint32_t sample, uint32_t bounce,
                double *direct, orc_stats *st)
{
    direct[0] = direct[1] = direct[2] = 0;
    for (int32_t l = 0; l < sc->n_lights; l++)
    {
        const double *L = sc->lights + (size_t)l * 16;
        double ro[3], rd[3], pdf;
        double x1 = orc_rng(sc->seed, pixel, sample, bounce, 4 * l + 0);
        double x2 = orc_rng(sc->seed, pixel, sample, bounce, 4 * l + 1);
        if (L[0] == 0.0)
            pdf = orc_square_light_sample(L, p, x1, x2, ro, rd);
        else
        {
            double x3 = orc_rng(sc->seed, pixel, sample, bounce, 4 * l + 2);
            pdf = orc_sphere_light_sample(L, p, x1, x2, x3, ro, rd);
        }
        if (pdf < 1e-10)
            continue;
        if (st)
            st->shadow_rays++;
        if (!vistest_impl(sc, ro, p, st))
            continue;
        double mrd[3] = {-rd[0], -rd[1], -rd[2]};
        double c = dot3(n, mrd);
        if (c < 0)
            continue;
        double wo[3] = {-ray_d[0], -ray_d[1], -ray_d[2]}, f[3];
        orc_bxdf_eval(mat, rd, wo, n, f);
        for (int i = 0; i < 3; i++)
            direct[i] += (L[1 + i] * c / pdf) * f[i];
    }
}

/* ------------------------------------------------------------------------------------------ */
/* cPathTracer::GenerateRay (cPathTracer.cpp:129-213)                                            */
/* ------------------------------------------------------------------------------------------ */
static void mat4_mul(const double *a, const double *b, double *o)
{
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
        {
            r[i * 4 + j] = SUM4(a[i * 4 + 0] * b[0 * 4 + j], a[i * 4 + 1] * b[1 * 4 + j], a[i * 4 + 2] * b[2 * 4 + j],
                                a[i * 4 + 3] * b[3 * 4 + j]);
        }
    memcpy(o, r, sizeof(r));
}

/* general 4x4 inverse by cofactors (Eigen's Matrix4d::inverse() is restated, not copied) */
static void mat4_inverse(const double *m, double *inv)
{
    double a[16];
    a[0] = m[5] * m[10] * m[15] - m[5] * m[11] * m[14] - m[9] * m[6] * m[15] + m[9] * m[7] * m[14] +
           m[13] * m[6] * m[11] - m[13] * m[7] * m[10];
    a[4] = -m[4] * m[10] * m[15] + m[4] * m[11] * m[14] + m[8] * m[6] * m[15] - m[8] * m[7] * m[14] -
           m[12] * m[6] * m[11] + m[12] * m[7] * m[10];
    a[8] = m[4] * m[9] * m[15] - m[4] * m[11] * m[13] - m[8] * m[5] * m[15] + m[8] * m[7] * m[13] +
           m[12] * m[5] * m[11] - m[12] * m[7] * m[9];
    a[12] = -m[4] * m[9] * m[14] + m[4] * m[10] * m[13] + m[8] * m[5] * m[14] - m[8] * m[6] * m[13] -
            m[12] * m[5] * m[10] + m[12] * m[6] * m[9];
    a[1] = -m[1] * m[10] * m[15] + m[1] * m[11] * m[14] + m[9] * m[2] * m[15] - m[9] * m[3] * m[14] -
           m[13] * m[2] * m[11] + m[13] * m[3] * m[10];
    a[5] = m[0] * m[10] * m[15] - m[0] * m[11] * m[14] - m[8] * m[2] * m[15] + m[8] * m[3] * m[14] +
           m[12] * m[2] * m[11] - m[12] * m[3] * m[10];
    a[9] = -m[0] * m[9] * m[15] + m[0] * m[11] * m[13] + m[8] * m[1] * m[15] - m[8] * m[3] * m[13] -
           m[12] * m[1] * m[11] + m[12] * m[3] * m[9];
    a[13] = m[0] * m[9] * m[14] - m[0] * m[10] * m[13] - m[8] * m[1] * m[14] + m[8] * m[2] * m[13] +
            m[12] * m[1] * m[10] - m[12] * m[2] * m[9];
    a[2] = m[1] * m[6] * m[15] - m[1] * m[7] * m[14] - m[5] * m[2] * m[15] + m[5] * m[3] * m[14] +
           m[13] * m[2] * m[7] - m[13] * m[3] * m[6];
    a[6] = -m[0] * m[6] * m[15] + m[0] * m[7] * m[14] + m[4] * m[2] * m[15] - m[4] * m[3] * m[14] -
           m[12] * m[2] * m[7] + m[12] * m[3] * m[6];
    a[10] = m[0] * m[5] * m[15] - m[0] * m[7] * m[13] - m[4] * m[1] * m[15] + m[4] * m[3] * m[13] +
            m[12] * m[1] * m[7] - m[12] * m[3] * m[5];
    a[14] = -m[0] * m[5] * m[14] + m[0] * m[6] * m[13] + m[4] * m[1] * m[14] - m[4] * m[2] * m[13] -
            m[12] * m[1] * m[6] + m[12] * m[2] * m[5];
    a[3] = -m[1] * m[6] * m[11] + m[1] * m[7] * m[10] + m[5] * m[2] * m[11] - m[5] * m[3] * m[10] -
           m[9] * m[2] * m[7] + m[9] * m[3] * m[6];
    a[7] = m[0] * m[6] * m[11] - m[0] * m[7] * m[10] - m[4] * m[2] * m[11] + m[4] * m[3] * m[10] +
           m[8] * m[2] * m[7] - m[8] * m[3] * m[6];
    a[11] = -m[0] * m[5] * m[11] + m[0] * m[7] * m[9] + m[4] * m[1] * m[11] - m[4] * m[3] * m[9] -
            m[8] * m[1] * m[7] + m[8] * m[3] * m[5];
    a[15] = m[0] * m[5] * m[10] - m[0] * m[6] * m[9] - m[4] * m[1] * m[10] + m[4] * m[2] * m[9] +
            m[8] * m[1] * m[6] - m[8] * m[2] * m[5];
    double det = SUM4(m[0] * a[0], m[1] * a[4], m[2] * a[8], m[3] * a[12]);
    double id = 1.0 / det;
    for (int i = 0; i < 16; i++)
        inv[i] = a[i] * id;
}

/* the raster->world matrix  mat = V^-1 * mat3 * mat2 * mat1  (cPathTracer.cpp:148-182) */
void orc_camera_matrix(const orc_scene *sc, double *mat)
{
    double W = sc->width, H = sc->height;
    double m1[16] = {1.0 / W, 0, 0, 0.5 / W, 0, 1.0 / H, 0, 0.5 / H, 0, 0, 1, 0, 0, 0, 0, 1};
    double m2[16] = {2, 0, 0, -1, 0, -2, 0, 1, 0, 0, 1, 0, 0, 0, 0, 1};
    double tn = tan(sc->fov_deg / 180.0 * M_PI / 2);
    double m3[16] = {0};
    m3[0] = sc->width * 1.0 / sc->height * tn * sc->near_;
    m3[5] = tn * sc->near_;
    m3[10] = 0, m3[11] = -sc->near_;
    m3[15] = 1;
    double m4[16], t[16];
    mat4_inverse(sc->view, m4);
    mat4_mul(m4, m3, t);
    mat4_mul(t, m2, t);
    mat4_mul(t, m1, mat);
}

/* one primary ray: pixel (row y, col x) -> origin, direction */
static void primary_ray(const orc_scene *sc, const double *mat, int x, int y, double *o, double *d)
{
    double v[4] = {(double)x, (double)y, 1, 1}, pos[3];
    for (int i = 0; i < 3; i++)
    {
        pos[i] = SUM4(mat[i * 4 + 0] * v[0], mat[i * 4 + 1] * v[1], mat[i * 4 + 2] * v[2], mat[i * 4 + 3] * v[3]);
    }
    double g[3] = {pos[0] - sc->cam_pos[0], pos[1] - sc->cam_pos[1], pos[2] - sc->cam_pos[2]};
    normalize3(g, d);
    o[0] = sc->cam_pos[0], o[1] = sc->cam_pos[1], o[2] = sc->cam_pos[2];
}

/* rays_o / rays_d: [H*W][3], index y*W + x */
void orc_generate_rays(const orc_scene *sc, double *rays_o, double *rays_d)
{
    double mat[16];
    orc_camera_matrix(sc, mat);
    for (int y = 0; y < sc->height; y++)
        for (int x = 0; x < sc->width; x++)
        {
            size_t id = (size_t)y * sc->width + x;
            primary_ray(sc, mat, x, y, rays_o + id * 3, rays_d + id * 3);
        }
}

/* first-hit face ids of all camera rays (-1 = miss), optionally the hit points */
void orc_first_hit(const orc_scene *sc, int32_t *ids, double *pts)
{
    double mat[16];
    orc_camera_matrix(sc, mat);
#pragma omp parallel for schedule(dynamic)
    for (int y = 0; y < sc->height; y++)
        for (int x = 0; x < sc->width; x++)
        {
            size_t id = (size_t)y * sc->width + x;
            double o[3], d[3], p[3];
            primary_ray(sc, mat, x, y, o, d);
            ids[id] = closest(sc, o, d, p, NULL, NULL, NULL);
            if (pts)
                pts[id * 3] = p[0], pts[id * 3 + 1] = p[1], pts[id * 3 + 2] = p[2];
        }
}

/* ------------------------------------------------------------------------------------------ */
/* cPathTracer::RayTracePrimaryRay (cPathTracer.cpp:555-704): one pixel, all samples             */
/* sum_out receives the pre-final-clamp sum  (sum_s final_s / spp_total)                         */
/* ------------------------------------------------------------------------------------------ */
#define ORC_MAX_DEPTH 64
/* per-bounce diagnostics of one sample (tests only): the face hit (-1: miss) and the margin by
 * which it won its closest-hit search (runner-up distance minus its own; infinity: no runner-up) */
typedef struct
{
    int32_t n;
    int32_t face[ORC_MAX_DEPTH];
    double margin[ORC_MAX_DEPTH];
    orc_tie tie; /* which way the ties of this path fall (mask 0: the reference's own rule) */
    /* sensitivity probe: the direction sampled at bounce perturb_bounce (-1: none) is turned by
     * about perturb_eps radians before the path goes on */
    int32_t perturb_bounce;
    double perturb_eps;
} orc_probe;

/* turns the unit vector w by about eps radians (towards a fixed, generally non-parallel direction made
 * of its own components) and renormalises */
static void perturb_direction(double *w, double eps)
{
    double e[3] = {w[1], w[2], -w[0]};
    double k = dot3(e, w);
    for (int i = 0; i < 3; i++)
        e[i] -= k * w[i]; /* the part of e orthogonal to w */
    double en = norm3(e);
    if (!(en > 1e-3))
    {
        e[0] = w[2], e[1] = -w[0], e[2] = w[1];
        k = dot3(e, w);
        for (int i = 0; i < 3; i++)
            e[i] -= k * w[i];
        en = norm3(e);
    }
    double t[3];
    for (int i = 0; i < 3; i++)
        t[i] = w[i] + eps * e[i] / en;
    normalize3(t, w);
}

/* one iteration of the sample loop (cPathTracer.cpp:578-693): final = the sample's radiance after
 * the back-to-front combine, before the division by the sample count */
static void trace_sample(const orc_scene *sc, const double *o0, const double *d0, uint32_t pixel,
                         uint32_t sample, double *final, orc_stats *st, orc_probe *probe)
{
    int depth = sc->depth > ORC_MAX_DEPTH ? ORC_MAX_DEPTH : sc->depth;
    double o[3] = {o0[0], o0[1], o0[2]}, d[3] = {d0[0], d0[1], d0[2]};
    double D[ORC_MAX_DEPTH][3], C[ORC_MAX_DEPTH][3];
    int len = 0;
    final[0] = final[1] = final[2] = 0;
    if (st)
        st->paths++;
    if (probe)
        probe->n = 0;
    for (int bounce = 0; bounce < depth; bounce++)
    {
        double p[3], second = INFINITY;
        if (st)
            st->segments++;
        int32_t f = closest(sc, o, d, p, st, probe ? &second : NULL, probe ? &probe->tie : NULL);
        if (probe)
        {
            double e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
            probe->face[probe->n] = f;
            probe->margin[probe->n] = f < 0 ? INFINITY : second - norm3(e);
            probe->n++;
        }
        int32_t mid;
        if (f < 0 || (mid = sc->mat_ids[f]) == -1)
        {
            final[0] = final[1] = final[2] = 0;
            break;
        }
        const double *mat_ = sc->mats + (size_t)mid * 12;
        if (norm3(mat_) > 1e-6)
        {
            final[0] = mat_[0], final[1] = mat_[1], final[2] = mat_[2];
            break;
        }
        const double *v = sc->verts + (size_t)f * 9;
        double n[3];
        if (sc->enable_bary_normal)
        {
            double w[3], acc[3] = {0, 0, 0};
            const double *vn = sc->normals + (size_t)f * 9;
            orc_barycentric(p, v, v + 3, v + 6, w);
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++)
                    acc[k] += w[i] * vn[i * 3 + k];
            normalize3(acc, n);
        }
        else
            orc_normal_from_3pts(v, v + 3, v + 6, n);

        nee(sc, mat_, n, p, d, pixel, sample, (uint32_t)bounce, D[len], st);
        if (sc->enable_indirect)
        {
            double wo[3] = {-d[0], -d[1], -d[2]}, wi[3], xi[2];
            xi[0] = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, 4 * sc->n_lights + 0);
            xi[1] = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, 4 * sc->n_lights + 1);
            orc_bxdf_sample(mat_, n, wo, xi, wi, C[len]);
            if (probe && probe->perturb_bounce == bounce)
                perturb_direction(wi, probe->perturb_eps);
            len++;
            for (int i = 0; i < 3; i++)
                o[i] = p[i], d[i] = wi[i];
        }
        else
        {
            C[len][0] = C[len][1] = C[len][2] = 0;
            len++;
            break;
        }
    }
    for (int i = len - 1; i >= 0; i--)
    {
        for (int k = 0; k < 3; k++)
            final[k] = D[i][k] + C[i][k] * final[k];
        double m = final[0];
        if (final[1] > m)
            m = final[1];
        if (final[2] > m)
            m = final[2];
        if (0.0 > m)
            m = 0.0; /* the 4th component is 0 and takes part in maxCoeff */
        if (m > 1)
            for (int k = 0; k < 3; k++)
                final[k] /= m;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* The physically based estimator ("fast" mode).  It has NO counterpart in the reference, whose  */
/* estimator samples every light without a weight, draws pole-peaked "uniform" directions, never */
/* shadows sphere lights, has no Russian roulette and clamps every bounce (cBxDF.cpp:16-52,      */
/* cMathUtil.cpp:230-267, cLight.cpp:216, cPathTracer.cpp:589,685-688).  This one is a forward   */
/* path tracer over the same scene description:                                                  */
/*  - BRDF  f = Kd/pi + Ks (Ns+2)/(2 pi) max(0, r.wi)^Ns  (the reference's lobe, clamped),       */
/*    sampled by lobe: cosine hemisphere with probability pd = max Kd / (max Kd + max Ks), else   */
/*    the Phong lobe about the mirror direction; two-sided (the normal is turned to the viewer); */
/*  - glass (Ni != 1): Schlick reflect / refract with the material's own Ni, a delta lobe;       */
/*  - every conf light is sampled at every non-delta bounce (square: uniform area, one-sided;     */
/*    sphere: uniform cone, or the whole sphere from inside) with a visibility test to the        */
/*    sampled point -- sphere lights included -- and weighted by the balance heuristic against    */
/*    the BRDF's density of that direction;                                                       */
/*  - the conf lights are shapes that emit and do not occlude: a BRDF-sampled ray that crosses    */
/*    one before its next surface collects Li with the complementary weight (weight 1 after a     */
/*    delta bounce); camera rays and delta chains from the camera do not see them;                */
/*  - faces of an emissive material (Ka) end the path; their Ka counts only when the camera sees  */
/*    them, directly or through delta bounces (they are the visible stand-ins of the lights: the  */
/*    scene format keeps the two apart, so nothing else can avoid counting the light twice);      */
/*    (at the last bounce the depth allows, the light sample counts whole);                       */
/*  - Russian roulette from bounce 3 on: survival q = min(0.95, max throughput);                  */
/*  - no clamp before the final image.                                                            */
/* Random numbers: light l dimensions 4l, 4l+1; lobe / Fresnel choice 4 n_lights, direction      */
/* 4 n_lights + 1, + 2, roulette 4 n_lights + 3.                                                  */
/* ------------------------------------------------------------------------------------------ */
static inline double max3(const double *a)
{
    double m = a[0] > a[1] ? a[0] : a[1];
    return m > a[2] ? m : a[2];
}

/* density (solid angle, seen from o along unit d) with which light L is sampled towards the
 * point where the ray (o, d) crosses it, and the distance to that point; 0 when the ray misses
 * the light or meets a square light from behind */
double orc_fast_light_pdf(const double *L, const double *o, const double *d, double *t_out)
{
    *t_out = INFINITY;
    if (L[0] == 0.0)
    {
        const double *P0 = L + 4, *P1 = L + 7, *P2 = L + 10;
        double N[3], e10[3], e21[3];
        orc_normal_from_3pts(P0, P1, P2, N);
        for (int i = 0; i < 3; i++)
            e10[i] = P1[i] - P0[i], e21[i] = P2[i] - P1[i];
        double dn = dot3(d, N);
        if (!(dn < 0))
            return 0; /* the light emits on the side its normal points to */
        double g[3] = {P0[0] - o[0], P0[1] - o[1], P0[2] - o[2]};
        double t = dot3(g, N) / dn;
        if (!(t > 1e-6))
            return 0;
        double h[3] = {o[0] + t * d[0] - P0[0], o[1] + t * d[1] - P0[1], o[2] + t * d[2] - P0[2]};
        double u = dot3(h, e10) / dot3(e10, e10), v = dot3(h, e21) / dot3(e21, e21);
        if (u < 0 || u > 1 || v < 0 || v > 1)
            return 0;
        *t_out = t;
        return t * t / (norm3(e21) * norm3(e10) * -dn);
    }
    const double *C = L + 4;
    double R = L[7];
    double g[3] = {C[0] - o[0], C[1] - o[1], C[2] - o[2]};
    double dc2 = dot3(g, g), b = dot3(g, d), disc = b * b - dc2 + R * R;
    if (!(disc >= 0))
        return 0;
    double sq = sqrt(disc);
    if (dc2 > R * R)
    {
        double t = b - sq;
        if (!(t > 1e-6))
            return 0;
        *t_out = t;
        return 1.0 / (2 * M_PI * (1 - sqrt(1 - R * R / dc2)));
    }
    *t_out = b + sq;
    return 1.0 / (4 * M_PI);
}

/* one light sample from the shaded point p: direction wi (unit, towards the light), the point q
 * on the light, the density of wi; returns 0 when the sample is of no use */
double orc_fast_light_sample(const double *L, const double *p, double xi1, double xi2, double *wi, double *q)
{
    if (L[0] == 0.0)
    {
        const double *P0 = L + 4, *P1 = L + 7, *P2 = L + 10;
        double N[3], e10[3], e21[3];
        orc_normal_from_3pts(P0, P1, P2, N);
        for (int i = 0; i < 3; i++)
            e10[i] = P1[i] - P0[i], e21[i] = P2[i] - P1[i];
        for (int i = 0; i < 3; i++)
        {
            q[i] = P0[i] + e10[i] * xi1;
            q[i] += e21[i] * xi2;
        }
        double w[3] = {q[0] - p[0], q[1] - p[1], q[2] - p[2]};
        double d2 = dot3(w, w);
        if (!(d2 > 0))
            return 0;
        double dl = sqrt(d2);
        for (int i = 0; i < 3; i++)
            wi[i] = w[i] / dl;
        double cl = -dot3(wi, N);
        if (!(cl > 0))
            return 0;
        return d2 / (norm3(e21) * norm3(e10) * cl);
    }
    const double *C = L + 4;
    double R = L[7];
    double g[3] = {C[0] - p[0], C[1] - p[1], C[2] - p[2]};
    double dc2 = dot3(g, g), dc = sqrt(dc2);
    if (dc2 > R * R)
    {
        double axis[3] = {g[0] / dc, g[1] / dc, g[2] / dc};
        double s2max = R * R / dc2, cmax = sqrt(1 - s2max);
        double ct = 1 - xi1 * (1 - cmax), st2 = 1 - ct * ct, st = sqrt(st2 > 0 ? st2 : 0), phi = 2 * M_PI * xi2;
        double loc[3] = {st * cos(phi), ct, -st * sin(phi)};
        rot_from_y(axis, loc, wi);
        double disc = R * R - dc2 * st2;
        double t = dc * ct - sqrt(disc > 0 ? disc : 0);
        for (int i = 0; i < 3; i++)
            q[i] = p[i] + t * wi[i];
        return 1.0 / (2 * M_PI * (1 - cmax));
    }
    /* inside the sphere: any direction reaches it */
    double ct = 1 - 2 * xi1, st = sqrt(1 - ct * ct > 0 ? 1 - ct * ct : 0), phi = 2 * M_PI * xi2;
    wi[0] = st * cos(phi), wi[1] = ct, wi[2] = -st * sin(phi);
    double b = dot3(g, wi), t = b + sqrt(b * b - dc2 + R * R);
    for (int i = 0; i < 3; i++)
        q[i] = p[i] + t * wi[i];
    return 1.0 / (4 * M_PI);
}

/* BRDF value and sampling density for outward unit directions wi (to the light) and wo (to the
 * viewer) about the normal n (n.wo >= 0) */
void orc_fast_brdf(const double *mat, const double *n, const double *wo, const double *wi, double *f, double *pdf)
{
    double cn = dot3(n, wi);
    f[0] = f[1] = f[2] = 0, *pdf = 0;
    if (!(cn > 0))
        return;
    double k = 2 * dot3(n, wo), r[3] = {k * n[0] - wo[0], k * n[1] - wo[1], k * n[2] - wo[2]};
    double ca = dot3(r, wi), ns = mat[10];
    double pw = ca > 0 ? pow(ca, ns) : 0;
    double kd = max3(mat + 3), ks = max3(mat + 6);
    for (int i = 0; i < 3; i++)
        f[i] = mat[3 + i] / M_PI + mat[6 + i] * ((ns + 2) / (2 * M_PI)) * pw;
    if (kd + ks > 0)
    {
        double pd = kd / (kd + ks);
        *pdf = pd * cn / M_PI + (1 - pd) * ((ns + 1) / (2 * M_PI)) * pw;
    }
}

/* next direction of a BRDF surface from three draws; returns 0 when the path ends here */
int orc_fast_brdf_sample(const double *mat, const double *n, const double *wo, const double *xi, double *wi)
{
    double kd = max3(mat + 3), ks = max3(mat + 6), ns = mat[10];
    if (!(kd + ks > 0))
        return 0;
    double pd = kd / (kd + ks), phi = 2 * M_PI * xi[2];
    if (xi[0] < pd)
    {
        double r = sqrt(xi[1]), h = sqrt(1 - xi[1]);
        double loc[3] = {r * cos(phi), h, -r * sin(phi)};
        rot_from_y(n, loc, wi);
    }
    else
    {
        double ca = pow(xi[1], 1.0 / (ns + 1)), sa = sqrt(1 - ca * ca > 0 ? 1 - ca * ca : 0);
        double k = 2 * dot3(n, wo), r[3] = {k * n[0] - wo[0], k * n[1] - wo[1], k * n[2] - wo[2]};
        double loc[3] = {sa * cos(phi), ca, -sa * sin(phi)};
        rot_from_y(r, loc, wi);
    }
    return dot3(n, wi) > 0;
}

/* glass: reflect or refract, chosen by Schlick's reflectance against one draw */
void orc_fast_glass_sample(const double *mat, const double *n_, const double *wo, double xi, double *wi)
{
    double ni = mat[9], n[3] = {n_[0], n_[1], n_[2]};
    double ci = dot3(n, wo), eta = 1.0 / ni;
    if (ci < 0)
    {
        n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
        ci = -ci, eta = ni;
    }
    double s2t = eta * eta * (1 - ci * ci);
    double r0 = (1 - ni) / (1 + ni);
    r0 *= r0;
    int reflect = 1;
    double ct = 0;
    if (s2t < 1)
    {
        ct = sqrt(1 - s2t);
        double c = eta > 1 ? ct : ci, x = 1 - c, x2 = x * x;
        double Rf = r0 + (1 - r0) * (x2 * x2 * x);
        reflect = xi < Rf;
    }
    for (int i = 0; i < 3; i++)
        wi[i] = reflect ? 2 * ci * n[i] - wo[i] : (eta * ci - ct) * n[i] - eta * wo[i];
}

static void fast_sample(const orc_scene *sc, const double *o0, const double *d0, uint32_t pixel, uint32_t sample,
                        double *Lout, orc_stats *st, orc_probe *probe)
{
    int depth = sc->depth > ORC_MAX_DEPTH ? ORC_MAX_DEPTH : sc->depth;
    double o[3] = {o0[0], o0[1], o0[2]}, d[3] = {d0[0], d0[1], d0[2]};
    double T[3] = {1, 1, 1}, Lsum[3] = {0, 0, 0};
    double pdf_prev = 0; /* density of the direction this segment was sampled with; < 0: a delta lobe */
    int eye_chain = 1;   /* only delta bounces since the camera */
    const int D0 = 4 * sc->n_lights;
    if (st)
        st->paths++;
    if (probe)
        probe->n = 0;
    for (int bounce = 0; bounce < depth; bounce++)
    {
        double p[3], second = INFINITY;
        if (st)
            st->segments++;
        int32_t f = closest(sc, o, d, p, st, probe ? &second : NULL, probe ? &probe->tie : NULL);
        double t_geom = INFINITY;
        if (f >= 0)
        {
            double e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
            t_geom = norm3(e);
        }
        if (probe)
        {
            probe->face[probe->n] = f;
            probe->margin[probe->n] = f < 0 ? INFINITY : second - t_geom;
            probe->n++;
        }
        /* conf lights crossed by a sampled direction before the surface: emitters that do not occlude */
        if (bounce > 0 && !eye_chain)
            for (int32_t l = 0; l < sc->n_lights; l++)
            {
                const double *L = sc->lights + (size_t)l * 16;
                double tl, pl = orc_fast_light_pdf(L, o, d, &tl);
                if (pl > 0 && tl < t_geom)
                {
                    double w = pdf_prev < 0 ? 1.0 : pdf_prev / (pdf_prev + pl);
                    if (pdf_prev >= 0 && sc->fast_strategy)
                        w = sc->fast_strategy == 2 ? 1.0 : 0.0;
                    for (int k = 0; k < 3; k++)
                        Lsum[k] += T[k] * L[1 + k] * w;
                }
            }
        int32_t mid;
        if (f < 0 || (mid = sc->mat_ids[f]) == -1)
            break;
        const double *mat_ = sc->mats + (size_t)mid * 12;
        if (norm3(mat_) > 1e-6)
        {
            if (eye_chain)
                for (int k = 0; k < 3; k++)
                    Lsum[k] += T[k] * mat_[k];
            break;
        }
        const double *v = sc->verts + (size_t)f * 9;
        double n[3];
        if (sc->enable_bary_normal)
        {
            double w[3], acc[3] = {0, 0, 0};
            const double *vn = sc->normals + (size_t)f * 9;
            orc_barycentric(p, v, v + 3, v + 6, w);
            for (int i = 0; i < 3; i++)
                for (int k = 0; k < 3; k++)
                    acc[k] += w[i] * vn[i * 3 + k];
            normalize3(acc, n);
        }
        else
            orc_normal_from_3pts(v, v + 3, v + 6, n);
        double wo[3] = {-d[0], -d[1], -d[2]}, wi[3];
        if (orc_is_bsdf(mat_))
        {
            if (!sc->enable_indirect)
                break;
            double xi = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, D0);
            orc_fast_glass_sample(mat_, n, wo, xi, wi);
            pdf_prev = -1;
        }
        else
        {
            if (dot3(n, wo) < 0)
                n[0] = -n[0], n[1] = -n[1], n[2] = -n[2];
            for (int32_t l = 0; l < sc->n_lights; l++)
            {
                const double *L = sc->lights + (size_t)l * 16;
                double x1 = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, 4 * l + 0);
                double x2 = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, 4 * l + 1);
                double wl[3], q[3], fl[3], pb;
                double pl = orc_fast_light_sample(L, p, x1, x2, wl, q);
                if (!(pl > 0))
                    continue;
                orc_fast_brdf(mat_, n, wo, wl, fl, &pb);
                if (!(dot3(n, wl) > 0))
                    continue;
                if (st)
                    st->shadow_rays++;
                if (!vistest_impl(sc, q, p, st))
                    continue;
                /* cos / pl times the balance weight pl / (pl + pb); at the last bounce no sampled
                 * direction follows that could collect the other share: the light sample counts whole */
                int last = !sc->enable_indirect || bounce + 1 >= depth;
                double w = last ? dot3(n, wl) / pl : dot3(n, wl) / (pl + pb);
                if (sc->fast_strategy && !last)
                    w = sc->fast_strategy == 1 ? dot3(n, wl) / pl : 0.0;
                for (int k = 0; k < 3; k++)
                    Lsum[k] += T[k] * L[1 + k] * fl[k] * w;
            }
            if (!sc->enable_indirect)
                break;
            double xi[3];
            for (int k = 0; k < 3; k++)
                xi[k] = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, D0 + k);
            if (!orc_fast_brdf_sample(mat_, n, wo, xi, wi))
                break;
            double fb[3], pb;
            orc_fast_brdf(mat_, n, wo, wi, fb, &pb);
            if (!(pb > 0))
                break;
            double c = dot3(n, wi) / pb;
            for (int k = 0; k < 3; k++)
                T[k] *= fb[k] * c;
            pdf_prev = pb;
            eye_chain = 0;
        }
        if (bounce >= 3)
        {
            double q = max3(T);
            if (q > 0.95)
                q = 0.95;
            double xr = orc_rng(sc->seed, pixel, sample, (uint32_t)bounce, D0 + 3);
            if (!(xr < q))
                break;
            for (int k = 0; k < 3; k++)
                T[k] /= q;
        }
        for (int i = 0; i < 3; i++)
            o[i] = p[i], d[i] = wi[i];
    }
    Lout[0] = Lsum[0], Lout[1] = Lsum[1], Lout[2] = Lsum[2];
}

static void trace_pixel(const orc_scene *sc, const double *mat, int x, int y, double *sum_out,
                        orc_stats *st)
{
    uint32_t pixel = (uint32_t)(y * sc->width + x);
    double o0[3], d0[3];
    primary_ray(sc, mat, x, y, o0, d0);
    double color[3] = {0, 0, 0};
    for (int s = 0; s < sc->spp; s++)
    {
        double final[3];
        if (sc->estimator == 1)
            fast_sample(sc, o0, d0, pixel, (uint32_t)(sc->sample_offset + s), final, st, NULL);
        else
            trace_sample(sc, o0, d0, pixel, (uint32_t)(sc->sample_offset + s), final, st, NULL);
        for (int k = 0; k < 3; k++)
            color[k] += final[k] / sc->spp_total;
    }
    sum_out[0] = color[0], sum_out[1] = color[1], sum_out[2] = color[2];
}

/* One sample of one pixel with its diagnostics (tests only): rgb = the sample's radiance (not yet
 * divided by the sample count); faces / margins: [depth] per-bounce face hit and winning margin.
 * Closest-hit searches whose winner and runner-up lie within `gap` are the path's ties, numbered as
 * they are met; bit k of tie_mask makes tie k fall to the runner-up; *n_ties receives how many the
 * path (as traced under this mask) met.  Returns the number of bounces traced. */
int32_t orc_trace_sample_ties(const orc_scene *sc, int x, int y, int sample, uint32_t tie_mask, double gap,
                              double *rgb, int32_t *faces, double *margins, int32_t *n_ties)
{
    double mat[16], o0[3], d0[3];
    orc_probe pr;
    pr.tie.mask = tie_mask, pr.tie.n = 0, pr.tie.gap = gap;
    pr.perturb_bounce = -1, pr.perturb_eps = 0;
    orc_camera_matrix(sc, mat);
    primary_ray(sc, mat, x, y, o0, d0);
    if (sc->estimator == 1)
        fast_sample(sc, o0, d0, (uint32_t)(y * sc->width + x), (uint32_t)sample, rgb, NULL, &pr);
    else
        trace_sample(sc, o0, d0, (uint32_t)(y * sc->width + x), (uint32_t)sample, rgb, NULL, &pr);
    for (int i = 0; i < pr.n; i++)
    {
        if (faces)
            faces[i] = pr.face[i];
        if (margins)
            margins[i] = pr.margin[i];
    }
    if (n_ties)
        *n_ties = pr.tie.n;
    return pr.n;
}

/* Sensitivity of one sample to its own roundings (tests only; the reference's estimator): the sample
 * traced with the direction it draws at `bounce` turned by about eps radians.  The difference to the
 * unperturbed value, divided by eps, says how strongly the rest of the path amplifies an error of that
 * direction -- curved, smoothly shaded geometry multiplies it at every further bounce.  Returns the
 * number of bounces traced (the perturbation has no effect when the path ends before `bounce`). */
int32_t orc_trace_sample_perturbed(const orc_scene *sc, int x, int y, int sample, int bounce, double eps, double *rgb)
{
    double mat[16], o0[3], d0[3];
    orc_probe pr;
    pr.tie.mask = 0u, pr.tie.n = 0, pr.tie.gap = -1.0;
    pr.perturb_bounce = bounce, pr.perturb_eps = eps;
    orc_camera_matrix(sc, mat);
    primary_ray(sc, mat, x, y, o0, d0);
    if (sc->estimator == 1)
        return -1;
    trace_sample(sc, o0, d0, (uint32_t)(y * sc->width + x), (uint32_t)sample, rgb, NULL, &pr);
    return pr.n;
}

/* the same with every tie falling the reference's way */
int32_t orc_trace_sample(const orc_scene *sc, int x, int y, int sample, double *rgb, int32_t *faces,
                         double *margins)
{
    return orc_trace_sample_ties(sc, x, y, sample, 0u, -1.0, rgb, faces, margins, NULL);
}

/* the final per-pixel clamp (cPathTracer.cpp:701-702) */
void orc_finalize(const double *sum, double *out, int64_t n_pixels)
{
    for (int64_t i = 0; i < n_pixels; i++)
    {
        double r = sum[i * 3], g = sum[i * 3 + 1], b = sum[i * 3 + 2];
        double m = r > g ? r : g;
        m = m > b ? m : b;
        m = m > 0 ? m : 0;
        if (m > 1)
            r /= m, g /= m, b /= m;
        out[i * 3] = r, out[i * 3 + 1] = g, out[i * 3 + 2] = b;
    }
}

/* cPathTracer::RayTracing (cPathTracer.cpp:517-553): OpenMP dynamic over rows, DrawRegion filter.
 * Pixels outside the region are written as 0 (the reference leaves them uninitialised).
 * sum: [H*W][3] pre-final-clamp sums.  threads <= 0 keeps the OpenMP default. */
void orc_render(const orc_scene *sc, double *sum, orc_stats *stats, int threads)
{
    double mat[16];
    orc_camera_matrix(sc, mat);
    orc_stats total;
    memset(&total, 0, sizeof(total));
    memset(sum, 0, sizeof(double) * 3 * (size_t)sc->width * sc->height);
#ifdef _OPENMP
    if (threads > 0)
        omp_set_num_threads(threads);
#endif
#pragma omp parallel
    {
        orc_stats local;
        memset(&local, 0, sizeof(local));
#pragma omp for schedule(dynamic)
        for (int row = 0; row < sc->height; row++)
        {
            if (row < sc->region[0] || row > sc->region[2])
                continue;
            for (int col = 0; col < sc->width; col++)
            {
                if (col < sc->region[1] || col > sc->region[3])
                    continue;
                trace_pixel(sc, mat, col, row, sum + ((size_t)row * sc->width + col) * 3,
                            stats ? &local : NULL);
            }
        }
#pragma omp critical
        {
            total.paths += local.paths;
            total.segments += local.segments;
            total.closest_rays += local.closest_rays;
            total.shadow_rays += local.shadow_rays;
            total.tri_tests += local.tri_tests;
        }
    }
    if (stats)
        *stats = total;
}

int orc_num_threads(void)
{
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
