/*
 * ref_accels.c -- CPU restatement of the reference's two culling structures (TEST INFRASTRUCTURE
 * ONLY: loaded by tests/ and tools/, never by the product).
 *
 * The reference answers RayCast / VisTest through one of three structures: the plain loop over all
 * faces ("None", restated in pt_oracle.c and used as ground truth), a uniform grid of boxes
 * ("AABB") and an octree ("OCTREE").  The latter two only decide WHICH faces get the exact
 * ray/triangle test; this file restates them, quirks included, so that the faces they drop can be
 * counted against the plain loop:
 *   - the ray/box slab test runs in float and the segment/box test is two of them
 *     (src/geometry/cBaseMesh.cpp:60-99), with tRay::Init's patched reciprocal
 *     (src/geometry/cBaseMesh.h:154-165);
 *   - faces are binned with a 13-axis separating-axis test whose nine edge axes treat touching
 *     as separated (cBaseMesh.cpp:117-223);
 *   - the mesh's upper bound starts from the smallest POSITIVE double (cBaseMesh.cpp:235);
 *   - tOctreeNode::split pushes a face into a child without asking whether that child has itself
 *     split already (cAccelStruct.cpp:335-353); faces pushed that way would sit in an inner node
 *     where no query looks.  A node splits exactly when it holds `capacity` faces, so a child can
 *     only fill up on its parent's last face and the case cannot arise; `stranded` counts such
 *     references all the same, and is expected to stay 0;
 *   - a split never ends when `capacity` faces all meet one ever smaller box (that many faces
 *     round one vertex): the reference overflows its stack, here the build gives up at depth 48;
 *   - both structures visit boxes in storage order and keep the FIRST of equally distant hits,
 *     the plain loop keeps the lowest face id;
 *   - the octree is cached beside the mesh as text (`<mesh path>.octree`, cAccelStruct.cpp:485-611):
 *     one line per node in breadth-first order with the box bounds printed at the stream's default
 *     six significant digits, so an octree READ from its cache culls with boxes that differ from
 *     the ones it was built with (orc_octree_write / orc_octree_read restate both directions; the
 *     reference ships no .octree file, so the format is pinned by its writer's source alone).
 * PARITY STATUS: unpinned (the reference holds no vectors for these structures and cannot be built
 * here); Eigen's reductions are restated with left-to-right sums.
 */
#include <float.h>
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int orc_tri_raycast(const double *o, const double *d, const double *p1, const double *p2, const double *p3, double *pt);
double orc_tri_raycast_t(const double *o, const double *d, const double *p1, const double *p2, const double *p3);

typedef struct
{
    double b[3][2]; /* tAABB::bound */
    int32_t *faces; /* tAABB::mFaceId */
    int32_t n, cap;
} box_t;

typedef struct node_s
{
    struct node_s *child[8]; /* Children[x][y][z] at x*4 + y*2 + z */
    box_t box;
    int leaf;
    int32_t id; /* tOctreeNode::id: nodes are numbered as they are created */
} node_t;

typedef struct
{
    int type; /* 1 grid, 2 octree */
    const double *verts;
    int32_t n_tris;
    box_t *boxes; /* grid: non-empty boxes in id[0], id[1], id[2] order */
    int32_t n_boxes;
    node_t *root;
    node_t **all; /* every node ever created, for freeing */
    int32_t n_all, cap_all;
    int32_t capacity;
    int failed; /* the split recursion did not end (the reference would overflow its stack) */
    int64_t stranded; /* face references left in inner nodes or dropped with a replaced subtree */
    int64_t exits;    /* faces that met no box (grid) or no child at a split (octree): the reference
                         prints an error and calls exit(1) there; here the face is left out */
} accel_t;

static void box_push(box_t *b, int32_t f)
{
    if (b->n == b->cap)
    {
        b->cap = b->cap ? 2 * b->cap : 8;
        b->faces = (int32_t *)realloc(b->faces, sizeof(int32_t) * (size_t)b->cap);
    }
    b->faces[b->n++] = f;
}

/* std::min / std::max on floats, NaN behaviour included */
static inline float fmin_std(float a, float b) { return (b < a) ? b : a; }
static inline float fmax_std(float a, float b) { return (a < b) ? b : a; }

/* tRay::Init (cBaseMesh.h:154-165): reciprocal direction, infinities replaced by
 * 1e10 * sign(dir[0]) whatever the axis */
static void inv_dir(const double *d, double *inv)
{
    for (int i = 0; i < 3; i++)
    {
        inv[i] = 1.0 / d[i];
        if (isinf(inv[i]))
            inv[i] = 1e10 * (d[0] > 0 ? 1 : -1);
    }
}

/* tAABB::intersect(const tRay&) (cBaseMesh.cpp:60-92): products in double, rounded to float,
 * compared in float */
static int box_ray(const box_t *bx, const double *o, const double *inv)
{
    float t1 = (float)((bx->b[0][0] - o[0]) * inv[0]);
    float t2 = (float)((bx->b[0][1] - o[0]) * inv[0]);
    float t3 = (float)((bx->b[1][0] - o[1]) * inv[1]);
    float t4 = (float)((bx->b[1][1] - o[1]) * inv[1]);
    float t5 = (float)((bx->b[2][0] - o[2]) * inv[2]);
    float t6 = (float)((bx->b[2][1] - o[2]) * inv[2]);
    float tmin = fmax_std(fmax_std(fmin_std(t1, t2), fmin_std(t3, t4)), fmin_std(t5, t6));
    float tmax = fmin_std(fmin_std(fmax_std(t1, t2), fmax_std(t3, t4)), fmax_std(t5, t6));
    if (tmax < 0)
        return 0;
    if (tmin > tmax)
        return 0;
    return 1;
}

/* tAABB::intersect(const tLine&) (cBaseMesh.cpp:94-99): the ray from either end towards the other */
static int box_segment(const box_t *bx, const double *p1, const double *p2)
{
    double d1[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    double d2[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double i1[3], i2[3];
    inv_dir(d1, i1);
    inv_dir(d2, i2);
    return box_ray(bx, p1, i1) && box_ray(bx, p2, i2);
}

static void project(const double (*pts)[3], int n, const double *axis, double *mn, double *mx)
{
    *mn = DBL_MAX, *mx = -DBL_MAX;
    for (int i = 0; i < n; i++)
    {
        double v = axis[0] * pts[i][0] + axis[1] * pts[i][1] + axis[2] * pts[i][2];
        if (v < *mn)
            *mn = v;
        if (v > *mx)
            *mx = v;
    }
}

static void cross(const double *a, const double *b, double *o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}

/* Eigen normalize(): divides by the norm when the squared norm is positive */
static void normalize(double *v)
{
    double z = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    if (z > 0)
    {
        double n = sqrt(z);
        v[0] /= n, v[1] /= n, v[2] /= n;
    }
}

/* tAABB::intersect(const tFace*) (cBaseMesh.cpp:117-223) */
static int box_face(const box_t *bx, const double *tri /*[3][3]*/)
{
    double tv[3][3], bv[8][3];
    memcpy(tv, tri, sizeof(tv));
    for (int x = 0; x < 2; x++)
        for (int y = 0; y < 2; y++)
            for (int z = 0; z < 2; z++)
            {
                bv[x * 4 + y * 2 + z][0] = bx->b[0][x];
                bv[x * 4 + y * 2 + z][1] = bx->b[1][y];
                bv[x * 4 + y * 2 + z][2] = bx->b[2][z];
            }
    static const double unit[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    double tmn, tmx, bmn, bmx;
    for (int i = 0; i < 3; i++)
    {
        project(tv, 3, unit[i], &tmn, &tmx);
        if (tmx < bx->b[i][0] || tmn > bx->b[i][1])
            return 0;
    }
    double v1[3], v2[3], nrm[3];
    for (int k = 0; k < 3; k++)
        v1[k] = tv[1][k] - tv[0][k], v2[k] = tv[2][k] - tv[1][k];
    cross(v1, v2, nrm);
    normalize(nrm);
    double off = nrm[0] * tv[0][0] + nrm[1] * tv[0][1] + nrm[2] * tv[0][2];
    project(bv, 8, nrm, &bmn, &bmx);
    if (bmx < off || bmn > off)
        return 0;
    double edge[3][3];
    for (int k = 0; k < 3; k++)
    {
        edge[0][k] = tv[0][k] - tv[1][k];
        edge[1][k] = tv[1][k] - tv[2][k];
        edge[2][k] = tv[2][k] - tv[0][k];
    }
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++)
        {
            double ax[3];
            cross(edge[i], unit[j], ax);
            normalize(ax);
            double len = sqrt(ax[0] * ax[0] + ax[1] * ax[1] + ax[2] * ax[2]);
            if (fabs(len - 1) > 1e-10)
                continue; /* the edge is parallel to this box axis */
            project(bv, 8, ax, &bmn, &bmx);
            project(tv, 3, ax, &tmn, &tmx);
            if (bmx <= tmn || bmn >= tmx)
                return 0;
        }
    return 1;
}

/* cBaseMesh bounds (cBaseMesh.cpp:235-237, 295-301) widened by 1e-5 (cAccelStruct.cpp:101-104) */
static void scene_bounds(const double *verts, int32_t n_tris, double *lower, double *upper)
{
    for (int i = 0; i < 3; i++)
        upper[i] = DBL_MIN, lower[i] = DBL_MAX; /* sic: the smallest positive double */
    for (int64_t v = 0; v < (int64_t)n_tris * 3; v++)
        for (int i = 0; i < 3; i++)
        {
            double p = verts[v * 3 + i];
            if (p > upper[i])
                upper[i] = p;
            if (p < lower[i])
                lower[i] = p;
        }
    for (int i = 0; i < 3; i++)
        upper[i] += 1e-5, lower[i] -= 1e-5;
}

/* ---- octree (cAccelStruct.cpp:266-482) ---------------------------------------------------- */
static node_t *node_new(accel_t *A, double b[3][2])
{
    node_t *n = (node_t *)calloc(1, sizeof(node_t));
    memcpy(n->box.b, b, sizeof(n->box.b));
    n->leaf = 1;
    n->id = A->n_all;
    if (A->n_all == A->cap_all)
    {
        A->cap_all = A->cap_all ? 2 * A->cap_all : 64;
        A->all = (node_t **)realloc(A->all, sizeof(node_t *) * (size_t)A->cap_all);
    }
    A->all[A->n_all++] = n;
    return n;
}

static int64_t subtree_refs(const node_t *n)
{
    if (!n)
        return 0;
    int64_t s = n->box.n;
    for (int i = 0; i < 8; i++)
        s += subtree_refs(n->child[i]);
    return s;
}

#define SPLIT_DEPTH_LIMIT 48

/* tOctreeNode::split (cAccelStruct.cpp:291-375) */
static void node_split(accel_t *A, node_t *nd, int depth)
{
    if (depth > SPLIT_DEPTH_LIMIT)
    {
        A->failed = 1;
        return;
    }
    nd->leaf = 0;
    for (int i = 0; i < 8; i++) /* a second split replaces the children; what they held is gone */
        A->stranded += subtree_refs(nd->child[i]);
    double w[3], st[3];
    for (int i = 0; i < 3; i++)
        w[i] = nd->box.b[i][1] - nd->box.b[i][0], st[i] = nd->box.b[i][0];
    for (int x = 0; x < 2; x++)
        for (int y = 0; y < 2; y++)
            for (int z = 0; z < 2; z++)
            {
                double b[3][2];
                b[0][0] = st[0] + x / 2.0 * w[0], b[0][1] = st[0] + (x + 1) / 2.0 * w[0];
                b[1][0] = st[1] + y / 2.0 * w[1], b[1][1] = st[1] + (y + 1) / 2.0 * w[1];
                b[2][0] = st[2] + z / 2.0 * w[2], b[2][1] = st[2] + (z + 1) / 2.0 * w[2];
                nd->child[x * 4 + y * 2 + z] = node_new(A, b);
            }
    for (int32_t k = 0; k < nd->box.n && !A->failed; k++)
    {
        const int32_t f = nd->box.faces[k];
        int placed = 0;
        for (int c = 0; c < 8; c++)
        {
            node_t *ch = nd->child[c];
            if (box_face(&ch->box, A->verts + (size_t)f * 9))
            {
                placed = 1;
                box_push(&ch->box, f); /* leaf or not */
                if (ch->box.n == A->capacity)
                    node_split(A, ch, depth + 1);
            }
        }
        if (!placed) /* e.g. an axis-aligned face lying in the split plane: touching counts as apart */
            A->exits++;
    }
    nd->box.n = 0;
}

/* cOctree::AddFace (cAccelStruct.cpp:426-482) */
static void add_face(accel_t *A, node_t *nd, int32_t f, int depth)
{
    if (A->failed || !box_face(&nd->box, A->verts + (size_t)f * 9))
        return;
    if (nd->leaf)
    {
        box_push(&nd->box, f);
        if (nd->box.n == A->capacity)
            node_split(A, nd, depth);
    }
    else
        for (int c = 0; c < 8; c++)
            add_face(A, nd->child[c], f, depth + 1);
}

static int64_t inner_refs(const node_t *n)
{
    if (!n || n->leaf)
        return 0;
    int64_t s = n->box.n;
    for (int i = 0; i < 8; i++)
        s += inner_refs(n->child[i]);
    return s;
}

/* ---- public ------------------------------------------------------------------------------- */
void orc_accel_free(accel_t *A)
{
    if (!A)
        return;
    for (int32_t i = 0; i < A->n_boxes; i++)
        free(A->boxes[i].faces);
    free(A->boxes);
    for (int32_t i = 0; i < A->n_all; i++)
    {
        free(A->all[i]->box.faces);
        free(A->all[i]);
    }
    free(A->all);
    free(A);
}

/* type 1: grid of param^3 boxes (cAABB::Init, cAccelStruct.cpp:95-179); type 2: octree whose
 * leaves split when they hold param faces (cOctree::Init, :386-424, without the .octree cache).
 * verts must outlive the structure.  NULL when the octree's split recursion does not end. */
accel_t *orc_accel_build(const double *verts, int32_t n_tris, int type, int param)
{
    accel_t *A = (accel_t *)calloc(1, sizeof(accel_t));
    A->type = type, A->verts = verts, A->n_tris = n_tris, A->capacity = param;
    double lower[3], upper[3];
    scene_bounds(verts, n_tris, lower, upper);
    if (type == 1)
    {
        const int D = param;
        box_t *all = (box_t *)calloc((size_t)D * D * D, sizeof(box_t));
        int id[3], nb = 0;
        for (id[0] = 0; id[0] < D; id[0]++)
            for (id[1] = 0; id[1] < D; id[1]++)
                for (id[2] = 0; id[2] < D; id[2]++, nb++)
                    for (int i = 0; i < 3; i++)
                    {
                        all[nb].b[i][0] = lower[i] + (upper[i] - lower[i]) / D * id[i];
                        all[nb].b[i][1] = lower[i] + (upper[i] - lower[i]) / D * (id[i] + 1);
                    }
        for (int32_t f = 0; f < n_tris; f++)
        {
            int placed = 0;
            for (int b = 0; b < nb; b++)
                if (box_face(&all[b], verts + (size_t)f * 9))
                    box_push(&all[b], f), placed = 1;
            if (!placed)
                A->exits++;
        }
        A->boxes = (box_t *)calloc((size_t)nb, sizeof(box_t));
        for (int b = 0; b < nb; b++)
            if (all[b].n)
                A->boxes[A->n_boxes++] = all[b];
        free(all);
    }
    else
    {
        double b[3][2];
        for (int i = 0; i < 3; i++)
            b[i][0] = lower[i], b[i][1] = upper[i];
        A->root = node_new(A, b);
        for (int32_t f = 0; f < n_tris && !A->failed; f++)
            add_face(A, A->root, f, 0);
        if (A->failed)
        {
            orc_accel_free(A);
            return NULL;
        }
        A->stranded += inner_refs(A->root);
    }
    return A;
}

/* out[0..4]: boxes (grid) or nodes reachable from the root (octree); face references queries can
 * reach; face references no query reaches; faces that no reachable leaf or box lists at all; how
 * many times the reference's builder would have stopped the process */
static void count_reachable(const node_t *n, int64_t *nodes, int64_t *refs, uint8_t *seen)
{
    (*nodes)++;
    if (n->leaf)
    {
        *refs += n->box.n;
        for (int32_t i = 0; i < n->box.n; i++)
            seen[n->box.faces[i]] = 1;
        return;
    }
    for (int i = 0; i < 8; i++)
        count_reachable(n->child[i], nodes, refs, seen);
}

void orc_accel_info(const accel_t *A, int64_t *out /*[5]*/)
{
    uint8_t *seen = (uint8_t *)calloc((size_t)(A->n_tris > 0 ? A->n_tris : 1), 1);
    out[0] = out[1] = 0;
    if (A->type == 1)
    {
        out[0] = A->n_boxes;
        for (int32_t b = 0; b < A->n_boxes; b++)
        {
            out[1] += A->boxes[b].n;
            for (int32_t i = 0; i < A->boxes[b].n; i++)
                seen[A->boxes[b].faces[i]] = 1;
        }
    }
    else
        count_reachable(A->root, &out[0], &out[1], seen);
    out[2] = A->stranded;
    out[4] = A->exits;
    out[3] = 0;
    for (int32_t f = 0; f < A->n_tris; f++)
        out[3] += !seen[f];
    free(seen);
}

static inline void try_face(const accel_t *A, int32_t f, const double *o, const double *d, double *dist, int32_t *target, double *pt)
{
    const double *v = A->verts + (size_t)f * 9;
    double p[3];
    if (!orc_tri_raycast(o, d, v, v + 3, v + 6, p))
        return;
    if (isnan(p[0]) || isnan(p[1]) || isnan(p[2]))
        return;
    double e[3] = {p[0] - o[0], p[1] - o[1], p[2] - o[2]};
    double n = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
    if (n < *dist && n > 1e-6)
    {
        pt[0] = p[0], pt[1] = p[1], pt[2] = p[2];
        *dist = n;
        *target = f;
    }
}

/* cAABB::RayCast (cAccelStruct.cpp:181-212) / cOctree::RayCast (:614-669): face id or -1 */
int32_t orc_accel_raycast(const accel_t *A, const double *o, const double *d, double *pt)
{
    double inv[3], dist = DBL_MAX;
    int32_t target = -1;
    inv_dir(d, inv);
    pt[0] = pt[1] = pt[2] = NAN;
    if (A->type == 1)
    {
        for (int32_t b = 0; b < A->n_boxes; b++)
        {
            if (!box_ray(&A->boxes[b], o, inv))
                continue;
            for (int32_t i = 0; i < A->boxes[b].n; i++)
                try_face(A, A->boxes[b].faces[i], o, d, &dist, &target, pt);
        }
        return target;
    }
    /* breadth first over the nodes whose box the ray meets */
    int32_t cap = 256, head = 0, tail = 0;
    const node_t **q = (const node_t **)malloc(sizeof(node_t *) * (size_t)cap);
    q[tail++] = A->root;
    while (head < tail)
    {
        const node_t *n = q[head++];
        if (!box_ray(&n->box, o, inv))
            continue;
        if (n->leaf)
        {
            for (int32_t i = 0; i < n->box.n; i++)
                try_face(A, n->box.faces[i], o, d, &dist, &target, pt);
            continue;
        }
        if (tail + 8 > cap)
        {
            cap *= 2;
            q = (const node_t **)realloc((void *)q, sizeof(node_t *) * (size_t)cap);
        }
        for (int c = 0; c < 8; c++)
            q[tail++] = n->child[c];
    }
    free((void *)q);
    return target;
}

static inline int blocks(const accel_t *A, int32_t f, const double *p1, const double *dir, double dist, double *cur_min)
{
    const double *v = A->verts + (size_t)f * 9;
    double t = orc_tri_raycast_t(p1, dir, v, v + 3, v + 6);
    if (!isnan(t) && t < *cur_min - 1e-6 && t > 1e-6)
    {
        *cur_min = t;
        if (*cur_min < dist)
            return 1;
    }
    return 0;
}

/* cAABB::VisTest (cAccelStruct.cpp:214-263) / cOctree::VisTest (:671-730): 1 = visible */
int orc_accel_vistest(const accel_t *A, const double *p1, const double *p2)
{
    double e[3] = {p1[0] - p2[0], p1[1] - p2[1], p1[2] - p2[2]};
    double dist = sqrt(e[0] * e[0] + e[1] * e[1] + e[2] * e[2]), cur_min = dist;
    double dir[3] = {p2[0] - p1[0], p2[1] - p1[1], p2[2] - p1[2]};
    {
        /* Eigen normalized(): divides by the norm when it is positive */
        double z = dir[0] * dir[0] + dir[1] * dir[1] + dir[2] * dir[2];
        if (z > 0)
        {
            double n = sqrt(z);
            dir[0] /= n, dir[1] /= n, dir[2] /= n;
        }
    }
    if (A->type == 1)
    {
        for (int32_t b = 0; b < A->n_boxes; b++)
        {
            if (!box_segment(&A->boxes[b], p1, p2))
                continue;
            for (int32_t i = 0; i < A->boxes[b].n; i++)
                if (blocks(A, A->boxes[b].faces[i], p1, dir, dist, &cur_min))
                    return 0;
        }
        return 1;
    }
    int32_t cap = 256, head = 0, tail = 0, vis = 1;
    const node_t **q = (const node_t **)malloc(sizeof(node_t *) * (size_t)cap);
    q[tail++] = A->root;
    while (head < tail && vis)
    {
        const node_t *n = q[head++];
        if (!box_segment(&n->box, p1, p2))
            continue;
        if (n->leaf)
        {
            for (int32_t i = 0; i < n->box.n && vis; i++)
                if (blocks(A, n->box.faces[i], p1, dir, dist, &cur_min))
                    vis = 0;
            continue;
        }
        if (tail + 8 > cap)
        {
            cap *= 2;
            q = (const node_t **)realloc((void *)q, sizeof(node_t *) * (size_t)cap);
        }
        for (int c = 0; c < 8; c++)
            q[tail++] = n->child[c];
    }
    free((void *)q);
    return vis;
}

/* n rays at once (OpenMP over rays) */
void orc_accel_raycast_many(const accel_t *A, int64_t n, const double *o, const double *d, int32_t *ids, double *pts)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++)
        ids[i] = orc_accel_raycast(A, o + i * 3, d + i * 3, pts + i * 3);
}

void orc_accel_vistest_many(const accel_t *A, int64_t n, const double *p1, const double *p2, int32_t *vis)
{
#pragma omp parallel for schedule(dynamic, 256)
    for (int64_t i = 0; i < n; i++)
        vis[i] = orc_accel_vistest(A, p1 + i * 3, p2 + i * 3);
}

/* ---- the .octree cache (cOctree::WriteOctree / ReadOctree, cAccelStruct.cpp:485-611) ----------
 * Line of a node: id, leaf flag, the three (low, high) bound pairs, then the eight child ids in
 * Children[x][y][z] order (inner node) or the face ids (leaf), every item followed by one space.
 * A bound pair is printed the way Eigen prints a row vector to a stream of default precision:
 * six significant digits, the two numbers padded on the left to the width of the longer one. */
static void print_pair(FILE *f, double lo, double hi)
{
    char a[64], b[64];
    snprintf(a, sizeof(a), "%g", lo);
    snprintf(b, sizeof(b), "%g", hi);
    int w = (int)strlen(a) > (int)strlen(b) ? (int)strlen(a) : (int)strlen(b);
    fprintf(f, "%*s %*s ", w, a, w, b);
}

/* 0: written; 1: not an octree / cannot open the file */
int orc_octree_write(const accel_t *A, const char *path)
{
    if (!A || A->type != 2 || !A->root)
        return 1;
    FILE *f = fopen(path, "w");
    if (!f)
        return 1;
    /* breadth first from the root, as the reference's queue does */
    const node_t **queue = (const node_t **)malloc(sizeof(node_t *) * (size_t)(A->n_all > 0 ? A->n_all : 1));
    int32_t head = 0, tail = 0;
    queue[tail++] = A->root;
    while (head < tail)
    {
        const node_t *n = queue[head++];
        fprintf(f, "%d %d ", (int)n->id, n->leaf ? 1 : 0);
        for (int i = 0; i < 3; i++)
            print_pair(f, n->box.b[i][0], n->box.b[i][1]);
        if (!n->leaf)
            for (int c = 0; c < 8; c++)
                fprintf(f, "%d ", (int)n->child[c]->id), queue[tail++] = n->child[c];
        else
            for (int32_t i = 0; i < n->box.n; i++)
                fprintf(f, "%d ", (int)n->box.faces[i]);
        fprintf(f, "\n");
    }
    free(queue);
    fclose(f);
    return 0;
}

#define OCTREE_READ_IDS 1000000 /* buffer_num of the reference's reader: larger ids end the process there */

/* Reads a cache file into a structure that answers queries like a built octree.  *status: 0 read;
 * 1 cannot open; 2 an id beyond the reference's buffer (it prints "buffer exceed" and exits);
 * 3 no node 0 / a face id outside the mesh / a malformed line.  verts must outlive the result. */
accel_t *orc_octree_read(const double *verts, int32_t n_tris, const char *path, int *status)
{
    int st_local = 0;
    if (!status)
        status = &st_local;
    *status = 0;
    FILE *f = fopen(path, "r");
    if (!f)
    {
        *status = 1;
        return NULL;
    }
    accel_t *A = (accel_t *)calloc(1, sizeof(accel_t));
    A->type = 2, A->verts = verts, A->n_tris = n_tris;
    node_t **by_id = (node_t **)calloc((size_t)OCTREE_READ_IDS + 1, sizeof(node_t *));
    char *line = NULL;
    size_t cap = 0;
    while (*status == 0)
    {
        /* a line of any length */
        size_t len = 0;
        int ch;
        while ((ch = fgetc(f)) != EOF && ch != '\n')
        {
            if (len + 2 > cap)
                cap = cap ? 2 * cap : 256, line = (char *)realloc(line, cap);
            line[len++] = (char)ch;
        }
        if (len == 0 && ch == EOF)
            break;
        if (cap == 0)
            cap = 256, line = (char *)realloc(line, cap);
        line[len] = 0;
        char *p = line, *e;
        long id = strtol(p, &e, 10);
        if (e == p)
            break; /* the reader stops at the first line without a node id */
        p = e;
        long leaf = strtol(p, &e, 10);
        if (e == p)
        {
            *status = 3;
            break;
        }
        p = e;
        if (id < 0 || id > OCTREE_READ_IDS)
        {
            *status = 2;
            break;
        }
        double b[3][2];
        int ok = 1;
        for (int i = 0; i < 3 && ok; i++)
            for (int j = 0; j < 2 && ok; j++)
            {
                b[i][j] = strtod(p, &e);
                ok = e != p;
                p = e;
            }
        if (!ok)
        {
            *status = 3;
            break;
        }
        node_t *n = by_id[id];
        if (!n)
            n = by_id[id] = node_new(A, b);
        else
            memcpy(n->box.b, b, sizeof(b)); /* made earlier as somebody's child, with a placeholder box */
        n->id = (int32_t)id;
        n->leaf = leaf != 0;
        if (!n->leaf)
        {
            for (int c = 0; c < 8 && *status == 0; c++)
            {
                long cid = strtol(p, &e, 10);
                if (e == p)
                    *status = 3;
                else if (cid < 0 || cid > OCTREE_READ_IDS)
                    *status = 2;
                else
                {
                    if (!by_id[cid])
                        by_id[cid] = node_new(A, b), by_id[cid]->id = (int32_t)cid;
                    n->child[c] = by_id[cid];
                }
                p = e;
            }
        }
        else
        {
            for (;;)
            {
                long face = strtol(p, &e, 10);
                if (e == p)
                    break;
                p = e;
                if (face < 0 || face >= n_tris)
                {
                    *status = 3;
                    break;
                }
                box_push(&n->box, (int32_t)face);
            }
        }
    }
    free(line);
    fclose(f);
    A->root = by_id[0];
    if (*status == 0 && !A->root)
        *status = 3;
    /* an inner node whose line never came keeps null children: refuse it rather than crash in a query */
    for (int32_t i = 0; i < A->n_all && *status == 0; i++)
        if (!A->all[i]->leaf)
            for (int c = 0; c < 8; c++)
                if (!A->all[i]->child[c])
                    *status = 3;
    free(by_id);
    if (*status != 0)
    {
        orc_accel_free(A);
        return NULL;
    }
    return A;
}
