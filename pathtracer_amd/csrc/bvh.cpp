// Binned-SAH BVH builder (host), collapsed to four children per node.  Replaces the reference's accel builders
// (src/geometry/cAccelStruct.cpp:95-179 uniform grid, :291-482 octree) with a structure laid out
// for the GPU; hit semantics stay those of the brute-force loops (src/geometry/cAccelStruct.cpp:28-87).
#include "bvh.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <queue>
#include <utility>

namespace ptx
{

namespace
{

struct Box
{
    double lo[3], hi[3];
    Box()
    {
        for (int k = 0; k < 3; k++)
            lo[k] = std::numeric_limits<double>::infinity(), hi[k] = -std::numeric_limits<double>::infinity();
    }
    void grow(const double *p)
    {
        for (int k = 0; k < 3; k++)
        {
            lo[k] = std::min(lo[k], p[k]);
            hi[k] = std::max(hi[k], p[k]);
        }
    }
    void grow(const Box &b)
    {
        for (int k = 0; k < 3; k++)
        {
            lo[k] = std::min(lo[k], b.lo[k]);
            hi[k] = std::max(hi[k], b.hi[k]);
        }
    }
    double area() const
    {
        double e0 = hi[0] - lo[0], e1 = hi[1] - lo[1], e2 = hi[2] - lo[2];
        if (e0 < 0)
            return 0;
        return 2 * (e0 * e1 + e1 * e2 + e2 * e0);
    }
};

// node of the binary tree the builder makes first
struct Node2
{
    float lo0[3], hi0[3], lo1[3], hi1[3];
    int32_t c0, c1; // >= 0: node index; < 0: leaf, ~c = (first << 3) | (count - 1)
};

// A reference: one triangle, or the part of one that lies in `box` (reference splitting below).
// The builder only ever sees references: boxes and box centres.
struct Builder
{
    const float *verts;
    std::vector<Box> tri_box;     // [references]
    std::vector<double> centroid; // [references][3]
    std::vector<int32_t> ref_tri; // [references] the triangle a reference belongs to
    std::vector<int32_t> order;
    std::vector<Node2> nodes;
    bool median = false;
    int max_leaf, n_leaves = 0, max_depth = 0, max_leaf_seen = 0;
    double pad_abs;

    Box range_box(int first, int count) const
    {
        Box b;
        for (int i = first; i < first + count; i++)
            b.grow(tri_box[order[i]]);
        return b;
    }

    void store_box(const Box &b, float *lo, float *hi) const
    {
        for (int k = 0; k < 3; k++)
        {
            double l = b.lo[k] - pad_abs, h = b.hi[k] + pad_abs;
            float fl = static_cast<float>(l), fh = static_cast<float>(h);
            if (static_cast<double>(fl) > l)
                fl = std::nextafter(fl, -std::numeric_limits<float>::infinity());
            if (static_cast<double>(fh) < h)
                fh = std::nextafter(fh, std::numeric_limits<float>::infinity());
            lo[k] = fl, hi[k] = fh;
        }
    }

    // returns child reference (node index or encoded leaf)
    int32_t build(int first, int count, int depth)
    {
        max_depth = std::max(max_depth, depth);
        if (count <= max_leaf)
            return make_leaf(first, count);

        Box cb;
        for (int i = first; i < first + count; i++)
            cb.grow(&centroid[order[i] * 3]);
        int axis = 0;
        double ext = cb.hi[0] - cb.lo[0];
        for (int k = 1; k < 3; k++)
            if (cb.hi[k] - cb.lo[k] > ext)
                ext = cb.hi[k] - cb.lo[k], axis = k;

        int mid = first + count / 2;
        if (median)
        {
            std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                             [&](int32_t a, int32_t b) { return centroid[a * 3 + axis] < centroid[b * 3 + axis]; });
        }
        else if (ext > 0)
        {
            // surface-area heuristic over binned centroids, all three axes: the split with the
            // smallest  area(left) * count(left) + area(right) * count(right)
            constexpr int NB = 32;
            double best = std::numeric_limits<double>::infinity();
            int best_b = -1, best_axis = axis;
            double best_k1 = 0;
            for (int ax = 0; ax < 3; ax++)
            {
                const double e = cb.hi[ax] - cb.lo[ax];
                if (!(e > 0))
                    continue;
                Box bb[NB];
                int bc[NB] = {0};
                const double k1 = NB * (1 - 1e-9) / e;
                for (int i = first; i < first + count; i++)
                {
                    int b = static_cast<int>((centroid[order[i] * 3 + ax] - cb.lo[ax]) * k1);
                    b = std::min(std::max(b, 0), NB - 1);
                    bb[b].grow(tri_box[order[i]]);
                    bc[b]++;
                }
                double right_area[NB];
                int right_cnt[NB];
                Box acc;
                int c = 0;
                for (int b = NB - 1; b >= 0; b--)
                {
                    acc.grow(bb[b]);
                    c += bc[b];
                    right_area[b] = acc.area(), right_cnt[b] = c;
                }
                acc = Box();
                c = 0;
                for (int b = 0; b < NB - 1; b++)
                {
                    acc.grow(bb[b]);
                    c += bc[b];
                    if (c == 0 || right_cnt[b + 1] == 0)
                        continue;
                    const double cost = acc.area() * c + right_area[b + 1] * right_cnt[b + 1];
                    if (cost < best)
                        best = cost, best_b = b, best_axis = ax, best_k1 = k1;
                }
            }
            if (best_b >= 0)
            {
                const double lo = cb.lo[best_axis];
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](int32_t t) {
                    int b = static_cast<int>((centroid[t * 3 + best_axis] - lo) * best_k1);
                    b = std::min(std::max(b, 0), NB - 1);
                    return b <= best_b;
                });
                mid = static_cast<int>(it - order.begin());
            }
        }
        if (mid <= first || mid >= first + count)
        {
            // degenerate (all centroids equal): split by index
            mid = first + count / 2;
        }
        int32_t me = static_cast<int32_t>(nodes.size());
        nodes.emplace_back();
        Box b0 = range_box(first, mid - first), b1 = range_box(mid, first + count - mid);
        int32_t c0 = build(first, mid - first, depth + 1);
        int32_t c1 = build(mid, first + count - mid, depth + 1);
        Node2 &nd = nodes[me];
        store_box(b0, nd.lo0, nd.hi0);
        store_box(b1, nd.lo1, nd.hi1);
        nd.c0 = c0, nd.c1 = c1;
        return me;
    }

    int32_t make_leaf(int first, int count)
    {
        // leaves hold at most 8 triangles; larger degenerate groups are chained through nodes
        if (count > 8)
        {
            int32_t me = static_cast<int32_t>(nodes.size());
            nodes.emplace_back();
            int half = count / 2;
            Box b0 = range_box(first, half), b1 = range_box(first + half, count - half);
            int32_t c0 = make_leaf(first, half), c1 = make_leaf(first + half, count - half);
            Node2 &nd = nodes[me];
            store_box(b0, nd.lo0, nd.hi0);
            store_box(b1, nd.lo1, nd.hi1);
            nd.c0 = c0, nd.c1 = c1;
            return me;
        }
        n_leaves++;
        max_leaf_seen = std::max(max_leaf_seen, count);
        return ~static_cast<int32_t>((first << 3) | (count - 1));
    }
};


// ---------------------------------------------------------------------------------------------
// Reference splitting before the build.  A large triangle among small ones (a wall behind
// furniture) has a box that overlaps everything in front of it, and every ray through that space
// visits the triangle's leaf.  So the triangles with the emptiest boxes are cut -- not the
// geometry, only the REFERENCES: a reference is a convex polygon (the triangle clipped to a box),
// it is cut by an axis plane into two polygons, and each half goes into the tree with the box of
// its own polygon.  The tree then holds some triangles in several leaves; the intersector tests
// the whole triangle wherever it meets it, so hits, distances and ties are those of the unsplit
// tree (the same triangle found twice is the same hit).  Which reference is cut next: the one with
// the largest  box area - 2 x (sum of the polygon's three axis projections)  -- the surface its box
// has beyond what a flat polygon needs.  Where: the coarsest plane of a power-of-two grid over the
// scene box that crosses it (neighbouring references are cut by the same planes and their halves
// line up).  The number of extra references is bounded (`budget`).
// ---------------------------------------------------------------------------------------------
struct Poly
{
    int n = 0;
    double p[10][3]; // a triangle clipped by six planes has at most nine corners
    Box box() const
    {
        Box b;
        for (int i = 0; i < n; i++)
            b.grow(p[i]);
        return b;
    }
    // twice the sum of the polygon's areas projected along the three axes
    double proj2() const
    {
        double a[3] = {0, 0, 0};
        for (int i = 1; i + 1 < n; i++)
        {
            const double u[3] = {p[i][0] - p[0][0], p[i][1] - p[0][1], p[i][2] - p[0][2]};
            const double v[3] = {p[i + 1][0] - p[0][0], p[i + 1][1] - p[0][1], p[i + 1][2] - p[0][2]};
            a[0] += u[1] * v[2] - u[2] * v[1];
            a[1] += u[2] * v[0] - u[0] * v[2];
            a[2] += u[0] * v[1] - u[1] * v[0];
        }
        return std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]);
    }
    // the part on the side `keep_low ? x[axis] <= c : x[axis] >= c` (closed: a corner in the plane belongs to both)
    Poly clip(int axis, double c, bool keep_low) const
    {
        Poly out;
        for (int i = 0; i < n; i++)
        {
            const double *a = p[i], *b = p[(i + 1) % n];
            const double da = keep_low ? a[axis] - c : c - a[axis], db = keep_low ? b[axis] - c : c - b[axis];
            if (da <= 0 && out.n < 10)
                std::memcpy(out.p[out.n++], a, sizeof(double) * 3);
            if ((da < 0 && db > 0) || (da > 0 && db < 0))
            {
                const double t = da / (da - db);
                double q[3];
                for (int k = 0; k < 3; k++)
                    q[k] = a[k] + t * (b[k] - a[k]);
                q[axis] = c;
                if (out.n < 10)
                    std::memcpy(out.p[out.n++], q, sizeof(double) * 3);
            }
        }
        return out;
    }
};

struct SplitRef
{
    Poly poly;
    Box box;
    int32_t tri;
    double priority;
};

double split_priority(const SplitRef &r)
{
    return r.box.area() - r.poly.proj2();
}

// the coarsest plane of the power-of-two grid over [lo, hi] of the scene that lies strictly inside (a, b)
bool grid_plane(double lo, double hi, double a, double b, double &plane, int &level)
{
    if (!(hi > lo) || !(b > a))
        return false;
    const double ua = (a - lo) / (hi - lo);
    for (level = 1; level <= 40; level++)
    {
        const double cells = std::ldexp(1.0, level);
        const double k = std::floor(ua * cells) + 1; // first plane above a
        const double c = lo + (hi - lo) * (k / cells);
        if (c > a && c < b)
        {
            plane = c;
            return true;
        }
    }
    return false;
}

void split_references(const float *verts, int n_tris, double budget, std::vector<SplitRef> &refs)
{
    refs.resize(n_tris);
    Box scene;
    for (int f = 0; f < n_tris; f++)
    {
        SplitRef &r = refs[f];
        r.poly.n = 3;
        for (int j = 0; j < 3; j++)
            for (int k = 0; k < 3; k++)
                r.poly.p[j][k] = verts[f * 9 + j * 3 + k];
        r.box = r.poly.box();
        r.tri = f;
        r.priority = split_priority(r);
        scene.grow(r.box);
    }
    int64_t extra = static_cast<int64_t>(budget * n_tris);
    if (extra <= 0 || n_tris < 16)
        return;
    // references whose box has less spare surface than a 4096th of the scene's are left alone
    const double floor_priority = scene.area() / 4096.0 * 1e-3;
    std::priority_queue<std::pair<double, int32_t>> open;
    for (int f = 0; f < n_tris; f++)
        if (refs[f].priority > floor_priority)
            open.push({refs[f].priority, f});
    while (extra > 0 && !open.empty())
    {
        const int32_t i = open.top().second;
        open.pop();
        // the most important grid plane over the three axes (lowest level; ties: the longest extent)
        int best_axis = -1, best_level = 99;
        double best_plane = 0, best_ext = -1;
        for (int ax = 0; ax < 3; ax++)
        {
            double plane;
            int level;
            if (!grid_plane(scene.lo[ax], scene.hi[ax], refs[i].box.lo[ax], refs[i].box.hi[ax], plane, level))
                continue;
            const double ext = refs[i].box.hi[ax] - refs[i].box.lo[ax];
            if (level < best_level || (level == best_level && ext > best_ext))
                best_axis = ax, best_level = level, best_plane = plane, best_ext = ext;
        }
        if (best_axis < 0)
            continue;
        SplitRef lo_half, hi_half;
        lo_half.poly = refs[i].poly.clip(best_axis, best_plane, true);
        hi_half.poly = refs[i].poly.clip(best_axis, best_plane, false);
        if (lo_half.poly.n < 3 || hi_half.poly.n < 3)
            continue; // the plane only touches the polygon
        lo_half.tri = hi_half.tri = refs[i].tri;
        lo_half.box = lo_half.poly.box(), hi_half.box = hi_half.poly.box();
        // a half's box never leaves the box it was cut from (rounding of the cut points aside)
        for (int k = 0; k < 3; k++)
        {
            lo_half.box.lo[k] = std::max(lo_half.box.lo[k], refs[i].box.lo[k]), lo_half.box.hi[k] = std::min(lo_half.box.hi[k], refs[i].box.hi[k]);
            hi_half.box.lo[k] = std::max(hi_half.box.lo[k], refs[i].box.lo[k]), hi_half.box.hi[k] = std::min(hi_half.box.hi[k], refs[i].box.hi[k]);
        }
        lo_half.priority = split_priority(lo_half), hi_half.priority = split_priority(hi_half);
        // cutting must pay: the two boxes together clearly smaller than the one
        if (!(lo_half.box.area() + hi_half.box.area() < 0.9 * refs[i].box.area()))
            continue;
        refs[i] = lo_half;
        refs.push_back(hi_half);
        extra--;
        if (lo_half.priority > floor_priority)
            open.push({lo_half.priority, i});
        if (hi_half.priority > floor_priority)
            open.push({hi_half.priority, static_cast<int32_t>(refs.size()) - 1});
    }
}


// collapse of the binary tree into 4-wide nodes.  The binary tree ends in leaves of up to four
// triangles; a 4-wide node adopts, largest box first, the children of its inner children and the
// triangles of its leaf children while it has room, and a leaf that is left over becomes a node of
// triangle children itself.
struct Collapser
{
    const Builder &B;
    std::vector<BvhNode> out;
    int max_depth = 0;

    struct Slot
    {
        int32_t ref; // >= 0: binary node; < 0: leaf, ~ref = (first << 3) | (count - 1)
        float lo[3], hi[3];
        double area() const
        {
            double e0 = double(hi[0]) - lo[0], e1 = double(hi[1]) - lo[1], e2 = double(hi[2]) - lo[2];
            return e0 * e1 + e1 * e2 + e2 * e0;
        }
        int leaf_first() const { return (~ref) >> 3; }
        int leaf_count() const { return ((~ref) & 7) + 1; }
    };
    static void children(const Node2 &n, Slot &a, Slot &b)
    {
        a.ref = n.c0, b.ref = n.c1;
        for (int k = 0; k < 3; k++)
            a.lo[k] = n.lo0[k], a.hi[k] = n.hi0[k], b.lo[k] = n.lo1[k], b.hi[k] = n.hi1[k];
    }
    Slot triangle(int k) const
    {
        Slot s;
        s.ref = ~static_cast<int32_t>((k << 3) | 0);
        B.store_box(B.tri_box[B.order[k]], s.lo, s.hi);
        return s;
    }

    void emit(int32_t me, const Slot *slot, const int32_t *ref, int n)
    {
        BvhNode &nd = out[me];
        std::memset(&nd, 0, sizeof(nd));
        for (int i = 0; i < 4; i++)
        {
            nd.child[i] = i < n ? ref[i] : BVH_EMPTY;
            for (int k = 0; k < 3; k++)
            {
                nd.lo[k][i] = i < n ? slot[i].lo[k] : std::numeric_limits<float>::infinity();
                nd.hi[k][i] = i < n ? slot[i].hi[k] : std::numeric_limits<float>::infinity();
            }
        }
    }

    // a node whose children are the triangles of one leaf
    int32_t leaf_node(const Slot &leaf, int depth, int &need)
    {
        max_depth = std::max(max_depth, depth);
        const int32_t me = static_cast<int32_t>(out.size());
        out.emplace_back();
        Slot slot[4];
        int32_t ref[4];
        const int n = leaf.leaf_count(); // <= 4
        for (int i = 0; i < n; i++)
        {
            slot[i] = triangle(leaf.leaf_first() + i);
            ref[i] = ~static_cast<int32_t>(leaf.leaf_first() + i);
        }
        emit(me, slot, ref, n);
        need = n - 1;
        return me;
    }

    // returns the index of the 4-wide node made from `slot[0 .. n)`, and through `need` the stack
    // entries a traversal of its subtree can hold at once
    int32_t collapse(Slot *slot, int n, int depth, int &need)
    {
        max_depth = std::max(max_depth, depth);
        while (n < 4)
        {
            int pick = -1;
            double best = -1;
            for (int i = 0; i < n; i++)
            {
                const bool inner = slot[i].ref >= 0;
                const bool open_leaf = !inner && slot[i].leaf_count() > 1 && n - 1 + slot[i].leaf_count() <= 4;
                if ((inner || open_leaf) && slot[i].area() > best)
                    best = slot[i].area(), pick = i;
            }
            if (pick < 0)
                break;
            if (slot[pick].ref >= 0)
            {
                const Node2 &c = B.nodes[slot[pick].ref];
                children(c, slot[pick], slot[n]);
                n++;
            }
            else
            {
                const int first = slot[pick].leaf_first(), cnt = slot[pick].leaf_count();
                slot[pick] = triangle(first);
                for (int i = 1; i < cnt; i++)
                    slot[n++] = triangle(first + i);
            }
        }
        const int32_t me = static_cast<int32_t>(out.size());
        out.emplace_back();
        int deepest = 0;
        int32_t ref[4];
        for (int i = 0; i < n; i++)
        {
            int sub = 0;
            if (slot[i].ref >= 0)
            {
                Slot sub_slot[4];
                children(B.nodes[slot[i].ref], sub_slot[0], sub_slot[1]);
                ref[i] = collapse(sub_slot, 2, depth + 1, sub);
            }
            else if (slot[i].leaf_count() > 1)
                ref[i] = leaf_node(slot[i], depth + 1, sub);
            else
                ref[i] = ~static_cast<int32_t>(slot[i].leaf_first());
            deepest = std::max(deepest, sub);
        }
        emit(me, slot, ref, n);
        need = (n - 1) + deepest;
        return me;
    }
};

// Renumbers the nodes so that the ones a ray is most likely to visit come first: starting at the
// root, the node with the largest box (the chance that a random line meets a convex box grows with
// its surface area) is taken next, `hot` times; the other nodes follow in their depth-first order.
// The tracers keep the first nodes in LDS, so the low ids should be the busy ones.
constexpr int HOT_NODES = 4096;
void hot_nodes_first(std::vector<BvhNode> &nodes, int hot)
{
    const int n = static_cast<int>(nodes.size());
    if (n <= 2)
        return;
    auto child_area = [&](const BvhNode &nd, int j) {
        const double e0 = double(nd.hi[0][j]) - nd.lo[0][j], e1 = double(nd.hi[1][j]) - nd.lo[1][j], e2 = double(nd.hi[2][j]) - nd.lo[2][j];
        return e0 * e1 + e1 * e2 + e2 * e0;
    };
    std::vector<int32_t> new_id(n, -1);
    std::priority_queue<std::pair<double, int32_t>> open; // (area, old id); equal areas: the higher old id first, any fixed rule will do
    open.push({std::numeric_limits<double>::infinity(), 0});
    int next = 0;
    while (!open.empty() && next < hot)
    {
        const int32_t me = open.top().second;
        open.pop();
        new_id[me] = next++;
        for (int j = 0; j < 4; j++)
            if (nodes[me].child[j] >= 0)
                open.push({child_area(nodes[me], j), nodes[me].child[j]});
    }
    for (int i = 0; i < n; i++)
        if (new_id[i] < 0)
            new_id[i] = next++;
    std::vector<BvhNode> out(n);
    for (int i = 0; i < n; i++)
    {
        BvhNode nd = nodes[i];
        for (int j = 0; j < 4; j++)
            if (nd.child[j] >= 0)
                nd.child[j] = new_id[nd.child[j]];
        out[new_id[i]] = nd;
    }
    nodes.swap(out);
}

} // namespace

void build_bvh(const float *verts, int n_tris, Bvh &out, bool median, double split_budget)
{
    Builder B;
    B.verts = verts;
    B.max_leaf = 4;
    B.median = median;
    std::vector<SplitRef> refs;
    split_references(verts, std::max(n_tris, 0), median ? 0.0 : split_budget, refs);
    const int n_refs = static_cast<int>(refs.size());
    B.tri_box.resize(n_refs);
    B.centroid.resize(static_cast<size_t>(n_refs) * 3);
    B.ref_tri.resize(n_refs);
    B.order.resize(n_refs);
    double max_abs = 0;
    for (int r = 0; r < n_refs; r++)
    {
        B.order[r] = r;
        B.tri_box[r] = refs[r].box;
        B.ref_tri[r] = refs[r].tri;
        for (int k = 0; k < 3; k++)
        {
            B.centroid[static_cast<size_t>(r) * 3 + k] = 0.5 * (refs[r].box.lo[k] + refs[r].box.hi[k]);
            max_abs = std::max(max_abs, std::max(std::fabs(refs[r].box.lo[k]), std::fabs(refs[r].box.hi[k])));
        }
    }
    B.pad_abs = 1e-6 * (1.0 + max_abs);

    out = Bvh();
    out.median = median;
    out.n_leaves = n_refs;
    Collapser C{B};
    int need = 0;
    if (n_refs <= 0)
    {
        C.out.emplace_back();
        C.emit(0, nullptr, nullptr, 0); // a root without children
    }
    else if (n_refs <= 4)
    {
        // no tree to speak of: one node whose children are the triangles
        Collapser::Slot all;
        all.ref = ~static_cast<int32_t>(n_refs - 1);
        C.leaf_node(all, 0, need);
    }
    else
    {
        B.build(0, n_refs, 0); // more triangles than a leaf holds: the root is binary node 0
        Collapser::Slot top[4];
        Collapser::children(B.nodes[0], top[0], top[1]);
        C.collapse(top, 2, 0, need);
    }
    out.nodes = std::move(C.out);
    hot_nodes_first(out.nodes, HOT_NODES);
    out.tri_order.resize(n_refs);
    for (int k = 0; k < n_refs; k++)
        out.tri_order[k] = B.ref_tri[B.order[k]];
    out.max_depth = C.max_depth;
    out.stack_need = need;
}

} // namespace ptx
