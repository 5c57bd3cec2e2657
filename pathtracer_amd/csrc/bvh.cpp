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

struct Builder
{
    const float *verts;
    std::vector<Box> tri_box;
    std::vector<double> centroid; // [F][3]
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

void build_bvh(const float *verts, int n_tris, Bvh &out, bool median)
{
    Builder B;
    B.verts = verts;
    B.max_leaf = 4;
    B.median = median;
    B.tri_box.resize(n_tris);
    B.centroid.resize(static_cast<size_t>(n_tris) * 3);
    B.order.resize(n_tris);
    double max_abs = 0;
    for (int f = 0; f < n_tris; f++)
    {
        B.order[f] = f;
        double c[3] = {0, 0, 0};
        for (int j = 0; j < 3; j++)
        {
            double p[3];
            for (int k = 0; k < 3; k++)
            {
                p[k] = verts[f * 9 + j * 3 + k];
                c[k] += p[k];
                max_abs = std::max(max_abs, std::fabs(p[k]));
            }
            B.tri_box[f].grow(p);
        }
        for (int k = 0; k < 3; k++)
            B.centroid[f * 3 + k] = c[k] / 3.0;
    }
    B.pad_abs = 1e-6 * (1.0 + max_abs);

    out = Bvh();
    out.median = median;
    out.n_leaves = std::max(n_tris, 0);
    Collapser C{B};
    int need = 0;
    if (n_tris <= 0)
    {
        C.out.emplace_back();
        C.emit(0, nullptr, nullptr, 0); // a root without children
    }
    else if (n_tris <= 4)
    {
        // no tree to speak of: one node whose children are the triangles
        Collapser::Slot all;
        all.ref = ~static_cast<int32_t>(n_tris - 1);
        C.leaf_node(all, 0, need);
    }
    else
    {
        B.build(0, n_tris, 0); // more triangles than a leaf holds: the root is binary node 0
        Collapser::Slot top[4];
        Collapser::children(B.nodes[0], top[0], top[1]);
        C.collapse(top, 2, 0, need);
    }
    out.nodes = std::move(C.out);
    hot_nodes_first(out.nodes, HOT_NODES);
    out.tri_order = std::move(B.order);
    out.max_depth = C.max_depth;
    out.stack_need = need;
}

} // namespace ptx
