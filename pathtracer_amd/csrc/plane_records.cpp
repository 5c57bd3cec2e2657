#include "plane_records.hpp"

#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

namespace ptx
{
namespace
{

typedef __int128 i128;

// float -> integer on a common power-of-two grid (exact), or false when the values of the two
// triangles span too many binary orders of magnitude for the 128-bit determinant below
bool to_grid(const float *const pts[4], long long out[4][3])
{
    int emin = std::numeric_limits<int>::max(), emax = std::numeric_limits<int>::min();
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 3; c++)
        {
            const float x = pts[p][c];
            if (!std::isfinite(x))
                return false;
            if (x == 0.0f)
                continue;
            int e;
            std::frexp(x, &e); // x = m 2^e, 0.5 <= |m| < 1; 24-bit mantissa -> lowest bit at e - 24
            emin = std::min(emin, e - 24);
            emax = std::max(emax, e);
        }
    if (emin > emax)
    {
        std::memset(out, 0, sizeof(long long) * 12);
        return true;
    }
    if (emax - emin > 38)
        return false;
    for (int p = 0; p < 4; p++)
        for (int c = 0; c < 3; c++)
            out[p][c] = static_cast<long long>(std::ldexp(static_cast<double>(pts[p][c]), -emin)); // exact
    return true;
}

// exact sign of det[q1 - q0, q2 - q0, q3 - q0]; coordinates below 2^39
int orient_exact(const long long q[4][3])
{
    i128 a[3], b[3], c[3];
    for (int k = 0; k < 3; k++)
    {
        a[k] = static_cast<i128>(q[1][k] - q[0][k]);
        b[k] = static_cast<i128>(q[2][k] - q[0][k]);
        c[k] = static_cast<i128>(q[3][k] - q[0][k]);
    }
    const i128 det = a[0] * (b[1] * c[2] - b[2] * c[1]) - a[1] * (b[0] * c[2] - b[2] * c[0]) + a[2] * (b[0] * c[1] - b[1] * c[0]);
    return det > 0 ? 1 : (det < 0 ? -1 : 0);
}

bool same_point(const float *p, const float *q) { return std::memcmp(p, q, 12) == 0; }

float round_up(double x)
{
    float f = static_cast<float>(x);
    if (static_cast<double>(f) < x)
        f = std::nextafter(f, std::numeric_limits<float>::infinity());
    return f;
}

} // namespace

void build_plane_records(const float *verts, int n_tris, PlaneRecords &out)
{
    out = PlaneRecords();
    out.rec_tri.assign(64, -1);
    out.tri_rec.assign(std::max(n_tris, 1), 0);
    std::vector<int> partner(n_tris, -1), lone(n_tris, -1); // lone: index (0..2) of A's vertex off the shared edge
    std::vector<int> far_vertex(n_tris, -1);                // index (0..2) of B's vertex off the shared edge

    // ---- pair up the halves of quad faces -------------------------------------------------------
    for (int i = 0; i < n_tris; i++)
    {
        if (partner[i] >= 0)
            continue;
        const float *A = verts + static_cast<size_t>(i) * 9;
        for (int j = i + 1; j < n_tris && partner[i] < 0; j++)
        {
            if (partner[j] >= 0)
                continue;
            const float *B = verts + static_cast<size_t>(j) * 9;
            int shared_a[3] = {0, 0, 0}, shared_b[3] = {0, 0, 0}, n_shared = 0;
            for (int p = 0; p < 3; p++)
                for (int q = 0; q < 3; q++)
                    if (!shared_b[q] && !shared_a[p] && same_point(A + 3 * p, B + 3 * q))
                        shared_a[p] = shared_b[q] = 1, n_shared++;
            if (n_shared != 2)
                continue;
            int la = 0, lb = 0;
            for (int p = 0; p < 3; p++)
            {
                if (!shared_a[p])
                    la = p;
                if (!shared_b[p])
                    lb = p;
            }
            const float *pts[4] = {A + 3 * la, A + 3 * ((la + 1) % 3), A + 3 * ((la + 2) % 3), B + 3 * lb};
            long long g[4][3];
            if (!to_grid(pts, g) || orient_exact(g) != 0)
                continue;
            // B's free vertex must lie strictly beyond the shared edge (u + v > 1 in A's basis)
            double e1[3], e2[3], sv[3], n[3];
            for (int c = 0; c < 3; c++)
            {
                e1[c] = double(pts[1][c]) - double(pts[0][c]);
                e2[c] = double(pts[2][c]) - double(pts[0][c]);
                sv[c] = double(pts[3][c]) - double(pts[0][c]);
            }
            n[0] = e1[1] * e2[2] - e1[2] * e2[1], n[1] = e1[2] * e2[0] - e1[0] * e2[2], n[2] = e1[0] * e2[1] - e1[1] * e2[0];
            const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
            if (!(nn > 0))
                continue;
            const double ax = e2[1] * n[2] - e2[2] * n[1], ay = e2[2] * n[0] - e2[0] * n[2], az = e2[0] * n[1] - e2[1] * n[0];
            const double bx = n[1] * e1[2] - n[2] * e1[1], by = n[2] * e1[0] - n[0] * e1[2], bz = n[0] * e1[1] - n[1] * e1[0];
            const double us = (ax * sv[0] + ay * sv[1] + az * sv[2]) / nn, vs = (bx * sv[0] + by * sv[1] + bz * sv[2]) / nn;
            if (!(us + vs > 1.0 + 1e-9))
                continue;
            partner[i] = j, partner[j] = i;
            lone[i] = la, far_vertex[j] = lb;
        }
    }

    // ---- records, in face order of their first triangle ----------------------------------------
    const double eps = std::ldexp(1.0, -24);
    for (int i = 0; i < n_tris; i++)
    {
        if (partner[i] >= 0 && partner[i] < i)
            continue; // second half of a pair
        const int r = out.n_recs++;
        out.rec.resize(static_cast<size_t>(out.n_recs) * PLANE_REC_STRIDE, 0.f);
        float *o = &out.rec[static_cast<size_t>(r) * PLANE_REC_STRIDE];
        const float *A = verts + static_cast<size_t>(i) * 9;
        const int la = partner[i] >= 0 ? lone[i] : 0;
        const float *P0 = A + 3 * la, *P1 = A + 3 * ((la + 1) % 3), *P2 = A + 3 * ((la + 2) % 3);
        if (r < 32)
            out.rec_tri[2 * r] = i;
        out.tri_rec[i] = r;
        double e1[3], e2[3], n[3], p0[3];
        for (int c = 0; c < 3; c++)
        {
            p0[c] = P0[c];
            e1[c] = double(P1[c]) - p0[c];
            e2[c] = double(P2[c]) - p0[c];
        }
        n[0] = e1[1] * e2[2] - e1[2] * e2[1], n[1] = e1[2] * e2[0] - e1[0] * e2[2], n[2] = e1[0] * e2[1] - e1[1] * e2[0];
        const double nn = n[0] * n[0] + n[1] * n[1] + n[2] * n[2];
        if (!(nn > 0) || !std::isfinite(nn))
        {
            // degenerate triangle: the exact test never accepts it -> no candidate bit at all
            if (r < 32)
                out.rec_tri[2 * r] = -1;
            o[13] = o[15] = -std::numeric_limits<float>::infinity(); // NaN-free and always "outside"
            o[22] = std::numeric_limits<float>::infinity();
            continue;
        }
        const double inv_len = 1.0 / std::sqrt(nn);
        const double a[3] = {(e2[1] * n[2] - e2[2] * n[1]) / nn, (e2[2] * n[0] - e2[0] * n[2]) / nn, (e2[0] * n[1] - e2[1] * n[0]) / nn};
        const double b[3] = {(n[1] * e1[2] - n[2] * e1[1]) / nn, (n[2] * e1[0] - n[0] * e1[2]) / nn, (n[0] * e1[1] - n[1] * e1[0]) / nn};
        const double c0 = (n[0] * p0[0] + n[1] * p0[1] + n[2] * p0[2]) * inv_len;
        const double a0 = a[0] * p0[0] + a[1] * p0[1] + a[2] * p0[2], b0 = b[0] * p0[0] + b[1] * p0[1] + b[2] * p0[2];
        double ulo = 0, uhi = 1, vlo = 0, vhi = 1;
        o[22] = std::numeric_limits<float>::infinity();
        if (partner[i] >= 0)
        {
            const int j = partner[i];
            const float *S = verts + static_cast<size_t>(j) * 9 + 3 * far_vertex[j];
            const double sv[3] = {double(S[0]) - p0[0], double(S[1]) - p0[1], double(S[2]) - p0[2]};
            const double us = a[0] * sv[0] + a[1] * sv[1] + a[2] * sv[2], vs = b[0] * sv[0] + b[1] * sv[1] + b[2] * sv[2];
            ulo = std::min(ulo, us), uhi = std::max(uhi, us), vlo = std::min(vlo, vs), vhi = std::max(vhi, vs);
            if (r < 32)
                out.rec_tri[2 * r + 1] = j;
            out.tri_rec[j] = r;
            o[22] = 0.f;
            out.n_pairs++;
        }
        const double a1 = std::fabs(a[0]) + std::fabs(a[1]) + std::fabs(a[2]), b1 = std::fabs(b[0]) + std::fabs(b[1]) + std::fabs(b[2]);
        const double E = 64 * eps * std::max(a1, b1), E0 = 16 * eps * (std::max(std::fabs(a0), std::fabs(b0)) + 1);
        for (int c = 0; c < 3; c++)
        {
            o[c] = static_cast<float>(n[c] * inv_len);
            o[4 + c] = static_cast<float>(a[c]);
            o[8 + c] = static_cast<float>(b[c]);
        }
        o[3] = static_cast<float>(c0);
        // the device compares a.X (not a.X - a.P0) with the range, so the offsets move into the
        // centres; half widths carry the constant part of the error bound
        o[12] = static_cast<float>(0.5 * (ulo + uhi) + a0), o[13] = round_up(0.5 * (uhi - ulo) * (1 + 4 * eps) + 4 * eps * (1 + std::fabs(a0)) + E0);
        o[14] = static_cast<float>(0.5 * (vlo + vhi) + b0), o[15] = round_up(0.5 * (vhi - vlo) * (1 + 4 * eps) + 4 * eps * (1 + std::fabs(b0)) + E0);
        o[16] = round_up(E), o[17] = round_up(2 * E0 + 8 * eps * (1 + std::fabs(a0) + std::fabs(b0)));
        o[18] = static_cast<float>(1.0 + a0 + b0);
        o[20] = 32.0f / 16777216.0f; // = PQ_ET of the device code
        o[21] = round_up(std::fabs(c0));
        out.smax = std::max(out.smax, o[21]);
    }
}

std::vector<uint32_t> blocker_masks(const float *verts, const PlaneRecords &pr, const double light[4][3])
{
    const int n = pr.n_recs;
    std::vector<uint32_t> masks(32, 0xFFFFFFFFu);
    if (n > 32)
        return masks;
    // vertices of every record (3 or 4) and the plane of its first triangle, in double
    struct RecGeo
    {
        double v[4][3];
        int nv = 0;
        double nrm[3] = {0, 0, 0}, c = 0;
        bool plane_ok = false;
    };
    std::vector<RecGeo> geo(n);
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300};
    for (int r = 0; r < n; r++)
    {
        RecGeo &g = geo[r];
        for (int half = 0; half < 2; half++)
        {
            const int t = pr.rec_tri[2 * r + half];
            if (t < 0)
                continue;
            for (int k = 0; k < 3; k++)
            {
                const float *p = verts + static_cast<size_t>(t) * 9 + 3 * k;
                bool seen = false;
                for (int q = 0; q < g.nv; q++)
                    seen = seen || (g.v[q][0] == p[0] && g.v[q][1] == p[1] && g.v[q][2] == p[2]);
                if (!seen && g.nv < 4)
                {
                    for (int c3 = 0; c3 < 3; c3++)
                        g.v[g.nv][c3] = p[c3];
                    g.nv++;
                }
            }
        }
        const int t0 = pr.rec_tri[2 * r];
        if (t0 >= 0)
        {
            const float *p = verts + static_cast<size_t>(t0) * 9;
            double e1[3], e2[3];
            for (int c3 = 0; c3 < 3; c3++)
                e1[c3] = double(p[3 + c3]) - double(p[c3]), e2[c3] = double(p[6 + c3]) - double(p[c3]);
            double nx = e1[1] * e2[2] - e1[2] * e2[1], ny = e1[2] * e2[0] - e1[0] * e2[2], nz = e1[0] * e2[1] - e1[1] * e2[0];
            const double len = std::sqrt(nx * nx + ny * ny + nz * nz);
            if (len > 0 && std::isfinite(len))
            {
                g.nrm[0] = nx / len, g.nrm[1] = ny / len, g.nrm[2] = nz / len;
                g.c = g.nrm[0] * p[0] + g.nrm[1] * p[1] + g.nrm[2] * p[2];
                g.plane_ok = true;
            }
        }
        for (int q = 0; q < g.nv; q++)
            for (int c3 = 0; c3 < 3; c3++)
                lo[c3] = std::min(lo[c3], g.v[q][c3]), hi[c3] = std::max(hi[c3], g.v[q][c3]);
    }
    for (int q = 0; q < 4; q++)
        for (int c3 = 0; c3 < 3; c3++)
        {
            if (!std::isfinite(light[q][c3]))
                return masks;
            lo[c3] = std::min(lo[c3], light[q][c3]), hi[c3] = std::max(hi[c3], light[q][c3]);
        }
    const double diag = std::sqrt((hi[0] - lo[0]) * (hi[0] - lo[0]) + (hi[1] - lo[1]) * (hi[1] - lo[1]) + (hi[2] - lo[2]) * (hi[2] - lo[2]));
    // An end point may sit on the wrong side of the plane by rounding, by no more than
    // delta = 1e-12 diag (thousands of ulps of the largest coordinate).  With the other end `clear`
    // away from the plane, the crossing then lies within delta * diag / clear of that end point;
    // the visibility test ignores whatever lies within 1e-6 (absolute) of an end point, and
    // clear = 1e-5 diag^2 keeps the crossing ten times closer than that.
    const double tol = 1e-9 * diag, clear = 1e-5 * diag * diag;
    for (int i = 0; i < n; i++)
    {
        uint32_t m = 0;
        for (int j = 0; j < n; j++)
        {
            const RecGeo &pj = geo[j];
            bool can_block = true;
            if (pj.plane_ok && geo[i].nv >= 3) // j == i too: a segment leaving record i towards a light clear of
                                               // its plane does not meet that plane again
            {
                for (int sgn = -1; sgn <= 1 && can_block; sgn += 2)
                {
                    double min_i = 1e300, min_l = 1e300;
                    for (int q = 0; q < geo[i].nv; q++)
                        min_i = std::min(min_i, sgn * (pj.nrm[0] * geo[i].v[q][0] + pj.nrm[1] * geo[i].v[q][1] + pj.nrm[2] * geo[i].v[q][2] - pj.c));
                    for (int q = 0; q < 4; q++)
                        min_l = std::min(min_l, sgn * (pj.nrm[0] * light[q][0] + pj.nrm[1] * light[q][1] + pj.nrm[2] * light[q][2] - pj.c));
                    if (min_i >= -tol && min_l >= -tol && std::max(min_i, min_l) >= clear)
                        can_block = false;
                }
            }
            if (can_block)
                m |= 1u << j;
        }
        masks[i] = m;
    }
    return masks;
}

void square_light_corners(const Light &L, double corners[4][3])
{
    for (int c3 = 0; c3 < 3; c3++)
    {
        const double e10 = L.p[1][c3] - L.p[0][c3], e21 = L.p[2][c3] - L.p[1][c3];
        corners[0][c3] = L.p[0][c3], corners[1][c3] = L.p[0][c3] + e10;
        corners[2][c3] = L.p[0][c3] + e10 + e21, corners[3][c3] = L.p[0][c3] + e21;
    }
}

} // namespace ptx
