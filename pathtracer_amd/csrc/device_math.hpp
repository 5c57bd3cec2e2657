// Device-side arithmetic of the path tracer (HIP, gfx950).
//
// Everything is templated on the scalar type R: R = double reproduces the reference's arithmetic
// operation by operation (this translation unit is compiled with -ffp-contract=off, so no product
// is fused into a following sum); R = float is the fast path.
//
// Reference formulas restated here (paths relative to the reference root):
//   ray/triangle ........ src/util/cMathUtil.cpp:141-163,205-227 (3x3 cofactor inverse)
//   reflect / refract ... src/util/cGeoUtil.cpp:117-127, 131-172
//   barycentric weights . src/util/cGeoUtil.cpp:174-231
//   hemisphere sample ... src/util/cMathUtil.cpp:230-267, 40-55, 340-345
//   lights .............. src/tracing/cLight.cpp:73-109, 188-255
//   BxDFs ............... src/tracing/cBxDF.cpp:69-95, 156-227
#pragma once
#include "sum_order.hpp"

#include <hip/hip_runtime.h>
#include <stdint.h>

namespace ptx
{

template <class R> struct V3
{
    R x, y, z;
};

template <class R> __device__ __forceinline__ V3<R> mk(R x, R y, R z) { return V3<R>{x, y, z}; }
template <class R> __device__ __forceinline__ V3<R> operator+(V3<R> a, V3<R> b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
template <class R> __device__ __forceinline__ V3<R> operator-(V3<R> a, V3<R> b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
template <class R> __device__ __forceinline__ V3<R> operator-(V3<R> a) { return {-a.x, -a.y, -a.z}; }
template <class R> __device__ __forceinline__ V3<R> operator*(V3<R> a, R s) { return {a.x * s, a.y * s, a.z * s}; }
template <class R> __device__ __forceinline__ V3<R> operator*(R s, V3<R> a) { return {s * a.x, s * a.y, s * a.z}; }
template <class R> __device__ __forceinline__ V3<R> operator/(V3<R> a, R s) { return {a.x / s, a.y / s, a.z / s}; }
template <class R> __device__ __forceinline__ R dot(V3<R> a, V3<R> b)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    return a.x * b.x + a.y * b.y + a.z * b.z;
}
template <class R> __device__ __forceinline__ V3<R> cross(V3<R> a, V3<R> b)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
template <class R> __device__ __forceinline__ R norm(V3<R> a) { return sqrt(PTX_SQ4(a.x * a.x, a.y * a.y, a.z * a.z)); }
// Eigen's normalized(): divide by the norm when the squared norm is > 0, else leave unchanged
template <class R> __device__ __forceinline__ V3<R> normalized(V3<R> a)
{
    R z = PTX_SQ4(a.x * a.x, a.y * a.y, a.z * a.z);
    if (z > R(0))
    {
        R n = sqrt(z);
        return {a.x / n, a.y / n, a.z / n};
    }
    return a;
}
template <class R> __device__ __forceinline__ bool has_nan(V3<R> a) { return isnan(a.x) || isnan(a.y) || isnan(a.z); }

// Shading-side helpers.  The intersector keeps the reference's exact operation order (its accept /
// reject decisions and distances are compared bit for bit with the CPU restatement; this
// translation unit is built with -ffp-contract=off).  The shading math uses algebraically
// identical but cheaper forms (reciprocal square roots, no inverse trigonometry) and, function by
// function, allows fused multiply-adds (#pragma clang fp contract(fast)), which moves results by a
// few ulps only.  The vector operators, norm, normalized, tri_test and square_light_point stay
// unfused: they feed the intersector.
__device__ __forceinline__ double rsqrt_(double x) { return rsqrt(x); }
__device__ __forceinline__ float rsqrt_(float x) { return rsqrtf(x); }
__device__ __forceinline__ void sincospi_(double x, double &s, double &c) { sincospi(x, &s, &c); }
__device__ __forceinline__ void sincospi_(float x, float &s, float &c) { sincospif(x, &s, &c); }
template <class R> __device__ __forceinline__ V3<R> fast_normalized(V3<R> a)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    R z = a.x * a.x + a.y * a.y + a.z * a.z;
    R k = z > R(0) ? rsqrt_(z) : R(1);
    return {a.x * k, a.y * k, a.z * k};
}

template <class R> struct Consts;
template <> struct Consts<double>
{
    static constexpr double pi = 3.14159265358979323846;
    static constexpr double tri_eps = 1e-10;  // barycentric / t slack of the reference test
    static constexpr double self_eps = 1e-6;  // hits closer than this are ignored
    static constexpr double big = 1.7976931348623157e308;
};
template <> struct Consts<float>
{
    static constexpr float pi = 3.14159265358979323846f;
    static constexpr float tri_eps = 1e-6f;
    static constexpr float self_eps = 1e-4f;
    static constexpr float big = 3.4028234e38f;
};

// ---------------------------------------------------------------------------------------------
// counter-based RNG: a draw is a pure function of (seed, pixel, sample, bounce, dimension)
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ uint64_t mix64(uint64_t z)
{
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}
__host__ __device__ __forceinline__ uint64_t rng_pixel_key(uint64_t seed, uint32_t pixel)
{
    return mix64(seed + 0x9E3779B97F4A7C15ULL * (uint64_t(pixel) + 1));
}
// counter word of one draw: sample in the high half; below it the dimension's high byte (bits
// 24-31), the bounce (bits 8-23, at most 65535) and the dimension's low byte -- distinct for every
// (sample, bounce < 2^16, dimension < 2^16), i.e. for up to 16 383 lights
__host__ __device__ __forceinline__ uint64_t rng_counter(uint32_t s, uint32_t b, uint32_t d)
{
    return (uint64_t(s) << 32) | (uint64_t(d >> 8) << 24) | (uint64_t(b) << 8) | uint64_t(d & 255u);
}
template <class R> __device__ __forceinline__ R rng_draw(uint64_t pixel_key, uint32_t sample, uint32_t bounce, uint32_t dim);
template <> __device__ __forceinline__ double rng_draw<double>(uint64_t k, uint32_t s, uint32_t b, uint32_t d)
{
    uint64_t h = mix64(k ^ rng_counter(s, b, d));
    return double(h >> 11) * (1.0 / 9007199254740992.0);
}
template <> __device__ __forceinline__ float rng_draw<float>(uint64_t k, uint32_t s, uint32_t b, uint32_t d)
{
    uint64_t h = mix64(k ^ rng_counter(s, b, d));
    return float(h >> 40) * (1.0f / 16777216.0f);
}

// ---------------------------------------------------------------------------------------------
// ray / triangle.  T points at the 12 precomputed scalars of one triangle:
//   p1, a = p1 - p2, b = p1 - p3, n = (a.y b.z - b.y a.z, a.z b.x - b.z a.x, a.x b.y - b.x a.y)
// (columns 0 and 1 of the reference's 3x3 system and the three cofactors that do not involve the
// ray).  Returns true when the reference accepts the hit and writes t.
// ---------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ bool tri_test(const R *__restrict__ T, V3<R> o, V3<R> d, R &t_out)
{
    const R eps = Consts<R>::tri_eps;
    const R ax = T[3], ay = T[4], az = T[5];
    const R bx = T[6], by = T[7], bz = T[8];
    const R rx = T[0] - o.x, ry = T[1] - o.y, rz = T[2] - o.z;
    const R c00 = by * d.z - d.y * bz;
    const R c10 = bz * d.x - d.z * bx;
    const R c20 = bx * d.y - d.x * by;
    const R det = PTX_SUM3(c00 * ax, c10 * ay, c20 * az);
    const R inv = R(1) / det;
    const R beta = PTX_SUM3((c00 * inv) * rx, (c10 * inv) * ry, (c20 * inv) * rz);
    const R i10 = (d.y * az - ay * d.z) * inv;
    const R i11 = (d.z * ax - az * d.x) * inv;
    const R i12 = (d.x * ay - ax * d.y) * inv;
    const R gamma = PTX_SUM3(i10 * rx, i11 * ry, i12 * rz);
    const R t = PTX_SUM3((T[9] * inv) * rx, (T[10] * inv) * ry, (T[11] * inv) * rz);
    const R alpha = R(1) - beta - gamma;
    t_out = t;
    return (R(0) - eps < alpha) && (alpha < R(1) + eps) && (R(0) - eps < beta) && (beta < R(1) + eps) &&
           (R(0) - eps < gamma) && (gamma < R(1) + eps) && (t > R(0) - eps);
}

// ---------------------------------------------------------------------------------------------
// geometry helpers
// ---------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ V3<R> reflect(V3<R> n, V3<R> in)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    R k = R(2) * dot(n, -in);
    return k * n + in;
}

// NaN components on total internal reflection
template <class R> __device__ __forceinline__ V3<R> refract(V3<R> n, V3<R> in, R ni)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    if (dot(n, in) > R(0))
    {
        n = -n;
        ni = R(1) / ni;
    }
    V3<R> nxi = cross(n, in);
    V3<R> c = cross(n, -nxi);
    R s = sqrt(R(1) - ni * ni * dot(nxi, nxi));
    return ni * c - n * s;
}

// 3x3 inverse times vector, cofactors of column 0 / determinant along column 0 (row-major m).
// The matrix holds absolute vertex positions (cGeoUtil.cpp:174-231), so a triangle whose plane passes
// near the origin makes the cofactors cancel.  In double precision the solve is therefore NOT
// contracted and summed in the order of the other short sums: a fused multiply-add moves the weights
// of such a triangle by 1e-9 of their size, enough to push a pixel past the f64 parity bar (12 of 8.3
// million at 3840 x 2160 on room.obj).  Single precision keeps its fused form: there the unfused
// cofactors can cancel to a determinant of exactly zero, and the weights become NaN.
template <class R> __device__ __forceinline__ V3<R> solve3_fused(const R *m, V3<R> p)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    R c00 = m[4] * m[8] - m[5] * m[7];
    R c10 = m[7] * m[2] - m[8] * m[1];
    R c20 = m[1] * m[5] - m[2] * m[4];
    R det = c00 * m[0] + c10 * m[3] + c20 * m[6];
    R id = R(1) / det;
    R i0 = c00 * id, i1 = c10 * id, i2 = c20 * id;
    R i3 = (m[5] * m[6] - m[3] * m[8]) * id;
    R i4 = (m[8] * m[0] - m[6] * m[2]) * id;
    R i5 = (m[2] * m[3] - m[0] * m[5]) * id;
    R i6 = (m[3] * m[7] - m[4] * m[6]) * id;
    R i7 = (m[6] * m[1] - m[7] * m[0]) * id;
    R i8 = (m[0] * m[4] - m[1] * m[3]) * id;
    return {i0 * p.x + i1 * p.y + i2 * p.z, i3 * p.x + i4 * p.y + i5 * p.z, i6 * p.x + i7 * p.y + i8 * p.z};
}
template <class R> __device__ __forceinline__ V3<R> solve3_exact(const R *m, V3<R> p)
{
#pragma clang fp contract(off)
    R c00 = m[4] * m[8] - m[5] * m[7];
    R c10 = m[7] * m[2] - m[8] * m[1];
    R c20 = m[1] * m[5] - m[2] * m[4];
    R det = PTX_SUM3(c00 * m[0], c10 * m[3], c20 * m[6]);
    R id = R(1) / det;
    R i0 = c00 * id, i1 = c10 * id, i2 = c20 * id;
    R i3 = (m[5] * m[6] - m[3] * m[8]) * id;
    R i4 = (m[8] * m[0] - m[6] * m[2]) * id;
    R i5 = (m[2] * m[3] - m[0] * m[5]) * id;
    R i6 = (m[3] * m[7] - m[4] * m[6]) * id;
    R i7 = (m[6] * m[1] - m[7] * m[0]) * id;
    R i8 = (m[0] * m[4] - m[1] * m[3]) * id;
    return {PTX_SUM3(i0 * p.x, i1 * p.y, i2 * p.z), PTX_SUM3(i3 * p.x, i4 * p.y, i5 * p.z), PTX_SUM3(i6 * p.x, i7 * p.y, i8 * p.z)};
}
__device__ __forceinline__ V3<double> solve3(const double *m, V3<double> p) { return solve3_exact(m, p); }
__device__ __forceinline__ V3<float> solve3(const float *m, V3<float> p) { return solve3_fused(m, p); }

// barycentric weights of p w.r.t. (a,b,c): [a b c]^-1 p with all-zero rows regularised.  The
// reference normalises the weight vector; the caller normalises the interpolated normal anyway, so
// that positive scale is dropped here.
template <class R> __device__ __forceinline__ V3<R> barycentric(V3<R> p, V3<R> a, V3<R> b, V3<R> c)
{
    R m[9] = {a.x, b.x, c.x, a.y, b.y, c.y, a.z, b.z, c.z};
    R pv[3] = {p.x, p.y, p.z};
#pragma unroll
    for (int i = 0; i < 3; i++)
    {
        R rn2 = m[i * 3] * m[i * 3] + m[i * 3 + 1] * m[i * 3 + 1] + m[i * 3 + 2] * m[i * 3 + 2];
        if (rn2 < R(1e-20))
        {
            m[i * 3] += R(1e-6), m[i * 3 + 1] += R(1e-6), m[i * 3 + 2] += R(1e-6);
            pv[i] += R(1e-6);
        }
    }
    return solve3(m, V3<R>{pv[0], pv[1], pv[2]});
}

// direction about `n`.  The reference draws an elevation theta = acos(1 - xi1) and an azimuth
// phi = 2 pi xi2, builds (cos theta cos phi, sin theta, -cos theta sin phi) about +Y and rotates +Y
// onto n with the quaternion of axis Y x (n - Y) and angle acos(n.y).  Same map here without inverse
// trigonometry: cos theta = 1 - xi1, sin theta = sqrt(xi1 (2 - xi1)), and the rotation in Rodrigues
// form with cos = n.y, sin = |(n.x, n.z)|, axis (n.z, 0, -n.x) / sin.
// v, given about +Y, in the frame whose +Y is the unit vector n (the rotation about Y x n)
template <class R> __device__ __forceinline__ V3<R> rotate_from_y(V3<R> n, V3<R> v);

template <class R> __device__ __forceinline__ V3<R> sample_hemisphere(V3<R> n, R xi1, R xi2)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    R ct = R(1) - xi1;
    R st = sqrt(xi1 * (R(2) - xi1));
    R sp, cp;
    sincospi_(R(2) * xi2, sp, cp);
    V3<R> v = {ct * cp, st, -ct * sp};
    return rotate_from_y(n, v);
}

template <class R> __device__ __forceinline__ V3<R> rotate_from_y(V3<R> n, V3<R> v)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    R s2 = n.x * n.x + n.z * n.z;
    if (s2 > R(0))
    {
        R inv = rsqrt_(s2);
        R kx = n.z * inv, kz = -n.x * inv;
        R sn = s2 * inv, cs = n.y;
        R kdv = (kx * v.x + kz * v.z) * (R(1) - cs);
        return {v.x * cs - kz * v.y * sn + kx * kdv, v.y * cs + (kz * v.x - kx * v.z) * sn, v.z * cs + kx * v.y * sn + kz * kdv};
    }
    // n = +Y: identity; n = -Y: half turn about X
    if (n.y < R(0))
        return {v.x, -v.y, -v.z};
    return v;
}

// ---------------------------------------------------------------------------------------------
// tables in device memory
// ---------------------------------------------------------------------------------------------
// material: 16 scalars  Ka[3] Kd[3] Ks[3] Ni Ns R0 is_bsdf is_emissive pad pad
constexpr int MAT_STRIDE = 16;
// light: 24 scalars  type Li[3] | square: P0[3] e10[3] e21[3] N[3] area | sphere: C[3] radius
constexpr int LIGHT_STRIDE = 24;

// kept out of line: the f64 pow is long and only glossy materials (Ns != 1) reach it
__device__ __attribute__((noinline)) double pow_outlined(double x, double y) { return pow(x, y); }
__device__ __attribute__((noinline)) float pow_outlined(float x, float y) { return powf(x, y); }

// x^n for a whole-number exponent by repeated squaring (|x| <= 1 here: a cosine): a few dozen
// roundings at most instead of pow's one, at a tenth of its instructions
template <class R> __device__ __forceinline__ R pow_whole(R x, R n)
{
    uint32_t e = uint32_t(n);
    R r = R(1), b = x;
    while (e)
    {
        if (e & 1u)
            r *= b;
        b *= b;
        e >>= 1;
    }
    return r;
}

template <class R> __device__ __forceinline__ V3<R> brdf_eval(const R *__restrict__ m, V3<R> wi, V3<R> wo, V3<R> n)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    const R pi = Consts<R>::pi;
    R ca = dot(reflect(n, wi), wo);
    R ns = m[10];
    // m[14]: Ns is a whole number (the host checked): Ns = 1, the default of a material without
    // a Phong lobe, is the common case
    R pw = (ns == R(1)) ? ca : (m[14] != R(0) ? pow_whole(ca, ns) : pow_outlined(ca, ns));
    const R ipi = R(1) / pi;
    R k = (ns + R(2)) * (R(0.5) * ipi) * pw;
    return {m[3] * ipi + m[6] * k, m[4] * ipi + m[7] * k, m[5] * ipi + m[8] * k};
}

template <class R> __device__ __forceinline__ V3<R> bsdf_eval(const R *__restrict__ m, V3<R> wi, V3<R> wo, V3<R> n)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    V3<R> ideal = refract(n, wi, R(1) / m[9]);
    if (has_nan(ideal))
        ideal = reflect(n, wi);
    R c = dot(ideal, wo);
    if (c < R(0))
        c = R(0);
    R v = R(1) / (R(4) * Consts<R>::pi) * c;
    return {v, v, v};
}

template <class R> __device__ __forceinline__ V3<R> bxdf_eval(const R *__restrict__ m, V3<R> wi, V3<R> wo, V3<R> n)
{
    return (m[12] != R(0)) ? bsdf_eval(m, wi, wo, n) : brdf_eval(m, wi, wo, n);
}

// next direction + indirect coefficient.  xi1, xi2: the two draws of this bounce (glass uses xi1)
template <class R>
__device__ __forceinline__ void bxdf_sample(const R *__restrict__ m, V3<R> n, V3<R> wo, R xi1, R xi2, V3<R> &wi, V3<R> &coef)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    const R pi = Consts<R>::pi;
    if (m[12] == R(0))
    {
        wi = sample_hemisphere(n, xi1, xi2);
        V3<R> f = brdf_eval(m, -wi, wo, n);
        R c = dot(n, wi) * (R(2) * pi); // (n.wi) / pdf with pdf = 1 / (2 pi)
        coef = {f.x * c, f.y * c, f.z * c};
    }
    else
    {
        R R0 = m[11];
        V3<R> mwo = -wo;
        R x = R(1) - fabs(dot(n, mwo)), x2 = x * x;
        R Rf = R0 + (R(1) - R0) * (x2 * x2 * x);
        if (xi1 < Rf)
            wi = reflect(n, mwo);
        else
        {
            wi = refract(n, mwo, R(1) / R(1.5));
            if (has_nan(wi))
                wi = reflect(n, mwo);
        }
        coef = {R(1), R(1), R(1)};
    }
}

// square light: uniform point on the parallelogram, one-sided; returns pdf (0 = skip)
template <class R> __device__ __forceinline__ V3<R> square_light_point(const R *__restrict__ L, R xi1, R xi2)
{
    V3<R> P0 = {L[4], L[5], L[6]}, e10 = {L[7], L[8], L[9]}, e21 = {L[10], L[11], L[12]};
    V3<R> s = P0 + e10 * xi1;
    return s + e21 * xi2;
}
template <class R>
__device__ __forceinline__ R square_light_sample(const R *__restrict__ L, V3<R> ref, R xi1, R xi2, V3<R> &ro, V3<R> &rd)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    V3<R> N = {L[13], L[14], L[15]};
    R area = L[16];
    V3<R> s = square_light_point(L, xi1, xi2);
    V3<R> g = ref - s;
    R z = g.x * g.x + g.y * g.y + g.z * g.z; // squared distance, also the normalisation of the direction
    R k = z > R(0) ? rsqrt_(z) : R(1);
    V3<R> dir = {g.x * k, g.y * k, g.z * k};
    R c = dot(dir, N);
    if (c <= R(0))
        return R(0);
    ro = s;
    rd = dir;
    return z / (area * c) * (R(1) / (R(2) * Consts<R>::pi));
}

// sphere light: the returned ray starts at the shaded point itself (so it is never shadowed) and
// points from the centre towards the point; only the cone pdf survives.  Inside the sphere a
// uniform surface point is used (3 draws).
// The three draws (dimensions 4 l, 4 l + 1, 4 l + 2 of the bounce) are only made inside the sphere:
// outside it the sample is a function of the geometry alone, and a draw is two 64-bit multiplies.
template <class R>
__device__ __forceinline__ R sphere_light_sample(const R *__restrict__ L, V3<R> ref, uint64_t key, uint32_t sample, uint32_t bounce, int l, V3<R> &ro,
                                                 V3<R> &rd)
{
#pragma clang fp contract(fast) // shading side: fused multiply-adds allowed
    const R pi = Consts<R>::pi;
    V3<R> C = {L[4], L[5], L[6]};
    R rad = L[7];
    R dc = norm(ref - C);
    ro = ref;
    if (dc > rad)
    {
        // 1 - cos(asin(x)) = x^2 / (1 + sqrt(1 - x^2)), without the two transcendental calls
        const R x = rad / dc, x2 = x * x;
        V3<R> to_light = fast_normalized(C - ref);
        rd = -to_light;
        return (R(1) + sqrt(R(1) - x2)) / (R(2) * pi * x2);
    }
    const R xi1 = rng_draw<R>(key, sample, bounce, 4 * l + 0), xi2 = rng_draw<R>(key, sample, bounce, 4 * l + 1);
    const R xi3 = rng_draw<R>(key, sample, bounce, 4 * l + 2);
    V3<R> u = fast_normalized(V3<R>{xi1 - R(0.5), xi2 - R(0.5), xi3 - R(0.5)});
    V3<R> lp = u * rad + C;
    rd = fast_normalized(ref - lp);
    return R(1) / (R(4) * pi);
}

// ---------------------------------------------------------------------------------------------
// The physically based estimator ("fast").  No counterpart in the reference (its estimator has no
// sampling weights, no roulette and never shadows sphere lights, src/tracing/cBxDF.cpp:16-52,
// src/tracing/cLight.cpp:216): BRDF f = Kd / pi + Ks (Ns + 2) / (2 pi) max(0, r.wi)^Ns sampled by
// lobe (cosine hemisphere with probability pd = max Kd / (max Kd + max Ks), else the Phong lobe
// about the mirror direction), glass as a delta lobe with the material's own Ni, every conf light
// sampled with a visibility test and the balance heuristic against the BRDF's density, conf lights
// as emitters that do not occlude for BRDF-sampled rays, Russian roulette from bounce 3 on.
// Directions here are unit vectors pointing AWAY from the surface.
// ---------------------------------------------------------------------------------------------
template <class R> __device__ __forceinline__ R max3(R a, R b, R c) { return fmax(fmax(a, b), c); }

// density (solid angle, from o along unit d) with which light L is sampled towards the point where
// the ray (o, d) crosses it, and the distance to that point; 0 when the ray misses the light or
// meets a square light from behind
template <class R> __device__ __forceinline__ R fast_light_pdf(const R *__restrict__ L, V3<R> o, V3<R> d, R &t_out)
{
#pragma clang fp contract(fast)
    const R pi = Consts<R>::pi;
    t_out = Consts<R>::big;
    if (L[0] == R(0))
    {
        const V3<R> P0 = {L[4], L[5], L[6]}, e10 = {L[7], L[8], L[9]}, e21 = {L[10], L[11], L[12]}, N = {L[13], L[14], L[15]};
        const R dn = dot(d, N);
        if (!(dn < R(0)))
            return R(0);
        const R t = dot(P0 - o, N) / dn;
        if (!(t > Consts<R>::self_eps))
            return R(0);
        const V3<R> h = o + t * d - P0;
        const R u = dot(h, e10) / dot(e10, e10), v = dot(h, e21) / dot(e21, e21);
        if (u < R(0) || u > R(1) || v < R(0) || v > R(1))
            return R(0);
        t_out = t;
        return t * t / (L[16] * -dn);
    }
    const V3<R> g = V3<R>{L[4], L[5], L[6]} - o;
    const R rad = L[7];
    const R dc2 = dot(g, g), b = dot(g, d), disc = b * b - dc2 + rad * rad;
    if (!(disc >= R(0)))
        return R(0);
    const R sq = sqrt(disc);
    if (dc2 > rad * rad)
    {
        const R t = b - sq;
        if (!(t > Consts<R>::self_eps))
            return R(0);
        t_out = t;
        return R(1) / (R(2) * pi * (R(1) - sqrt(R(1) - rad * rad / dc2)));
    }
    t_out = b + sq;
    return R(1) / (R(4) * pi);
}

template <class R> __device__ __forceinline__ R fast_sphere_light_sample(const R *__restrict__ L, V3<R> p, R xi1, R xi2, V3<R> &wi, V3<R> &q);

// one light sample from the shaded point p: unit direction wi towards the light, the point q on
// the light, the density of wi (0: the sample is of no use)
template <class R> __device__ __forceinline__ R fast_light_sample(const R *__restrict__ L, V3<R> p, R xi1, R xi2, V3<R> &wi, V3<R> &q)
{
    if (L[0] == R(0))
    {
        q = square_light_point(L, xi1, xi2); // unfused: a visibility test starts from this point
        {
#pragma clang fp contract(fast)
            const V3<R> N = {L[13], L[14], L[15]};
            const V3<R> w = q - p;
            const R d2 = dot(w, w);
            if (!(d2 > R(0)))
                return R(0);
            const R inv = rsqrt_(d2);
            wi = {w.x * inv, w.y * inv, w.z * inv};
            const R cl = -dot(wi, N);
            if (!(cl > R(0)))
                return R(0);
            return d2 / (L[16] * cl);
        }
    }
    return fast_sphere_light_sample(L, p, xi1, xi2, wi, q);
}

template <class R> __device__ __forceinline__ R fast_sphere_light_sample(const R *__restrict__ L, V3<R> p, R xi1, R xi2, V3<R> &wi, V3<R> &q)
{
#pragma clang fp contract(fast)
    const R pi = Consts<R>::pi;
    const V3<R> g = V3<R>{L[4], L[5], L[6]} - p;
    const R rad = L[7];
    const R dc2 = dot(g, g);
    R sp, cp;
    sincospi_(R(2) * xi2, sp, cp);
    if (dc2 > rad * rad)
    {
        const R idc = rsqrt_(dc2), dc = dc2 * idc;
        const V3<R> axis = {g.x * idc, g.y * idc, g.z * idc};
        const R s2max = rad * rad / dc2, cmax = sqrt(R(1) - s2max);
        const R ct = R(1) - xi1 * (R(1) - cmax), st2 = R(1) - ct * ct, st = sqrt(fmax(st2, R(0)));
        wi = rotate_from_y(axis, V3<R>{st * cp, ct, -st * sp});
        const R t = dc * ct - sqrt(fmax(rad * rad - dc2 * st2, R(0)));
        q = {p.x + t * wi.x, p.y + t * wi.y, p.z + t * wi.z};
        return R(1) / (R(2) * pi * (R(1) - cmax));
    }
    const R ct = R(1) - R(2) * xi1, st = sqrt(fmax(R(1) - ct * ct, R(0)));
    wi = {st * cp, ct, -st * sp};
    const R b = dot(g, wi), t = b + sqrt(b * b - dc2 + rad * rad);
    q = {p.x + t * wi.x, p.y + t * wi.y, p.z + t * wi.z};
    return R(1) / (R(4) * pi);
}

// BRDF value and sampling density of wi for the viewer direction wo about the normal n (n.wo >= 0)
template <class R> __device__ __forceinline__ V3<R> fast_brdf(const R *__restrict__ m, V3<R> n, V3<R> wo, V3<R> wi, R &pdf)
{
#pragma clang fp contract(fast)
    const R pi = Consts<R>::pi;
    const R cn = dot(n, wi);
    pdf = R(0);
    if (!(cn > R(0)))
        return V3<R>{R(0), R(0), R(0)};
    const R k = R(2) * dot(n, wo);
    const V3<R> r = {k * n.x - wo.x, k * n.y - wo.y, k * n.z - wo.z};
    const R ca = dot(r, wi), ns = m[10];
    R pw = R(0);
    if (ca > R(0))
        pw = (ns == R(1)) ? ca : (m[14] != R(0) ? pow_whole(ca, ns) : pow_outlined(ca, ns));
    const R kd = max3(m[3], m[4], m[5]), ks = max3(m[6], m[7], m[8]);
    const R ipi = R(1) / pi;
    const R kk = (ns + R(2)) * (R(0.5) * ipi) * pw;
    if (kd + ks > R(0))
    {
        const R pd = kd / (kd + ks);
        pdf = pd * cn * ipi + (R(1) - pd) * ((ns + R(1)) * (R(0.5) * ipi)) * pw;
    }
    return V3<R>{m[3] * ipi + m[6] * kk, m[4] * ipi + m[7] * kk, m[5] * ipi + m[8] * kk};
}

// next direction of a BRDF surface from three draws; false: the path ends here
template <class R> __device__ __forceinline__ bool fast_brdf_sample(const R *__restrict__ m, V3<R> n, V3<R> wo, R x0, R x1, R x2, V3<R> &wi)
{
#pragma clang fp contract(fast)
    const R kd = max3(m[3], m[4], m[5]), ks = max3(m[6], m[7], m[8]), ns = m[10];
    if (!(kd + ks > R(0)))
        return false;
    const R pd = kd / (kd + ks);
    R sp, cp;
    sincospi_(R(2) * x2, sp, cp);
    if (x0 < pd)
    {
        const R r = sqrt(x1), h = sqrt(R(1) - x1);
        wi = rotate_from_y(n, V3<R>{r * cp, h, -r * sp});
    }
    else
    {
        const R ca = pow_outlined(x1, R(1) / (ns + R(1))), sa = sqrt(fmax(R(1) - ca * ca, R(0)));
        const R k = R(2) * dot(n, wo);
        const V3<R> r = {k * n.x - wo.x, k * n.y - wo.y, k * n.z - wo.z};
        wi = rotate_from_y(r, V3<R>{sa * cp, ca, -sa * sp});
    }
    return dot(n, wi) > R(0);
}

// glass: reflect or refract, chosen by Schlick's reflectance against one draw
template <class R> __device__ __forceinline__ V3<R> fast_glass_sample(const R *__restrict__ m, V3<R> n, V3<R> wo, R xi)
{
#pragma clang fp contract(fast)
    const R ni = m[9];
    R ci = dot(n, wo), eta = R(1) / ni;
    if (ci < R(0))
    {
        n = -n;
        ci = -ci, eta = ni;
    }
    const R s2t = eta * eta * (R(1) - ci * ci);
    bool refl = true;
    R ct = R(0);
    if (s2t < R(1))
    {
        ct = sqrt(R(1) - s2t);
        const R c = eta > R(1) ? ct : ci, x = R(1) - c, x2 = x * x;
        const R Rf = m[11] + (R(1) - m[11]) * (x2 * x2 * x);
        refl = xi < Rf;
    }
    const R a = refl ? R(2) * ci : eta * ci - ct, b = refl ? R(1) : eta;
    return V3<R>{a * n.x - b * wo.x, a * n.y - b * wo.y, a * n.z - b * wo.z};
}

} // namespace ptx
