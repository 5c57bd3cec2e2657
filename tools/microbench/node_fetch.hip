// Micro-benchmark: how fast can a compute unit fetch 64-byte BVH node records at scattered addresses?
// Every lane walks a pseudo-random chain through a table of 64-byte records (the next index depends
// on the loaded data, as in a traversal).  Variants of the fetch:
//   0  per lane: four 16-byte loads of the lane's own record (64 lanes x 4 scattered accesses)
//   1  per quad: instruction i loads the record of quad member i, lane j of the quad its 16-byte
//      piece j (the four lanes of a quad read 64 contiguous bytes); data stays in the loading lane
//      (no transpose: measures the memory side alone)
//   2  per quad, loaded straight into LDS (global_load_lds_dwordx4), every lane then reads its own
//      record back with four 16-byte LDS reads
// usage: node_fetch <variant> [n_records] [steps] [active lanes of 64] [waves per SIMD]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                  \
    do                                                                            \
    {                                                                             \
        hipError_t e_ = (x);                                                      \
        if (e_ != hipSuccess)                                                     \
        {                                                                         \
            fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));               \
            exit(1);                                                              \
        }                                                                         \
    } while (0)

__device__ __forceinline__ uint32_t mix(uint32_t x)
{
    x ^= x >> 16, x *= 0x7feb352du, x ^= x >> 15, x *= 0x846ca68bu, x ^= x >> 16;
    return x;
}

__device__ __forceinline__ uint32_t fold(uint4 q) { return q.x ^ (q.y * 3u) ^ (q.z * 5u) ^ (q.w * 7u); }

typedef __attribute__((address_space(3))) void lds_void_t;
typedef __attribute__((address_space(1))) void global_void_t;

template <int VARIANT>
__global__ void __launch_bounds__(256) k_fetch(const uint4 *__restrict__ recs, uint32_t n, int steps, int active, uint32_t *out)
{
    __shared__ uint4 stage[4][4][64]; // [wave][instruction][lane]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint32_t cur = mix(blockIdx.x * 256u + threadIdx.x) % n;
    uint32_t acc = 0;
    const bool on = lane < active; // inactive lanes fetch nothing (variant 0) or tag along with record 0
    for (int s = 0; s < steps; s++)
    {
        uint4 a, b, c, d;
        if (VARIANT == 0)
        {
            if (on)
            {
                const uint4 *r = recs + (size_t)cur * 4;
                a = r[0], b = r[1], c = r[2], d = r[3];
            }
            else
                a = b = c = d = uint4{0, 0, 0, 0};
        }
        else
        {
            const uint32_t mine = on ? cur : 0u;
            const int j = lane & 3;
            const uint32_t c0 = uint32_t(__builtin_amdgcn_mov_dpp(int(mine), 0x00, 0xf, 0xf, true)); // quad_perm [0,0,0,0]
            const uint32_t c1 = uint32_t(__builtin_amdgcn_mov_dpp(int(mine), 0x55, 0xf, 0xf, true)); // [1,1,1,1]
            const uint32_t c2 = uint32_t(__builtin_amdgcn_mov_dpp(int(mine), 0xaa, 0xf, 0xf, true)); // [2,2,2,2]
            const uint32_t c3 = uint32_t(__builtin_amdgcn_mov_dpp(int(mine), 0xff, 0xf, 0xf, true)); // [3,3,3,3]
            if (VARIANT == 1)
            {
                a = recs[(size_t)c0 * 4 + j], b = recs[(size_t)c1 * 4 + j], c = recs[(size_t)c2 * 4 + j], d = recs[(size_t)c3 * 4 + j];
                // what a transpose would deliver: mix everything so that the loads stay
                const uint32_t t = fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
                a.x = t ^ uint32_t(__builtin_amdgcn_mov_dpp(int(t), 0xb1, 0xf, 0xf, true)) ^ uint32_t(__builtin_amdgcn_mov_dpp(int(t), 0x4e, 0xf, 0xf, true));
            }
            else
            {
                __builtin_amdgcn_global_load_lds((global_void_t *)(recs + (size_t)c0 * 4 + j), (lds_void_t *)&stage[wave][0][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void_t *)(recs + (size_t)c1 * 4 + j), (lds_void_t *)&stage[wave][1][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void_t *)(recs + (size_t)c2 * 4 + j), (lds_void_t *)&stage[wave][2][0], 16, 0, 0);
                __builtin_amdgcn_global_load_lds((global_void_t *)(recs + (size_t)c3 * 4 + j), (lds_void_t *)&stage[wave][3][0], 16, 0, 0);
                __builtin_amdgcn_s_waitcnt(0x0070 | 0xc000 | 0x0f00); // vmcnt(0): expcnt / lgkmcnt untouched
                __builtin_amdgcn_wave_barrier();
                // instruction j holds, at lanes 4k .. 4k+3, the record of lane 4k + j
                const uint4 *mineq = &stage[wave][j][lane & ~3];
                a = mineq[0], b = mineq[1], c = mineq[2], d = mineq[3];
                __builtin_amdgcn_wave_barrier();
            }
        }
        const uint32_t h = fold(a) ^ fold(b) ^ fold(c) ^ fold(d);
        acc += h;
        cur = mix(h + uint32_t(s) + cur) % n;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

int main(int argc, char **argv)
{
    const int variant = argc > 1 ? atoi(argv[1]) : 0;
    const uint32_t n = argc > 2 ? uint32_t(atoll(argv[2])) : 136000u;
    const int steps = argc > 3 ? atoi(argv[3]) : 2000;
    const int active = argc > 4 ? atoi(argv[4]) : 64;
    const int wps = argc > 5 ? atoi(argv[5]) : 3;
    std::vector<uint32_t> host((size_t)n * 16);
    uint32_t x = 12345u;
    for (auto &v : host)
        v = (x = x * 1664525u + 1013904223u);
    uint4 *recs;
    uint32_t *out;
    const int blocks = 256 * wps; // 4 waves per workgroup: wps workgroups per compute unit
    CHECK(hipMalloc(&recs, (size_t)n * 64));
    CHECK(hipMalloc(&out, (size_t)blocks * 256 * 4));
    CHECK(hipMemcpy(recs, host.data(), (size_t)n * 64, hipMemcpyHostToDevice));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0));
    CHECK(hipEventCreate(&e1));
    for (int rep = 0; rep < 3; rep++)
    {
        CHECK(hipEventRecord(e0));
        if (variant == 0)
            k_fetch<0><<<blocks, 256>>>(recs, n, steps, active, out);
        else if (variant == 1)
            k_fetch<1><<<blocks, 256>>>(recs, n, steps, active, out);
        else
            k_fetch<2><<<blocks, 256>>>(recs, n, steps, active, out);
        CHECK(hipEventRecord(e1));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        const double fetches = double(blocks) * 256.0 * (active / 64.0) * steps;
        if (rep == 2)
            printf("variant %d  records %u (%.1f MB)  active %d  waves/SIMD %d: %.3f ms, %.2f G records/s, %.0f cycles per wave step at 2.4 GHz\n",
                   variant, n, n * 64.0 / 1e6, active, wps, ms, fetches / ms / 1e6, ms * 1e-3 * 2.4e9 / steps);
    }
    return 0;
}
