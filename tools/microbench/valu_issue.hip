// Micro-benchmark: issue cost of the vector-ALU instruction classes the tracer kernels are made of,
// in shader cycles per wave64 instruction and SIMD.  One workgroup on one compute unit; W waves per
// SIMD run the same loop of 64 independent instructions of one opcode (eight accumulators, so a
// result is needed again only eight instructions later); the loop is timed with s_memtime.  At one
// wave per SIMD the figure is max(issue, latency / 8); from two waves on it is the issue cost alone:
// the cost the issue-cycle accounting of tools/issue_cycles.py multiplies the hardware's per-class
// instruction counters with.
// usage: valu_issue [iterations]          prints one JSON document
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

// eight instructions, accumulator k both destination and last source
#define I8(OP, A, B)                                    \
    OP " %0, " A B "%0\n" OP " %1, " A B "%1\n"         \
    OP " %2, " A B "%2\n" OP " %3, " A B "%3\n"         \
    OP " %4, " A B "%4\n" OP " %5, " A B "%5\n"         \
    OP " %6, " A B "%6\n" OP " %7, " A B "%7\n"
#define I64(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B) I8(OP, A, B)

typedef unsigned long long u64;
// T: type of the accumulators (what the destination register class is), S: of the two sources
#define BENCH(NAME, T, S, ASM)                                                                       \
    __global__ void __launch_bounds__(1024) NAME(unsigned long long *out, int iters, S s0, S s1)     \
    {                                                                                                \
        T a0 = T(threadIdx.x + 1), a1 = a0 + T(1), a2 = a0 + T(2), a3 = a0 + T(3), a4 = a0 + T(4),   \
          a5 = a0 + T(5), a6 = a0 + T(6), a7 = a0 + T(7);                                            \
        S x = s0 + S(threadIdx.x & 3), y = s1;                                                       \
        __syncthreads();                                                                             \
        const unsigned long long t0 = __builtin_readcyclecounter();                                  \
        for (int i = 0; i < iters; i++)                                                              \
            asm volatile(ASM : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), \
                         "+v"(a7) : "v"(x), "v"(y) : "vcc", "scc", "s10", "s11", "v60", "v61");                                         \
        const unsigned long long t1 = __builtin_readcyclecounter();                                  \
        if ((threadIdx.x & 63) == 0)                                                                 \
            out[threadIdx.x >> 6] = t1 - t0;                                                         \
        if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == T(123456789))                                   \
            out[100] = 1;                                                                            \
    }

BENCH(k_add_f32, float, float, I64("v_add_f32", "%8, ", ""))
BENCH(k_mul_f32, float, float, I64("v_mul_f32", "%8, ", ""))
BENCH(k_fma_f32, float, float, I64("v_fma_f32", "%8, ", "%9, "))
BENCH(k_rcp_f32, float, float, I64("v_rcp_f32", "", ""))
BENCH(k_rsq_f32, float, float, I64("v_rsq_f32", "", ""))
BENCH(k_sqrt_f32, float, float, I64("v_sqrt_f32", "", ""))
BENCH(k_sin_f32, float, float, I64("v_sin_f32", "", ""))
BENCH(k_exp_f32, float, float, I64("v_exp_f32", "", ""))
BENCH(k_add_f64, double, double, I64("v_add_f64", "%8, ", ""))
BENCH(k_mul_f64, double, double, I64("v_mul_f64", "%8, ", ""))
BENCH(k_fma_f64, double, double, I64("v_fma_f64", "%8, ", "%9, "))
BENCH(k_min_f64, double, double, I64("v_min_f64", "%8, ", ""))
BENCH(k_rcp_f64, double, double, I64("v_rcp_f64", "", ""))
BENCH(k_rsq_f64, double, double, I64("v_rsq_f64", "", ""))
BENCH(k_sqrt_f64, double, double, I64("v_sqrt_f64", "", ""))
BENCH(k_ldexp_f64, double, int, I64("v_ldexp_f64", "", "%0, %8 ;"))
BENCH(k_div_scale_f64, double, double, I64("v_div_scale_f64", "vcc, %8, %9, ", ""))
BENCH(k_div_fmas_f64, double, double, I64("v_div_fmas_f64", "%8, %9, ", ""))
BENCH(k_div_fixup_f64, double, double, I64("v_div_fixup_f64", "%8, %9, ", ""))
BENCH(k_trig_preop_f64, double, int, I64("v_trig_preop_f64", "", "%0, %8 ;"))
BENCH(k_fract_f64, double, double, I64("v_fract_f64", "", ""))
BENCH(k_cmp_lt_f64, double, double, I64("v_cmp_lt_f64 vcc,", "", "%8 ;")) // writes vcc; the accumulator is a source
BENCH(k_cmp_lt_f32, float, float, I64("v_cmp_lt_f32 vcc,", "", "%8 ;"))
BENCH(k_cmp_lt_u32, unsigned, unsigned, I64("v_cmp_lt_u32 vcc,", "", "%8 ;"))
BENCH(k_cvt_f64_f32, double, float, I64("v_cvt_f64_f32", "%8 ;", ""))
BENCH(k_cvt_f32_f64, float, double, I64("v_cvt_f32_f64", "%8 ;", ""))
BENCH(k_mov_b32, unsigned, unsigned, I64("v_mov_b32", "%8 ;", ""))
BENCH(k_mov_b64, double, double, I64("v_mov_b64", "%8 ;", ""))
BENCH(k_cndmask_b32, unsigned, unsigned, I64("v_cndmask_b32", "%8, %9, vcc ;", ""))
BENCH(k_cndmask_e64_sgpr, unsigned, unsigned, I64("v_cndmask_b32_e64", "%8, %9, s[10:11] ;", ""))
// a compare and the select that consumes it, as compiled code pairs them: two instructions per line
BENCH(k_cmp_cndmask_pair, unsigned, unsigned, I64("v_cmp_lt_u32 vcc, %8, %9\nv_cndmask_b32", "%8, %9, vcc ;", ""))
// selects between instructions that do not touch the mask: add, select, add, select
BENCH(k_add_cndmask_pair, unsigned, unsigned, I64("v_add_u32 %0, %8, %0\nv_cndmask_b32", "%8, %9, vcc ;", ""))
BENCH(k_readlane, unsigned, unsigned, I64("v_readlane_b32 s10,", "", "3 ;"))
BENCH(k_readfirstlane, unsigned, unsigned, I64("v_readfirstlane_b32 s10,", "", ";"))
BENCH(k_writelane, unsigned, unsigned, I64("v_writelane_b32", "s10, 3 ;", ""))
BENCH(k_mbcnt, unsigned, unsigned, I64("v_mbcnt_lo_u32_b32", "-1, ", ""))
BENCH(k_and_b32, unsigned, unsigned, I64("v_and_b32", "%8, ", ""))
BENCH(k_lshlrev_b32, unsigned, unsigned, I64("v_lshlrev_b32", "3, ", ""))
BENCH(k_and_or_b32, unsigned, unsigned, I64("v_and_or_b32", "%8, %9, ", ""))
BENCH(k_sub_f64_neg, double, double, I64("v_add_f64", "%8, -", ""))
// mixes: do instructions of different cost classes share the issue port or overlap?  two per line
BENCH(k_mix_fma_cvt, float, float, I64("v_cvt_f32_i32 %0, %0\nv_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_fma_max, float, float, I64("v_max_f32 %0, %8, %0\nv_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_mov_fma64, double, double, I64("v_mov_b32 v60, v61\nv_fma_f64", "%8, %9, ", ""))
BENCH(k_mix_fma_cmp, float, float, I64("v_cmp_lt_f32 vcc, %8, %0\nv_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_fma_rcp, float, float, I64("v_rcp_f32 %0, %0\nv_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_fma_salu, float, float, I64("s_add_u32 s10, s10, 1\nv_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_max_salu, float, float, I64("s_add_u32 s10, s10, 1\nv_max_f32", "%8, ", ""))
// packed f32: two operations per lane and instruction on 64-bit register pairs
BENCH(k_pk_fma_f32, double, double, I64("v_pk_fma_f32", "%8, %9, ", ""))
BENCH(k_pk_mul_f32, double, double, I64("v_pk_mul_f32", "%8, ", ""))
BENCH(k_pk_add_f32, double, double, I64("v_pk_add_f32", "%8, ", ""))
BENCH(k_mix_mov_pkfma, double, double, I64("v_mov_b32 v60, v61\nv_pk_fma_f32", "%8, %9, ", ""))
BENCH(k_mix_cmp_pkfma, double, double, I64("v_cmp_lt_f64 vcc, %8, %0\nv_pk_fma_f32", "%8, %9, ", ""))
BENCH(k_cvt_f32_ubyte1, float, unsigned, I64("v_cvt_f32_ubyte1", "%8 ;", ""))
BENCH(k_or_b32_sdwa_byte1, unsigned, unsigned, I64("v_or_b32_sdwa", "%8, %9 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:DWORD src1_sel:BYTE_1 ;", ""))
BENCH(k_perm_b32, unsigned, unsigned, I64("v_perm_b32", "%8, %9, ", ""))
BENCH(k_bfe_u32, unsigned, unsigned, I64("v_bfe_u32", "%8, 8, 8 ;", ""))
BENCH(k_min3_f32, float, float, I64("v_min3_f32", "%8, %9, ", ""))
BENCH(k_max_f32, float, float, I64("v_max_f32", "%8, ", ""))
BENCH(k_min_u32, unsigned, unsigned, I64("v_min_u32", "%8, ", ""))
BENCH(k_sub_f32, float, float, I64("v_sub_f32", "%8, ", ""))
BENCH(k_add_u32, unsigned, unsigned, I64("v_add_u32", "%8, ", ""))
BENCH(k_xor_b32, unsigned, unsigned, I64("v_xor_b32", "%8, ", ""))
BENCH(k_lshl_add_u32, unsigned, unsigned, I64("v_lshl_add_u32", "%8, 3, ", ""))
BENCH(k_add3_u32, unsigned, unsigned, I64("v_add3_u32", "%8, %9, ", ""))
BENCH(k_mul_lo_u32, unsigned, unsigned, I64("v_mul_lo_u32", "%8, ", ""))
BENCH(k_mul_hi_u32, unsigned, unsigned, I64("v_mul_hi_u32", "%8, ", ""))
BENCH(k_mul_u32_u24, unsigned, unsigned, I64("v_mul_u32_u24", "%8, ", ""))
BENCH(k_mad_u64_u32, u64, unsigned, I64("v_mad_u64_u32", "vcc, %8, %9, ", ""))
BENCH(k_lshlrev_b64, u64, unsigned, I64("v_lshlrev_b64", "3, ", ""))
BENCH(k_lshrrev_b64, u64, unsigned, I64("v_lshrrev_b64", "3, ", ""))
BENCH(k_lshl_add_u64, u64, u64, I64("v_lshl_add_u64", "%8, 3, ", ""))
BENCH(k_bpermute, unsigned, unsigned, I64("ds_bpermute_b32", "%8, ", "") "s_waitcnt lgkmcnt(0)\n")

struct Entry
{
    const char *name;
    void *fn;
    int kind; // 0: float sources, 1: double, 2: int / unsigned, 3: unsigned long long
};
#define E(N, K) {#N, (void *)k_##N, K}
static Entry entries[] = {
    E(add_f32, 0), E(mul_f32, 0), E(fma_f32, 0), E(rcp_f32, 0), E(rsq_f32, 0), E(sqrt_f32, 0), E(sin_f32, 0), E(exp_f32, 0),
    E(add_f64, 1), E(mul_f64, 1), E(fma_f64, 1), E(min_f64, 1), E(rcp_f64, 1), E(rsq_f64, 1), E(sqrt_f64, 1), E(ldexp_f64, 2),
    E(div_scale_f64, 1), E(div_fmas_f64, 1), E(div_fixup_f64, 1), E(trig_preop_f64, 2), E(fract_f64, 1),
    E(cmp_lt_f64, 1), E(cmp_lt_f32, 0), E(cmp_lt_u32, 2), E(cvt_f64_f32, 0), E(cvt_f32_f64, 1), E(mov_b32, 2), E(mov_b64, 1), E(cndmask_b32, 2),
    E(cndmask_e64_sgpr, 2), E(cmp_cndmask_pair, 2), E(add_cndmask_pair, 2), E(readlane, 2), E(readfirstlane, 2), E(writelane, 2), E(mbcnt, 2),
    E(mix_fma_cvt, 0), E(mix_fma_max, 0), E(mix_mov_fma64, 1), E(mix_fma_cmp, 0), E(mix_fma_rcp, 0), E(mix_fma_salu, 0), E(mix_max_salu, 0),
    E(pk_fma_f32, 1), E(pk_mul_f32, 1), E(pk_add_f32, 1), E(mix_mov_pkfma, 1), E(mix_cmp_pkfma, 1),
    E(cvt_f32_ubyte1, 2), E(or_b32_sdwa_byte1, 2), E(perm_b32, 2), E(bfe_u32, 2), E(min3_f32, 0), E(max_f32, 0), E(min_u32, 2), E(sub_f32, 0),
    E(and_b32, 2), E(lshlrev_b32, 2), E(and_or_b32, 2), E(sub_f64_neg, 1), E(add_u32, 2), E(xor_b32, 2), E(lshl_add_u32, 2), E(add3_u32, 2), E(mul_lo_u32, 2), E(mul_hi_u32, 2), E(mul_u32_u24, 2),
    E(mad_u64_u32, 2), E(lshlrev_b64, 2), E(lshrrev_b64, 2), E(lshl_add_u64, 3), E(bpermute, 2)};

int main(int argc, char **argv)
{
    const int iters = argc > 1 ? atoi(argv[1]) : 2000;
    unsigned long long *out;
    CHECK(hipMalloc(&out, 128 * sizeof(unsigned long long)));
    hipDeviceProp_t prop;
    CHECK(hipGetDeviceProperties(&prop, 0));
    printf("{\"device\": \"%s\", \"iterations\": %d, \"instructions_per_iteration\": 64,\n \"cycles_per_wave_instruction_per_simd\": {\n", prop.gcnArchName, iters);
    const int waves_per_simd[3] = {1, 2, 4};
    for (size_t e = 0; e < sizeof(entries) / sizeof(entries[0]); e++)
    {
        printf("  \"%s\": {", entries[e].name);
        for (int w = 0; w < 3; w++)
        {
            const int threads = 256 * waves_per_simd[w];
            float f0 = 1.0001f, f1 = 0.5f;
            double d0 = 1.0001, d1 = 0.5;
            int i0 = 3, i1 = 5;
            unsigned long long l0 = 3, l1 = 5;
            void *args_f[] = {&out, (void *)&iters, &f0, &f1}, *args_d[] = {&out, (void *)&iters, &d0, &d1};
            void *args_i[] = {&out, (void *)&iters, &i0, &i1}, *args_l[] = {&out, (void *)&iters, &l0, &l1};
            void **args = entries[e].kind == 0 ? args_f : entries[e].kind == 1 ? args_d : entries[e].kind == 2 ? args_i : args_l;
            for (int rep = 0; rep < 2; rep++) // the first run warms the instruction cache
                CHECK(hipLaunchKernel(entries[e].fn, dim3(1), dim3(threads), args, 0, 0));
            CHECK(hipDeviceSynchronize());
            std::vector<unsigned long long> h(16);
            CHECK(hipMemcpy(h.data(), out, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
            unsigned long long mx = 0;
            for (int k = 0; k < threads / 64; k++)
                mx = h[k] > mx ? h[k] : mx;
            // every SIMD issued waves_per_simd x iters x 64 instructions in mx cycles
            printf("%s\"%d\": %.3f", w ? ", " : "", waves_per_simd[w], double(mx) / (double(iters) * 64.0 * waves_per_simd[w]));
        }
        printf("}%s\n", e + 1 < sizeof(entries) / sizeof(entries[0]) ? "," : "");
    }
    printf(" }}\n");
    CHECK(hipFree(out));
    return 0;
}
