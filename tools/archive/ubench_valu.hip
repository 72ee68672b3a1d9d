// ubench_valu.hip -- issue cost of the VALU instructions the frame producer is made of, on gfx950.
// Each kernel runs ITER x 32 copies of one instruction on 8 independent register sets and stamps
// s_memtime around the loop; cycles per wave-instruction are reported at 1, 2 and 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o /tmp/ubench tools/archive/ubench_valu.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000
#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)

#define KERNEL(NAME, DECL, BODY, SINK)                                                                   \
    __global__ void k_##NAME(unsigned long long *out, double seed)                                       \
    {                                                                                                    \
        DECL                                                                                             \
        unsigned long long t0, t1;                                                                       \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");                        \
        for (int it = 0; it < ITER; it++)                                                                \
        {                                                                                                \
            BODY BODY BODY BODY                                                                          \
        }                                                                                                \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");                        \
        SINK                                                                                             \
        if ((threadIdx.x & 63) == 0)                                                                     \
            out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;                                 \
    }

#define D64 double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5 + 1e-3, c = 1.0000001;
#define S64 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678) out[0] = 1;
#define D32 float a0 = (float)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 0.5f + 1e-3f, c = 1.0000001f;
#define S32 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678f) out[0] = 1;
#define DI unsigned a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7, b = a0 * 3 + 1, c = 7;
#define SI if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345) out[0] = 1;

#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b));
#define X_MUL64(i) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a##i) : "v"(c));
#define X_ADD64(i) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define X_RCP64(i) asm volatile("v_rcp_f64 %0, %0" : "+v"(a##i));
#define X_RSQ64(i) asm volatile("v_rsq_f64 %0, %0" : "+v"(a##i));
#define X_DSC64(i) asm volatile("v_div_scale_f64 %0, vcc, %0, %1, %0" : "+v"(a##i) : "v"(b) : "vcc");
#define X_DFM64(i) asm volatile("v_div_fmas_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b) : "vcc");
#define X_DFX64(i) asm volatile("v_div_fixup_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b));
#define X_LDX64(i) asm volatile("v_ldexp_f64 %0, %0, 1" : "+v"(a##i));
#define X_CMP64(i) asm volatile("v_cmp_lt_f64 vcc, %0, %1" ::"v"(a##i), "v"(b) : "vcc");
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b));
#define X_FMAC32(i) asm volatile("v_fmac_f32 %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b));
#define X_CMP32(i) asm volatile("v_cmp_lt_f32 vcc, %0, %1" ::"v"(a##i), "v"(b) : "vcc");
#define X_CMP32S(i) asm volatile("v_cmp_lt_f32 s[20:21], %0, %1" ::"v"(a##i), "v"(b) : "s20", "s21");
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(a##i) : "v"(b) : "vcc");
#define X_MOV(i) asm volatile("v_mov_b32 %0, %1" : "+v"(a##i) : "v"(b));
#define X_OR3(i) asm volatile("v_or3_b32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
#define X_ADDC(i) asm volatile("v_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(a##i)::"vcc");
#define X_CVT(i) asm volatile("v_cvt_f32_f64 %0, %1" : "=v"(a##i) : "v"(bd));
#define X_PKFMA(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##i) : "v"(pc), "v"(pb));
#define X_PKMUL(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p##i) : "v"(pc));
#define X_ALIGN(i) asm volatile("v_alignbit_b32 %0, %0, %1, 31" : "+v"(a##i) : "v"(b));
#define X_SUB32(i) asm volatile("v_sub_f32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define X_CNDS(i) asm volatile("v_cndmask_b32 %0, %0, %1, s[20:21]" : "+v"(a##i) : "v"(b));
#define X_MFMA(i) asm volatile("v_mfma_f32_32x32x2_f32 %0, %1, %2, %0" : "+v"(m##i) : "v"(fa), "v"(fb));
#define X_FMACS(i) asm volatile("v_fmac_f32 %0, s20, %1" : "+v"(a##i) : "v"(b) : "s20");
#define X_FMACS2(i) asm volatile("v_fmac_f32 %0, s2%1, %2" : "+v"(a##i) : "n"(i), "v"(b));
#define X_MULS(i) asm volatile("v_mul_f32 %0, s20, %1" : "=v"(a##i) : "v"(b));
#define X_FMAS(i) asm volatile("v_fma_f32 %0, s20, %1, -%2" : "=v"(a##i) : "v"(b), "v"(c));
#define X_SUBREVS(i) asm volatile("v_subrev_f32 %0, s20, %0" : "+v"(a##i));
#define X_OR(i) asm volatile("v_or_b32 %0, %0, %1" : "+v"(a##i) : "v"(b));
#define X_CHAIN(i) asm volatile("v_mul_f32 %0, s20, %2\n v_fma_f32 %1, s20, %3, -%4\n v_fmac_f32 %0, s21, %3\n v_fmac_f32 %1, s21, %4\n v_fmac_f32 %0, s22, %4\n v_fmac_f32 %1, s22, %2\n v_fmac_f32 %1, %0, %0\n v_sub_f32 %0, %0, %2\n v_subrev_f32 %1, s23, %1\n v_or_b32 %0, %0, %1\n v_alignbit_b32 %5, %5, %0, 31" : "=&v"(a##i), "=&v"(u##i), "+v"(b), "+v"(c), "+v"(d), "+v"(e));
#define X_RDL(i) asm volatile("v_readlane_b32 s20, %0, 3" ::"v"(a##i) : "s20");

KERNEL(fma_f64, D64, REP8(X_FMA64), S64)
KERNEL(mul_f64, D64, REP8(X_MUL64), S64)
KERNEL(add_f64, D64, REP8(X_ADD64), S64)
KERNEL(rcp_f64, D64, REP8(X_RCP64), S64)
KERNEL(rsq_f64, D64, REP8(X_RSQ64), S64)
KERNEL(div_scale_f64, D64, REP8(X_DSC64), S64)
KERNEL(div_fmas_f64, D64, REP8(X_DFM64), S64)
KERNEL(div_fixup_f64, D64, REP8(X_DFX64), S64)
KERNEL(ldexp_f64, D64, REP8(X_LDX64), S64)
KERNEL(cmp_f64, D64, REP8(X_CMP64), S64)
KERNEL(fma_f32, D32, REP8(X_FMA32), S32)
KERNEL(fmac_f32, D32, REP8(X_FMAC32), S32)
KERNEL(cmp_f32_vcc, D32, REP8(X_CMP32), S32)
KERNEL(cmp_f32_sgpr, D32, REP8(X_CMP32S), S32)
KERNEL(cndmask_b32, DI, REP8(X_CND), SI)
KERNEL(mov_b32, DI, REP8(X_MOV), SI)
KERNEL(or3_b32, DI, REP8(X_OR3), SI)
KERNEL(addc_u32, DI, REP8(X_ADDC), SI)
KERNEL(readlane, DI, REP8(X_RDL), SI)
KERNEL(fmac_f32_sgpr, D32, REP8(X_FMACS), S32)
KERNEL(fmac_f32_sgpr_var, D32, REP8(X_FMACS2), S32)
KERNEL(mul_f32_sgpr, D32, REP8(X_MULS), S32)
KERNEL(fma_f32_sgpr, D32, REP8(X_FMAS), S32)
KERNEL(subrev_f32_sgpr, D32, REP8(X_SUBREVS), S32)
KERNEL(or_b32, DI, REP8(X_OR), SI)
#define DCH float a0, a1, a2, a3, a4, a5, a6, a7, u0, u1, u2, u3, u4, u5, u6, u7, b = (float)seed, c = b + 1, d = b + 2; unsigned e = 0;
#define SCH if (e == 12345u) out[0] = 1;
KERNEL(sweep_chain_x11, DCH, REP8(X_CHAIN), SCH)
KERNEL(alignbit, DI, REP8(X_ALIGN), SI)
KERNEL(sub_f32, D32, REP8(X_SUB32), S32)
KERNEL(cndmask_sgpr, DI, REP8(X_CNDS), SI)
// round 4: what the range checks and copies of unit() are made of
#define X_FREXP(i) asm volatile("v_frexp_exp_i32_f64 %0, %1" : "=v"(e##i) : "v"(a##i));
#define DFX double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; int e0 = 0, e1 = 0, e2 = 0, e3 = 0, e4 = 0, e5 = 0, e6 = 0, e7 = 0;
#define SFX if (e0 + e1 + e2 + e3 + e4 + e5 + e6 + e7 == 12345) out[0] = 1;
KERNEL(frexp_exp_f64, DFX, REP8(X_FREXP), SFX)
#define X_MOV64(i) asm volatile("v_mov_b64 %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(mov_b64, D64, REP8(X_MOV64), S64)
#define X_MIN3(i) asm volatile("v_min3_i32 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(min3_i32, DI, REP8(X_MIN3), SI)
#define X_BFI(i) asm volatile("v_bfi_b32 %0, %1, %0, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(bfi_b32, DI, REP8(X_BFI), SI)
#define X_CLASS64(i) asm volatile("v_cmp_class_f64 vcc, %0, %1" ::"v"(a##i), "v"(k) : "vcc");
#define DCL double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7; unsigned k = 0x1f8;
KERNEL(cmp_class_f64, DCL, REP8(X_CLASS64), S64)
#define X_ADDU(i) asm volatile("v_add_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(add_u32, DI, REP8(X_ADDU), SI)
#define X_CMPU(i) asm volatile("v_cmp_gt_u32 vcc, %0, %1" ::"v"(a##i), "v"(b) : "vcc");
KERNEL(cmp_gt_u32, DI, REP8(X_CMPU), SI)
// integer multiplies: what index arithmetic is made of
#define X_MULLO(i) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(mul_lo_u32, DI, REP8(X_MULLO), SI)
#define X_MULHI(i) asm volatile("v_mul_hi_u32 %0, %0, %1" : "+v"(a##i) : "v"(b));
KERNEL(mul_hi_u32, DI, REP8(X_MULHI), SI)
#define X_MAD24(i) asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(a##i) : "v"(b), "v"(c));
KERNEL(mad_u32_u24, DI, REP8(X_MAD24), SI)
#define DI64 unsigned long long a0 = (unsigned)seed, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7; unsigned b = (unsigned)seed * 3 + 1, c = 7;
#define SI64 if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345) out[0] = 1;
#define X_MAD64(i) asm volatile("v_mad_u64_u32 %0, s[20:21], %1, %2, %0" : "+v"(a##i) : "v"(b), "v"(c) : "s20", "s21");
KERNEL(mad_u64_u32, DI64, REP8(X_MAD64), SI64)
#define X_LSHLADD(i) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(a##i) : "v"(b));
KERNEL(lshl_add_u32, DI, REP8(X_LSHLADD), SI)
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f16v __attribute__((ext_vector_type(16)));
#define DPK f2 p0 = {(float)seed, 1.f}, p1 = p0 + 1.f, p2 = p0 + 2.f, p3 = p0 + 3.f, p4 = p0 + 4.f, p5 = p0 + 5.f, p6 = p0 + 6.f, p7 = p0 + 7.f, pb = p0 * 0.5f, pc = {1.0000001f, 0.9999999f};
#define SPK if (p0.x + p1.x + p2.x + p3.x + p4.y + p5.y + p6.y + p7.y == 12345.678f) out[0] = 1;
KERNEL(pk_fma_f32, DPK, REP8(X_PKFMA), SPK)
KERNEL(pk_mul_f32, DPK, REP8(X_PKMUL), SPK)
#define DMF f16v m0 = {}, m1 = {}, m2 = {}, m3 = {}; float fa = (float)seed, fb = fa * 0.5f;
#define SMF if (m0[0] + m1[1] + m2[2] + m3[3] == 12345.678f) out[0] = 1;
#define REP4(x) x(0) x(1) x(2) x(3)
KERNEL(mfma_32x32x2_f32, DMF, REP4(X_MFMA) REP4(X_MFMA), SMF)
__global__ void k_cvt_f32_f64(unsigned long long *out, double seed)
{
    float a0 = 0, a1 = 0, a2 = 0, a3 = 0, a4 = 0, a5 = 0, a6 = 0, a7 = 0;
    double bd = seed;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; it++)
    {
        REP8(X_CVT) REP8(X_CVT) REP8(X_CVT) REP8(X_CVT)
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.f)
        out[0] = 1;
    if ((threadIdx.x & 63) == 0)
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

typedef void (*kern_t)(unsigned long long *, double);
struct Entry
{
    const char *name;
    kern_t fn;
};

int main()
{
    Entry list[] = {{"v_fma_f64", k_fma_f64}, {"v_mul_f64", k_mul_f64}, {"v_add_f64", k_add_f64}, {"v_rcp_f64", k_rcp_f64},
                    {"v_rsq_f64", k_rsq_f64}, {"v_div_scale_f64", k_div_scale_f64}, {"v_div_fmas_f64", k_div_fmas_f64},
                    {"v_div_fixup_f64", k_div_fixup_f64}, {"v_ldexp_f64", k_ldexp_f64}, {"v_cmp_lt_f64", k_cmp_f64},
                    {"v_cvt_f32_f64", k_cvt_f32_f64}, {"v_fma_f32", k_fma_f32}, {"v_fmac_f32", k_fmac_f32},
                    {"v_cmp_lt_f32 vcc", k_cmp_f32_vcc}, {"v_cmp_lt_f32 sgpr", k_cmp_f32_sgpr}, {"v_cndmask_b32", k_cndmask_b32},
                    {"v_mov_b32", k_mov_b32}, {"v_or3_b32", k_or3_b32}, {"v_addc_co_u32", k_addc_u32}, {"v_readlane_b32", k_readlane}, {"v_fmac_f32 s,v", k_fmac_f32_sgpr}, {"v_fmac_f32 s2x,v", k_fmac_f32_sgpr_var}, {"v_mul_f32 s,v", k_mul_f32_sgpr}, {"v_fma_f32 s,v,-v", k_fma_f32_sgpr}, {"v_subrev_f32 s,v", k_subrev_f32_sgpr}, {"v_or_b32", k_or_b32}, {"sweep chain (11)", k_sweep_chain_x11}, {"v_alignbit_b32", k_alignbit}, {"v_sub_f32", k_sub_f32}, {"v_cndmask_b32 sgpr", k_cndmask_sgpr}, {"v_pk_fma_f32", k_pk_fma_f32}, {"v_pk_mul_f32", k_pk_mul_f32}, {"v_mfma_f32_32x32x2", k_mfma_32x32x2_f32},
                    {"v_frexp_exp_i32_f64", k_frexp_exp_f64}, {"v_mov_b64", k_mov_b64}, {"v_min3_i32", k_min3_i32}, {"v_bfi_b32", k_bfi_b32},
                    {"v_cmp_class_f64", k_cmp_class_f64}, {"v_add_u32", k_add_u32}, {"v_cmp_gt_u32", k_cmp_gt_u32},
                    {"v_mul_lo_u32", k_mul_lo_u32}, {"v_mul_hi_u32", k_mul_hi_u32}, {"v_mad_u32_u24", k_mad_u32_u24}, {"v_mad_u64_u32", k_mad_u64_u32},
                    {"v_lshl_add_u32", k_lshl_add_u32}};
    unsigned long long *d;
    hipMalloc(&d, 1 << 20);
    printf("%-20s %10s %10s %10s %10s   (cycles per wave-instruction per SIMD = wave cycles / instrs * waves... see columns)\n", "instruction",
           "1w/SIMD", "2w/SIMD", "3w/SIMD", "4w/SIMD");
    for (auto &e : list)
    {
        printf("%-20s", e.name);
        for (int w = 1; w <= 4; w++)
        {
            const int block = 256 * w; // 4 SIMDs x w waves, one block per CU
            const int grid = 256;
            if (block > 1024)
            { // 2 blocks per CU of 512
            }
            const int b = block > 1024 ? 512 : block, g = block > 1024 ? grid * (block / 512) : grid;
            hipLaunchKernelGGL(e.fn, dim3(g), dim3(b), 0, 0, d, 1.25);
            hipDeviceSynchronize();
            hipLaunchKernelGGL(e.fn, dim3(g), dim3(b), 0, 0, d, 1.25);
            hipDeviceSynchronize();
            const int waves = g * b / 64;
            std::vector<unsigned long long> h(waves);
            hipMemcpy(h.data(), d, waves * 8, hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            const double med = (double)h[waves / 2];
            // s_memtime ticks at a constant 100 MHz on gfx9; report SIMD-cycles per instruction instead through wall: use ratio to fma_f32 later
            printf(" %10.3f", med / (ITER * 32.0) / w);
        }
        printf("\n");
    }
    return 0;
}
