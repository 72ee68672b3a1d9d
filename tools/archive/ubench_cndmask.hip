// ubench_cndmask.hip -- what a select costs on gfx950, by encoding and by where its mask lives.
// ubench_valu.hip found v_cndmask_b32 with VCC at 15.5 cycles per wave instruction against 2.9 with an SGPR pair.  Here:
//   e32 vcc | e64 vcc | e64 s[20:21] | a compare into vcc followed by two selects on it (a 64-bit select, the kernel's pattern),
//   e32 and e64 | the same with the compare writing s[20:21].
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_cndmask tools/archive/ubench_cndmask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000
#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)
#define X0(i) asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub) : "vcc");
#define X1(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub) : "vcc");
#define X2(i) asm volatile("v_cndmask_b32_e64 %0, %0, %1, s[20:21]" : "+v"(u##i) : "v"(ub) : "s20", "s21");
#define X3(i) asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(u##i), "+v"(w##i) : "v"(da), "v"(db), "v"(ub) : "vcc");
#define X4(i) asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc" : "+v"(u##i), "+v"(w##i) : "v"(da), "v"(db), "v"(ub) : "vcc");
#define X5(i) asm volatile("v_cmp_lt_f64_e64 s[20:21], %2, %3\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]" : "+v"(u##i), "+v"(w##i) : "v"(da), "v"(db), "v"(ub) : "s20", "s21");
#define X6(i) asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n s_nop 4\n v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc" : "+v"(u##i), "+v"(w##i) : "v"(da), "v"(db), "v"(ub) : "vcc");
#define X7(i) asm volatile("v_cmp_lt_f64_e32 vcc, %2, %3\n s_mov_b64 s[20:21], vcc\n v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]" : "+v"(u##i), "+v"(w##i) : "v"(da), "v"(db), "v"(ub) : "vcc", "s20", "s21");

template <int WHAT>
__global__ void k(unsigned long long *out, double seed)
{
    unsigned u0 = (unsigned)seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, ub = u0 * 3 + 1;
    unsigned w0 = u0 + 8, w1 = u0 + 9, w2 = u0 + 10, w3 = u0 + 11, w4 = u0 + 12, w5 = u0 + 13, w6 = u0 + 14, w7 = u0 + 15;
    double da = seed, db = seed + threadIdx.x;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; it++)
    {
        if (WHAT == 0) { REP8(X0) REP8(X0) REP8(X0) REP8(X0) }
        if (WHAT == 1) { REP8(X1) REP8(X1) REP8(X1) REP8(X1) }
        if (WHAT == 2) { REP8(X2) REP8(X2) REP8(X2) REP8(X2) }
        if (WHAT == 3) { REP8(X3) REP8(X3) REP8(X3) REP8(X3) }
        if (WHAT == 4) { REP8(X4) REP8(X4) REP8(X4) REP8(X4) }
        if (WHAT == 5) { REP8(X5) REP8(X5) REP8(X5) REP8(X5) }
        if (WHAT == 6) { REP8(X6) REP8(X6) REP8(X6) REP8(X6) }
        if (WHAT == 7) { REP8(X7) REP8(X7) REP8(X7) REP8(X7) }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 + w0 + w1 + w2 + w3 + w4 + w5 + w6 + w7 == 12345)
        out[0] = 1;
    if ((threadIdx.x & 63) == 0)
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int WHAT>
void run(const char *name, int instrs)
{
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    printf("%-58s", name);
    for (int waves_per_simd : {1, 2, 4})
    {
        const int threads = 256 * waves_per_simd;
        unsigned long long *d;
        hipMalloc(&d, sizeof(unsigned long long) * cus * threads / 64);
        hipMemset(d, 0, sizeof(unsigned long long) * cus * threads / 64);
        hipLaunchKernelGGL(k<WHAT>, dim3(cus), dim3(threads), 0, 0, d, 1.5);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(cus * threads / 64);
        hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf("  %dw/SIMD %7.2f", waves_per_simd, (double)h[h.size() / 2] / (ITER * 32.0 * instrs) / waves_per_simd);
        hipFree(d);
    }
    printf("   cycles per wave instruction per SIMD\n");
}

int main()
{
    run<0>("v_cndmask_b32_e32 v, v, v, vcc", 1);
    run<1>("v_cndmask_b32_e64 v, v, v, vcc", 1);
    run<2>("v_cndmask_b32_e64 v, v, v, s[20:21]", 1);
    run<3>("v_cmp_lt_f64 vcc + 2 x v_cndmask_e32 vcc (3 instr)", 3);
    run<4>("v_cmp_lt_f64 vcc + 2 x v_cndmask_e64 vcc (3 instr)", 3);
    run<5>("v_cmp_lt_f64 s[20:21] + 2 x v_cndmask_e64 s[20:21] (3 instr)", 3);
    run<6>("v_cmp_lt_f64 vcc + s_nop 4 + 2 x v_cndmask_e32 vcc (3+1)", 3);
    run<7>("v_cmp_lt_f64 vcc + s_mov s[20:21], vcc + 2 x e64 s[20:21]", 3);
    return 0;
}
