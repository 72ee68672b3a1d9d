// ubench_exec.hip -- does gfx950 skip the passes of a wave64 VALU instruction whose lanes are all switched off?
// The same loop of independent v_fma_f64 / v_mul_f64 / v_cndmask_b32 / v_fma_f32 runs under exec masks with 64, 32 (lower
// half), 16, 1 and 4-scattered (one lane per 16-lane group) active lanes; s_memtime around the loop, 1 / 2 / 4 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O2 -o tools/ubench_exec tools/archive/ubench_exec.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>

#define ITER 2000
#define REP8(x) x(0) x(1) x(2) x(3) x(4) x(5) x(6) x(7)
#define X_FMA64(i) asm volatile("v_fma_f64 %0, %0, %1, %2" : "+v"(a##i) : "v"(c), "v"(b));
#define X_CND(i) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u##i) : "v"(ub) : "vcc");
#define X_FMA32(i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(f##i) : "v"(fc), "v"(fb));

template <int WHAT>
__global__ void k(unsigned long long *out, double seed, unsigned long long mask)
{
    double a0 = seed, a1 = seed + 1, a2 = seed + 2, a3 = seed + 3, a4 = seed + 4, a5 = seed + 5, a6 = seed + 6, a7 = seed + 7, b = seed * 0.5 + 1e-3, c = 1.0000001;
    unsigned u0 = (unsigned)seed, u1 = u0 + 1, u2 = u0 + 2, u3 = u0 + 3, u4 = u0 + 4, u5 = u0 + 5, u6 = u0 + 6, u7 = u0 + 7, ub = u0 * 3 + 1;
    float f0 = (float)seed, f1 = f0 + 1, f2 = f0 + 2, f3 = f0 + 3, f4 = f0 + 4, f5 = f0 + 5, f6 = f0 + 6, f7 = f0 + 7, fb = f0 * 0.5f + 1e-3f, fc = 1.0000001f;
    unsigned long long t0 = 0, t1 = 0;
    const bool on = (mask >> (threadIdx.x & 63)) & 1ull;
    if (on) // exec = mask for the whole loop
    {
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < ITER; it++)
        {
            if (WHAT == 0) { REP8(X_FMA64) REP8(X_FMA64) REP8(X_FMA64) REP8(X_FMA64) }
            if (WHAT == 1) { REP8(X_CND) REP8(X_CND) REP8(X_CND) REP8(X_CND) }
            if (WHAT == 2) { REP8(X_FMA32) REP8(X_FMA32) REP8(X_FMA32) REP8(X_FMA32) }
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 12345.678 || u0 + u1 + u2 + u3 + u4 + u5 + u6 + u7 == 12345 || f0 + f1 + f2 + f3 + f4 + f5 + f6 + f7 == 12345.678f)
        out[0] = 1;
    const unsigned long long first = __ballot(on);
    if ((int)(threadIdx.x & 63) == __builtin_ctzll(first ? first : 1ull))
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int WHAT>
void run(const char *name)
{
    const unsigned long long masks[] = {~0ull, 0xffffull, 0xffull, 0xfull, 0x3ull, 1ull, 0x0001000100010001ull, 0x1111111111111111ull, 0x5555555555555555ull, 0x00ff00ff00ff00ffull};
    const char *names[] = {"64", "low16", "low8", "low4", "low2", "1", "1per16", "1per4", "1per2", "8per16"};
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    for (int waves_per_simd : {1, 2, 4})
    {
        const int threads = 256 * waves_per_simd; // one block per CU: waves_per_simd waves on each of the 4 SIMDs
        unsigned long long *d;
        hipMalloc(&d, sizeof(unsigned long long) * cus * threads / 64);
        printf("%-10s %d wave(s)/SIMD:", name, waves_per_simd);
        for (int m = 0; m < 10; m++)
        {
            hipMemset(d, 0, sizeof(unsigned long long) * cus * threads / 64);
            hipLaunchKernelGGL(k<WHAT>, dim3(cus), dim3(threads), 0, 0, d, 1.5, masks[m]);
            hipDeviceSynchronize();
            std::vector<unsigned long long> h(cus * threads / 64);
            hipMemcpy(h.data(), d, h.size() * sizeof(unsigned long long), hipMemcpyDeviceToHost);
            std::sort(h.begin(), h.end());
            printf("  %s %.2f", names[m], (double)h[h.size() / 2] / (ITER * 32.0));
        }
        printf("   (s_memtime ticks per wave instruction)\n");
        hipFree(d);
    }
}

int main()
{
    run<0>("v_fma_f64");
    run<1>("v_cndmask");
    run<2>("v_fma_f32");
    return 0;
}
