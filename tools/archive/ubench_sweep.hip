// ubench_sweep.hip -- the phase-1 culling sweep of the production kernel in isolation (gfx950).
// Same arithmetic as trt_filter_sign (csrc/trt_filter.h); 64 spheres, ITER sweeps per wave, 1..4 waves per SIMD.
// Variants: table through scalar loads (SGPR operands) or LDS broadcast reads (VGPR operands); spheres handled one
// after the other, or 2 / 4 at a time with their instructions interleaved in the source (ILP within the wave).
// Prints SIMD cycles per wave-level sphere test.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include "../terminalraytracer_amd/csrc/trt_filter.h"

#define ITER 400
typedef const float __attribute__((address_space(4))) *cfp;

template <int FEED, int ILP>
__global__ void sweep(const float *table, unsigned long long *out, unsigned *sink, float seed)
{
    extern __shared__ __attribute__((aligned(16))) float4 l_tab[];
    for (int i = threadIdx.x; i < 64; i += blockDim.x)
        l_tab[i] = ((const float4 *)table)[i];
    __syncthreads();
    trt_ray_filter f;
    const float t = seed + 0.001f * threadIdx.x;
    f.dx = 0.6f + t, f.dy = 0.64f - t, f.dz = 0.48f, f.wx = 1.0f + t, f.wy = -2.0f, f.wz = 0.5f * t, f.neg_thr = -1.0f - t, f.cd_min = -3.0f + t, f.ok = 1;
    cfp tab = (cfp)(uintptr_t)table;
    unsigned acc = 0;
    unsigned long long t0, t1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
    for (int it = 0; it < ITER; it++)
    {
        unsigned bits[ILP];
#pragma unroll
        for (int q = 0; q < ILP; q++)
            bits[q] = ~0u;
        for (int g = 0; g < 64; g += 8)
        {
#pragma unroll
            for (int j = 0; j < 8; j += ILP)
            {
                float cx[ILP], cy[ILP], cz[ILP], kk[ILP], cd[ILP], cw[ILP];
#pragma unroll
                for (int q = 0; q < ILP; q++)
                {
                    if (FEED == 0)
                    {
                        const cfp e = tab + (g + j + q) * 4;
                        cx[q] = e[0], cy[q] = e[1], cz[q] = e[2], kk[q] = e[3];
                    }
                    else
                    {
                        const float4 e = l_tab[g + j + q];
                        cx[q] = e.x, cy[q] = e.y, cz[q] = e.z, kk[q] = e.w;
                    }
                }
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cd[q] = cx[q] * f.dx;
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cw[q] = __builtin_fmaf(cx[q], f.wx, f.neg_thr);
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cd[q] = __builtin_fmaf(cy[q], f.dy, cd[q]);
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cw[q] = __builtin_fmaf(cy[q], f.wy, cw[q]);
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cd[q] = __builtin_fmaf(cz[q], f.dz, cd[q]);
#pragma unroll
                for (int q = 0; q < ILP; q++)
                    cw[q] = __builtin_fmaf(cz[q], f.wz, cw[q]);
#pragma unroll
                for (int q = 0; q < ILP; q++)
                {
                    const float m = __builtin_fmaf(cd[q], cd[q], cw[q]) - kk[q];
                    const float n = cd[q] - f.cd_min;
                    bits[q] = __builtin_amdgcn_alignbit(bits[q], __builtin_bit_cast(unsigned, m) | __builtin_bit_cast(unsigned, n), 31);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ILP; q++)
            acc ^= bits[q];
        asm volatile("" : "+v"(f.dx), "+v"(f.wx)); // keep the sweep inside the timed loop
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
    if (acc == 0x12345678u)
        sink[0] = acc;
    if ((threadIdx.x & 63) == 0)
        out[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = t1 - t0;
}

template <int FEED, int ILP>
void run(const char *name, const float *d_tab, unsigned long long *d_out, unsigned *d_sink)
{
    printf("%-34s", name);
    for (int w = 1; w <= 4; w++)
    {
        const int block = 256, grid = 256 * w; // w blocks of 4 waves per CU
        for (int rep = 0; rep < 2; rep++)
        {
            hipLaunchKernelGGL((sweep<FEED, ILP>), dim3(grid), dim3(block), 64 * 16, 0, d_tab, d_out, d_sink, 0.25f);
            hipDeviceSynchronize();
        }
        const int waves = grid * block / 64;
        std::vector<unsigned long long> h(waves);
        hipMemcpy(h.data(), d_out, waves * 8, hipMemcpyDeviceToHost);
        std::sort(h.begin(), h.end());
        printf(" %9.2f", (double)h[waves / 2] / (ITER * 64.0) / w);
    }
    printf("\n");
}

int main()
{
    std::vector<float> tab(64 * 4);
    for (int i = 0; i < 64; i++)
    {
        tab[4 * i] = 0.1f * i - 3.f, tab[4 * i + 1] = 0.05f * i, tab[4 * i + 2] = 2.f - 0.07f * i, tab[4 * i + 3] = 1.f + 0.01f * i;
    }
    float *d_tab;
    unsigned long long *d_out;
    unsigned *d_sink;
    hipMalloc(&d_tab, tab.size() * 4);
    hipMalloc(&d_out, 1 << 20);
    hipMalloc(&d_sink, 64);
    hipMemcpy(d_tab, tab.data(), tab.size() * 4, hipMemcpyHostToDevice);
    printf("%-34s %9s %9s %9s %9s   cycles per wave-level sphere test (per SIMD)\n", "variant", "1w/SIMD", "2w/SIMD", "3w/SIMD", "4w/SIMD");
    run<0, 1>("scalar loads, 1 sphere at a time", d_tab, d_out, d_sink);
    run<0, 2>("scalar loads, 2 interleaved", d_tab, d_out, d_sink);
    run<0, 4>("scalar loads, 4 interleaved", d_tab, d_out, d_sink);
    run<1, 1>("LDS broadcast, 1 sphere at a time", d_tab, d_out, d_sink);
    run<1, 2>("LDS broadcast, 2 interleaved", d_tab, d_out, d_sink);
    run<1, 4>("LDS broadcast, 4 interleaved", d_tab, d_out, d_sink);
    return 0;
}
