// test_sweep_mfma.hip -- sweep64_mfma (csrc/trt_rounds.hpp) against the host evaluation of trt_filter_sign_mfma /
// trt_filter_sign_fixed_dir_mfma on random spheres and rays.  Diagnostic tool: hipcc ... && run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <vector>
#include <random>
#include "../terminalraytracer_amd/csrc/trt_rounds.hpp"

__global__ void k_sweep(const float *a_xy, const float *a_zk, const trt_ray_filter *flt, unsigned *out, int fixed)
{
    __shared__ float s_xy[128], s_zk[128];
    for (int i = threadIdx.x; i < 128; i += 64)
        s_xy[i] = a_xy[i], s_zk[i] = a_zk[i];
    __syncthreads();
    const int lane = threadIdx.x;
    trt_ray_filter f = flt[lane];
    unsigned h0, h1;
    if (fixed)
        trt::sweep64_mfma<true>(s_xy, s_zk, lane, f, h0, h1);
    else
        trt::sweep64_mfma<false>(s_xy, s_zk, lane, f, h0, h1);
    out[2 * lane] = h0;
    out[2 * lane + 1] = h1;
}

int main()
{
    std::mt19937 rng(5);
    std::uniform_real_distribution<float> U(-1.f, 1.f);
    int total_bad = 0;
    for (int fixed = 0; fixed < 2; fixed++)
    {
        std::vector<float> sph(64 * 4), axy(128), azk(128);
        for (int j = 0; j < 64; j++)
        {
            sph[4 * j] = 4 * U(rng), sph[4 * j + 1] = 2 * U(rng), sph[4 * j + 2] = 4 * U(rng);
            sph[4 * j + 3] = sph[4 * j] * sph[4 * j] + sph[4 * j + 1] * sph[4 * j + 1] + sph[4 * j + 2] * sph[4 * j + 2] - 0.1f - 0.1f * (U(rng) + 1) - (fixed ? 12.0f * (U(rng) + 1) : 0.0f);
        }
        for (int b = 0; b < 2; b++)
            for (int l = 0; l < 64; l++)
            {
                const float *e = &sph[4 * (32 * b + (l & 31))];
                axy[64 * b + l] = l < 32 ? e[0] : e[1];
                azk[64 * b + l] = l < 32 ? e[2] : e[3];
            }
        std::vector<trt_ray_filter> f(64);
        for (auto &r : f)
        {
            float d[3] = {U(rng), U(rng), U(rng)};
            float n = sqrtf(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
            r.dx = d[0] / n, r.dy = d[1] / n, r.dz = d[2] / n;
            r.wx = 6 * U(rng), r.wy = 6 * U(rng), r.wz = 6 * U(rng);
            r.neg_thr = -(0.25f * (r.wx * r.wx + r.wy * r.wy + r.wz * r.wz) - 1e-3f);
            r.cd_min = 0, r.ok = 1;
        }
        float *dxy, *dzk; trt_ray_filter *df; unsigned *dout;
        hipMalloc(&dxy, 512); hipMalloc(&dzk, 512); hipMalloc(&df, sizeof(trt_ray_filter) * 64); hipMalloc(&dout, 512);
        hipMemcpy(dxy, axy.data(), 512, hipMemcpyHostToDevice); hipMemcpy(dzk, azk.data(), 512, hipMemcpyHostToDevice);
        hipMemcpy(df, f.data(), sizeof(trt_ray_filter) * 64, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_sweep, dim3(1), dim3(64), 0, 0, dxy, dzk, df, dout, fixed);
        std::vector<unsigned> out(128);
        hipMemcpy(out.data(), dout, 512, hipMemcpyDeviceToHost);
        int bad = 0, cands = 0;
        for (int ray = 0; ray < 64; ray++)
            for (int j = 0; j < 64; j++)
            {
                const unsigned sign = fixed ? trt_filter_sign_fixed_dir_mfma(&f[ray], sph[4 * j], sph[4 * j + 1], sph[4 * j + 2], sph[4 * j + 3])
                                            : trt_filter_sign_mfma(&f[ray], sph[4 * j], sph[4 * j + 1], sph[4 * j + 2], sph[4 * j + 3]);
                const int want = !(sign >> 31);
                // sphere j of the chunk -> (h, p): j = 32 s + 8 q + 4 h + t, p = 16 s + 4 q + t
                const int s = j >> 5, q = (j >> 3) & 3, h = (j >> 2) & 1, t = j & 3, p = 16 * s + 4 * q + t;
                const int got = (out[2 * ray + h] >> (31 - p)) & 1;
                cands += want;
                if (got != want)
                {
                    if (bad < 8)
                        printf("fixed %d ray %d sphere %d: want %d got %d\n", fixed, ray, j, want, got);
                    bad++;
                }
            }
        for (int ray : {0, 1, 40})
        {
            unsigned w[2] = {0, 0};
            for (int j = 0; j < 64; j++)
            {
                const unsigned sign = fixed ? trt_filter_sign_fixed_dir_mfma(&f[ray], sph[4 * j], sph[4 * j + 1], sph[4 * j + 2], sph[4 * j + 3])
                                            : trt_filter_sign_mfma(&f[ray], sph[4 * j], sph[4 * j + 1], sph[4 * j + 2], sph[4 * j + 3]);
                const int s = j >> 5, q = (j >> 3) & 3, h = (j >> 2) & 1, t = j & 3, p = 16 * s + 4 * q + t;
                if (!(sign >> 31)) w[h] |= 0x80000000u >> p;
            }
            printf("fixed %d ray %2d: want %08x %08x   got %08x %08x\n", fixed, ray, w[0], w[1], out[2 * ray], out[2 * ray + 1]);
        }
        printf("fixed=%d: %d mismatches of 4096 (candidates %d)\n", fixed, bad, cands);
        total_bad += bad;
    }
    return total_bad != 0;
}
