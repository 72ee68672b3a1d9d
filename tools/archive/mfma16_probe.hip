// mfma16_probe.hip -- operand / result layout and rounding model of v_mfma_f32_16x16x4_f32 on gfx950.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/mfma16_probe tools/archive/mfma16_probe.hip && /tmp/mfma16_probe
// D[i][j] = sum_k A[i][k] * B[k][j] + C[i][j], i, j < 16, k < 4.  Expected layout (CDNA3 ISA):
//   A: lane l holds A[i = l%16][k = l/16];  B: lane l holds B[k = l/16][j = l%16];  C/D: lane l, register v holds [i = 4*(l/16) + v][j = l%16]
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef float f4v __attribute__((ext_vector_type(4)));
__global__ void probe(const float *A /*16x4*/, const float *B /*4x16*/, const float *C /*16x16*/, float *out /*64 x 4*/)
{
    const int l = threadIdx.x;
    const float a = A[(l % 16) * 4 + l / 16], b = B[(l / 16) * 16 + l % 16];
    f4v c;
    for (int v = 0; v < 4; v++)
        c[v] = C[(4 * (l / 16) + v) * 16 + l % 16];
    const f4v d = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    for (int v = 0; v < 4; v++)
        out[l * 4 + v] = d[v];
}
int main()
{
    std::vector<float> A(64), B(64), C(256), out(256);
    unsigned s = 12345;
    auto rnd = [&]() { s = s * 1664525u + 1013904223u; return ((int)(s >> 8) % 20001 - 10000) / 977.0f; };
    int model_hits[4] = {0, 0, 0, 0}, total = 0, layout_bad = 0;
    for (int trial = 0; trial < 200; trial++)
    {
        for (auto &x : A) x = rnd() * (trial % 3 == 0 ? 1e3f : 1.0f);
        for (auto &x : B) x = rnd();
        for (auto &x : C) x = rnd() * (trial % 2 ? 100.0f : 0.01f);
        float *dA, *dB, *dC, *dO;
        hipMalloc(&dA, 256); hipMalloc(&dB, 256); hipMalloc(&dC, 1024); hipMalloc(&dO, 1024);
        hipMemcpy(dA, A.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), 256, hipMemcpyHostToDevice); hipMemcpy(dC, C.data(), 1024, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dO);
        hipMemcpy(out.data(), dO, 1024, hipMemcpyDeviceToHost);
        hipFree(dA); hipFree(dB); hipFree(dC); hipFree(dO);
        for (int l = 0; l < 64; l++)
            for (int v = 0; v < 4; v++)
            {
                const int i = 4 * (l / 16) + v, j = l % 16;
                const float got = out[l * 4 + v];
                float m0 = C[i * 16 + j]; // model 0: FMA chain k = 0..3 starting from C
                for (int k = 0; k < 4; k++) m0 = fmaf(A[i * 4 + k], B[k * 16 + j], m0);
                float m1 = C[i * 16 + j]; // model 1: chain k = 3..0
                for (int k = 3; k >= 0; k--) m1 = fmaf(A[i * 4 + k], B[k * 16 + j], m1);
                double e = C[i * 16 + j]; // model 2: exact sum, one rounding
                for (int k = 0; k < 4; k++) e += (double)A[i * 4 + k] * (double)B[k * 16 + j];
                const float m2 = (float)e;
                float m3 = 0.0f; // model 3: chain of the products from 0, C added last
                for (int k = 0; k < 4; k++) m3 = fmaf(A[i * 4 + k], B[k * 16 + j], m3);
                m3 += C[i * 16 + j];
                total++;
                model_hits[0] += got == m0; model_hits[1] += got == m1; model_hits[2] += got == m2; model_hits[3] += got == m3;
                if (fabs(got - e) > 1e-4 * (fabs(e) + 1.0)) layout_bad++;
            }
    }
    printf("layout as documented: %d of %d results off\n", layout_bad, total);
    printf("bit-identical to: fma chain k=0..3 from C %d | chain k=3..0 %d | exact sum rounded once %d | products first, C last %d   (of %d)\n",
           model_hits[0], model_hits[1], model_hits[2], model_hits[3], total);
    return layout_bad != 0;
}
