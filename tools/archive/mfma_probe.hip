#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <cmath>
#include <cstring>
typedef float f16v __attribute__((ext_vector_type(16)));
typedef unsigned u2v __attribute__((ext_vector_type(2)));
// D[sphere][ray] = sum_k A[sphere][k] * B[k][ray], 32 spheres x 32 rays x K=4 (two K=2 MFMAs)
__global__ void probe(const float *sph /*32x4*/, const float *ray /*64x4: per LANE*/, float *out /*64 lanes x 16*/, unsigned *swapout)
{
    const int l = threadIdx.x;
    // A operands: lane l holds A[m=l%32][k=l/32]
    const float a01 = sph[(l % 32) * 4 + (l / 32)];       // k = 0 (x) for lanes<32, k = 1 (y) for lanes>=32
    const float a23 = sph[(l % 32) * 4 + 2 + (l / 32)];   // k = 2 (z), k = 3 (w)
    // each lane owns a ray: (x,y,z,w) ; B operand for ray block 0: lanes<32: x of ray l, lanes>=32: y of ray l-32
    float rx = ray[l * 4 + 0], ry = ray[l * 4 + 1], rz = ray[l * 4 + 2], rw = ray[l * 4 + 3];
    // v_permlane32_swap a, b : a <- [a.lo, b.lo], b <- [a.hi, b.hi]   (in place)
    float sx = rx, sy = ry, sz = rz, sw_ = rw;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(sx), "+v"(sy));
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(sz), "+v"(sw_));
    swapout[l] = __builtin_bit_cast(unsigned, sx); swapout[64 + l] = __builtin_bit_cast(unsigned, sy);
    const float b01_blk0 = sx, b01_blk1 = sy, b23_blk0 = sz, b23_blk1 = sw_;
    f16v acc = {0};
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01_blk0, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, b23_blk0, acc, 0, 0, 0);
    f16v acc1 = {0};
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a01, b01_blk1, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a23, b23_blk1, acc1, 0, 0, 0);
    for (int v = 0; v < 16; v++) { out[l * 16 + v] = acc[v]; out[64 * 16 + l * 16 + v] = acc1[v]; }
}
int main()
{
    std::vector<float> sph(32 * 4), ray(64 * 4), out(2 * 64 * 16);
    std::vector<unsigned> sw(128);
    for (int i = 0; i < 32 * 4; i++) sph[i] = 0.37f * (i % 7) - 1.f + 0.01f * i;
    for (int i = 0; i < 64 * 4; i++) ray[i] = 0.11f * (i % 5) - 0.3f + 0.003f * i;
    float *ds, *dr, *dout; unsigned *dsw;
    hipMalloc(&ds, sph.size() * 4); hipMalloc(&dr, ray.size() * 4); hipMalloc(&dout, out.size() * 4); hipMalloc(&dsw, 512);
    hipMemcpy(ds, sph.data(), sph.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dr, ray.data(), ray.size() * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, ds, dr, dout, dsw);
    hipMemcpy(out.data(), dout, out.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(sw.data(), dsw, 512, hipMemcpyDeviceToHost);
    // check swap semantics
    int swap_ok = 1;
    for (int l = 0; l < 64; l++) {
        float want_x = l < 32 ? ray[l * 4 + 0] : ray[(l - 32) * 4 + 1];            // blk0 operand: [x.lo, y.lo]
        float want_y = l < 32 ? ray[(l + 32) * 4 + 0] : ray[l * 4 + 1];            // blk1 operand: [x.hi, y.hi]
        float gx, gy; memcpy(&gx, &sw[l], 4); memcpy(&gy, &sw[64 + l], 4);
        if (gx != want_x || gy != want_y) swap_ok = 0;
    }
    printf("permlane32_swap gives [a.lo,b.lo],[a.hi,b.hi]: %s\n", swap_ok ? "yes" : "NO");
    // check layout: lane l, vgpr v -> ray = l%32 (+32*blk), sphere = (v/4)*8 + (l/32)*4 + v%4 ; fma chain order k=0..3
    int bad = 0, bitexact = 0, total = 0; int cnt[2][2][4] = {};
    for (int blk = 0; blk < 2; blk++) for (int l = 0; l < 64; l++) for (int v = 0; v < 16; v++) {
        int r = l % 32 + 32 * blk, s = (v / 4) * 8 + (l / 32) * 4 + v % 4;
        float acc = 0.f;
        for (int k = 0; k < 4; k++) acc = fmaf(sph[s * 4 + k], ray[r * 4 + k], acc);
        float got = out[blk * 64 * 16 + l * 16 + v];
        total++; if (got == acc) bitexact++; if (fabsf(got - acc) > 1e-5f * (fabsf(acc) + 1)) { bad++; cnt[blk][l/32][v/4]++; }
    }
    for (int blk=0;blk<2;blk++) for (int h=0;h<2;h++) printf("blk %d lanehalf %d wrong by v/4: %d %d %d %d\n", blk,h,cnt[blk][h][0],cnt[blk][h][1],cnt[blk][h][2],cnt[blk][h][3]);
    for (int l : {0, 5, 40}) for (int v : {0, 1}) { float got = out[64*16 + l*16+v]; int fs=-1, fr=-1; for (int r=0;r<64;r++) for (int s=0;s<32;s++){ float acc=0; for(int k=0;k<4;k++) acc=fmaf(sph[s*4+k], ray[r*4+k], acc); if (acc==got){fs=s;fr=r;}} printf("blk1 lane%d v%d got %g -> sphere %d ray %d\n", l, v, got, fs, fr);}
    // try to discover the true row for lane 40, blk 0
    for (int v = 0; v < 16; v++) { float got = out[40*16+v]; int found=-1; for (int s=0;s<32;s++){ float acc=0; for(int k=0;k<4;k++) acc=fmaf(sph[s*4+k], ray[(40%32)*4+k], acc); if (acc==got) found=s;} printf("lane40 v%d -> sphere %d\n", v, found);} 
    printf("layout check: %d wrong of %d; bit-identical to the k=0..3 fmaf chain: %d\n", bad, total, bitexact);
    return 0;
}
