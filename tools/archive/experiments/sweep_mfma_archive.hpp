// sweep_mfma_archive.hpp -- the two matrix-core forms of the FP32 culling sweep that were measured and REJECTED in round 1
// (DESIGN.md 4.8): 32x32x2 tiles for every sweep (166 VGPRs, 3 waves/SIMD: 3.34 ms against 3.30 ms on the VALU) and
// 16x16x4 tiles for the path rays only (131 VGPRs: 2.51 ms against 2.24 ms).  Kept for the record together with the
// probes that established the rounding model of the instructions (tools/archive/mfma_probe.hip, tools/archive/mfma16_probe.hip).
// NOT part of the library any more and not compiled by the build: these functions used the LdsImage fields a_xy / a_zk /
// a_zk_dir / mfma16_wave of the round-1 kernel (see git history of csrc/trt_rounds.hpp at 5726ee7 for the call sites).
#pragma once
typedef float f16v __attribute__((ext_vector_type(16)));

// v_permlane32_swap a, b:  a <- [a.lo32, b.lo32],  b <- [a.hi32, b.hi32]   (lo32 = lanes 0-31).  Through inline asm:
// hipcc (ROCm 7.2) folds MFMAs fed by the two results of __builtin_amdgcn_permlane32_swap into one (tools/mfma_probe).
// The s_nops cover the data hazards around it (VALU result -> permlane read, permlane result -> MFMA/VALU read):
// the compiler's hazard recogniser cannot look inside the asm statement (cdna_hip_programming.md 5.7).
TRT_DEV void lane_swap32(float &a, float &b)
{
    asm volatile("s_nop 4\n\tv_permlane32_swap_b32 %0, %1\n\ts_nop 4" : "+v"(a), "+v"(b));
}

// Phase 1 for one chunk of 64 spheres and the 64 rays of the wave, on the MATRIX cores.
// The culling test is a small dense contraction: for sphere row j and ray column i (K = 4)
//     cd[j][i] = (Cx,Cy,Cz,kk) . (dx,dy,dz, 0)              cw[j][i] = (Cx,Cy,Cz,kk) . (wx,wy,wz,-1) - thr_i
// and the verdict is the sign of fma(cd,cd,cw) (trt_filter.h).  v_mfma_f32_32x32x2_f32 evaluates exactly that FMA chain
// (k = 0..3, FP32, checked bit for bit against fmaf by tools/mfma_probe), on a pipe of its own, so the VALU is left with
// one FMA and one v_alignbit per (ray, sphere) -- v_alignbit alone for a fixed direction, whose cd^2 is in the table.
// Tiles: 2 blocks of 32 spheres x 2 blocks of 32 rays.  A operand of a sphere block: lane l holds A[l%32][k = l/32];
// B operand of a ray block: lane l holds B[k = l/32][l%32], made from the per-lane ray constants with one
// v_permlane32_swap per pair; D: lane l holds ray l%32 and, in register v, sphere row (v/4)*8 + (l/32)*4 + v%4.
// Returns, for THIS lane's ray, two 32-bit words (rows held by lanes < 32 / >= 32): bit 31-p of word h set = candidate
// sphere 32*(p>>4) + ((p&15)>>2)*8 + 4*h + (p&3) of the chunk.
template <bool FIXED>
TRT_DEV void sweep64_mfma(const float *a_xy, const float *a_zk, int lane, const trt_ray_filter &f, unsigned &half0, unsigned &half1)
{
    const float axy[2] = {a_xy[lane], a_xy[64 + lane]}, azk[2] = {a_zk[lane], a_zk[64 + lane]};
    float wxy[2] = {f.wx, f.wy}, wz1[2] = {f.wz, -1.0f}, thr[2] = {f.neg_thr, f.neg_thr};
    lane_swap32(wxy[0], wxy[1]);
    lane_swap32(wz1[0], wz1[1]);
    lane_swap32(thr[0], thr[1]); // thr[r] = -thr of ray l%32 + 32r in every lane
    float dxy[2] = {f.dx, f.dy}, dz0[2] = {f.dz, 0.0f};
    if (!FIXED)
    {
        lane_swap32(dxy[0], dxy[1]);
        lane_swap32(dz0[0], dz0[1]);
    }
    float word[2];
#pragma unroll
    for (int r = 0; r < 2; r++)
    {
        unsigned bits = ~0u;
#pragma unroll
        for (int s = 0; s < 2; s++)
        {
            f16v cw;
#pragma unroll
            for (int v = 0; v < 16; v++)
                cw[v] = thr[r];
            cw = __builtin_amdgcn_mfma_f32_32x32x2f32(axy[s], wxy[r], cw, 0, 0, 0);
            cw = __builtin_amdgcn_mfma_f32_32x32x2f32(azk[s], wz1[r], cw, 0, 0, 0);
            if (!FIXED)
            {
                f16v cd = {0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f, 0.0f};
                cd = __builtin_amdgcn_mfma_f32_32x32x2f32(axy[s], dxy[r], cd, 0, 0, 0);
                cd = __builtin_amdgcn_mfma_f32_32x32x2f32(azk[s], dz0[r], cd, 0, 0, 0);
#pragma unroll
                for (int v = 0; v < 16; v++)
                {
                    const float m = __builtin_fmaf(cd[v], cd[v], cw[v]);
                    bits = __builtin_amdgcn_alignbit(bits, __builtin_bit_cast(unsigned, m), 31);
                }
            }
            else
            {
#pragma unroll
                for (int v = 0; v < 16; v++)
                {
                    const float m = cw[v]; // NB: __builtin_bit_cast applied to the element lvalue cw[v] itself reads element 0
                    bits = __builtin_amdgcn_alignbit(bits, __builtin_bit_cast(unsigned, m), 31);
                }
            }
        }
        word[r] = __builtin_bit_cast(float, ~bits); // set = candidate
    }
    lane_swap32(word[0], word[1]); // both halves of the rows of this lane's own ray
    half0 = __builtin_bit_cast(unsigned, word[0]);
    half1 = __builtin_bit_cast(unsigned, word[1]);
}

#if TRT_SWEEP_MFMA == 2
typedef float f4v __attribute__((ext_vector_type(4)));

// Phase 1 of a PATH ray on the matrix cores, 16x16x4 tiles (experiment).  v_mfma_f32_16x16x4_f32 is the FMA
// chain over k = 0..3 starting from C (tools/mfma16_probe: bit-identical), i.e. the order of trt_filter_sign_mfma.
// Rows = spheres (A straight from the {Cx,Cy,Cz,kk} table: lane l reads component l/16 of sphere l%16 of the tile),
// columns = rays (B from the rays' vectors staged in LDS once per trace: lane l reads component l/16 of ray 16t + l%16),
// D: lane l, register v = sphere 4*(l/16) + v of the tile, ray 16t + l%16.  Each lane packs the sign bits it holds into
// their final positions of the ray's candidate word; the ray's own lane ORs the four partial words via LDS.
TRT_DEV void mfma16_stage_ray(float *wave, int lane, const trt_ray_filter &f)
{
    float *r = wave + lane * kMfma16RayFloats;
    r[0] = f.wx, r[1] = f.wy, r[2] = f.wz, r[3] = -1.0f;
    r[4] = f.dx, r[5] = f.dy, r[6] = f.dz, r[7] = 0.0f;
    r[8] = f.neg_thr;
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// bit 63-j set = sphere j of the chunk is REJECTED for this lane's ray
TRT_DEV unsigned long long mfma16_sweep(const float4 *table, float *wave, int lane)
{
    const int q = lane >> 4, c = lane & 15;
    float a[4];
#pragma unroll
    for (int s = 0; s < 4; s++)
        a[s] = ((const float *)table)[(16 * s + c) * 4 + q];
    unsigned long long *xch = (unsigned long long *)(wave + 64 * kMfma16RayFloats);
    const unsigned shift = 12u - 4u * (unsigned)q;
#pragma unroll
    for (int t = 0; t < 4; t++)
    {
        const float *ray = wave + (16 * t + c) * kMfma16RayFloats;
        const float bw = ray[q], bd = ray[4 + q], thr = ray[8];
        unsigned word[2] = {0u, 0u};
#pragma unroll
        for (int s = 0; s < 4; s++)
        {
            f4v cw = {thr, thr, thr, thr}, cd = {0.0f, 0.0f, 0.0f, 0.0f};
            cw = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bw, cw, 0, 0, 0);
            cd = __builtin_amdgcn_mfma_f32_16x16x4f32(a[s], bd, cd, 0, 0, 0);
            unsigned w = word[s >> 1];
#pragma unroll
            for (int v = 0; v < 4; v++)
            {
                const float m = __builtin_fmaf(cd[v], cd[v], cw[v]);
                w = __builtin_amdgcn_alignbit(w, __builtin_bit_cast(unsigned, m), 31);
            }
            word[s >> 1] = (s & 1) ? w : w << 12;
        }
        xch[lane * 4 + t] = ((unsigned long long)(word[0] << shift) << 32) | (word[1] << shift);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    const int own = lane >> 4;
    const unsigned long long rejected = xch[c * 4 + own] | xch[(c + 16) * 4 + own] | xch[(c + 32) * 4 + own] | xch[(c + 48) * 4 + own];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the next chunk overwrites the exchange buffer
    return rejected;
}
#endif
