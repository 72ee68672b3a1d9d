// trt_common.hpp -- what the kernels of the frame producer share: launch shape of the persistent grid, work-queue
// constants, the LDS image size, the culling-table view, the diagnostic stamp macros and the two small streaming
// kernels either side of the render kernel (ordered mean over a pixel's samples, RGB8 quantisation).
#pragma once

#include "trt_device.hpp"
#include "trt_filter.h"

namespace trt
{

// threads per workgroup of the persistent kernels: 256 = 4 waves share one LDS image, 4 workgroups per CU
#ifndef TRT_BLOCK
#define TRT_BLOCK 256
#endif
constexpr int kPersistentBlock = TRT_BLOCK;

// TRT_STAMP=1: diagnostic build with s_memtime stamps between the stages of the main loop; per-stage wave-cycle
// sums go to counters[4..] (read SHARES from it, never its run time: the stamps fence the schedule).
#ifndef TRT_STAMP
#define TRT_STAMP 0
#endif
#if TRT_STAMP == 2
// -DTRT_STAMP=2: the "clock" is s101, which tools/archive/count_isa.py makes a count of executed instructions (it inserts an add at the
// head of every basic block of the compiler's assembly: tools/archive/build_isa_count.sh); the per-stage sums are then instruction counts
#define TRT_STAMP_AT(slot)                                                  \
    do                                                                      \
    {                                                                       \
        unsigned now_;                                                      \
        __builtin_amdgcn_sched_barrier(0);                                  \
        asm volatile("s_mov_b32 %0, s101" : "=s"(now_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                  \
        stamp_sum[slot] += (unsigned)(now_ - (unsigned)stamp_prev);         \
        stamp_prev = now_;                                                  \
    } while (0)
#elif TRT_STAMP
#define TRT_STAMP_AT(slot)                                                                   \
    do                                                                                       \
    {                                                                                        \
        unsigned long long now_;                                                             \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(now_)::"memory");          \
        __builtin_amdgcn_sched_barrier(0);                                                   \
        stamp_sum[slot] += now_ - stamp_prev;                                                \
        stamp_prev = now_;                                                                   \
    } while (0)
#elif defined(TRT_MARKS) && TRT_MARKS == 2
// -DTRT_MARKS=2: the ISA PROFILE of the shipping kernels (tools/isa_profile.py, tools/build_isa_profile.sh).  A stage boundary
// writes its number to m0 -- which these kernels do not use otherwise; the post-pass checks that -- and the post-pass adds, at
// the head of every basic block, the block's instructions of every kind to lane m0 of one reserved VGPR per kind
// (v_readlane / s_add / v_writelane; s100, s101 and v240... are outside what the kernels allocate).  No stamp sums in
// SGPRs, no counting instantiation: the code profiled is the shipping instantiation's own, up to the scheduling barriers.
#define TRT_STAMP_AT(slot)                                       \
    do                                                           \
    {                                                            \
        __builtin_amdgcn_sched_barrier(0);                       \
        asm volatile("s_mov_b32 m0, %0 ; MARK" ::"n"(slot));     \
        __builtin_amdgcn_sched_barrier(0);                       \
    } while (0)
#elif defined(TRT_MARKS)
// -DTRT_MARKS=1: the stage boundaries as comments in the compiler's assembly (tools/archive/isa_stage_counts.py counts the
// instructions between them); a scheduling barrier keeps each stage's instructions on its own side
#define TRT_STAMP_AT(slot)                      \
    do                                          \
    {                                           \
        __builtin_amdgcn_sched_barrier(0);      \
        asm volatile("; MARK " #slot);          \
        __builtin_amdgcn_sched_barrier(0);      \
    } while (0)
#else
#define TRT_STAMP_AT(slot) \
    do                     \
    {                      \
    } while (0)
#endif
// boundaries that only the ISA profile knows (the stamp builds keep their 24 slots)
#if defined(TRT_MARKS) && TRT_MARKS == 2
#define TRT_MARK_AT(slot) TRT_STAMP_AT(slot)
#else
#define TRT_MARK_AT(slot) \
    do                    \
    {                     \
    } while (0)
#endif
constexpr int kProfileKinds = 11, kProfileAt = 40; // counters[kProfileAt + 64 kind + slot]: the ISA profile's sums

// work units (single samples) fetched from the global queue per atomic; a returning atomic per request saturates
// a single queue word (it cost 0.8 ms per frame before pooling)
#ifndef TRT_QUEUE_CHUNK
#define TRT_QUEUE_CHUNK 256
#endif
constexpr unsigned kQueueChunkSamples = TRT_QUEUE_CHUNK;
static_assert(kQueueChunkSamples >= 64, "a chunk must hold the 64 units the lanes of a wave can ask for in one round");
// The queue of a launch is ONE word, or eight (FrameView::queue_shift = 3), one per XCD: the eight L2s keep a contended line coherent
// by passing it around, so a word that only the waves of one XCD ask stays in that XCD's L2 -- a request costs a trip to L2 instead
// of a trip through the fabric, and chunks of half the size stop costing 14 % (profiles/r05/f_ab_log.txt section 16).  Word x hands
// out the chunks (j << shift) + x, j = 0, 1, ..., to the waves of the workgroups x, x + 8, ... -- which the dispatcher deals to the
// XCDs in turn; were it to deal them otherwise, the frame would be the same and only the trips longer.  A wave's FIRST chunk is
// its own: workgroup g, wave k: chunk (j << shift) + (g mod words), j = (g >> shift) * waves per workgroup + k; start_queue_kernel
// starts every word behind those.  The words carry equal work and equal numbers of waves; nobody steals.
constexpr int kQueueXcdShift = 3, kQueueStride = 32;              // words are a 128-byte line apart
constexpr int kQueueLaneWords = (1 << kQueueXcdShift) * kQueueStride; // per lane set (the context's stream, the alternate one)
constexpr int kQueueWords = 2 * kQueueLaneWords;
#ifndef TRT_QUEUE_SMALL_DIV
#define TRT_QUEUE_SMALL_DIV 2
#endif
constexpr unsigned kQueueChunkSmall = TRT_QUEUE_CHUNK / TRT_QUEUE_SMALL_DIV; // the chunk of a launch whose queue has a word per XCD
static_assert(kQueueChunkSmall >= 64, "see kQueueChunkSamples");
#ifndef TRT_CULL_GROUP
#define TRT_CULL_GROUP 8
#endif
constexpr int kCullGroup = TRT_CULL_GROUP; // the culling table is padded to a multiple of this many entries

struct PersistentLaunch
{
    unsigned grid, block;
};

inline PersistentLaunch persistent_launch_shape(int compute_units, int blocks_per_cu, long units)
{
    long want = (units + kPersistentBlock - 1) / kPersistentBlock;
    long cap = (long)compute_units * (blocks_per_cu > 0 ? blocks_per_cu : 1);
    return PersistentLaunch{(unsigned)(want < cap ? (want > 0 ? want : 1) : cap), (unsigned)kPersistentBlock};
}

// FP32 culling table of trt_filter.h on the device
struct CullView
{
    const float *table; // padded to a multiple of kCullGroup entries of {Cx,Cy,Cz,kk}
    int padded;
    double c0x, c0y, c0z;
    float cn, rm;
};

constexpr int kLdsCameraDoubles = 16;                       // basis x,y,z (9) eye (3) -screen_distance (1) basis z * -screen_distance (3)
constexpr int kDirGridDoubles = 14, kPointGridDoubles = 9;  // sizeof(trt_dirgrid) / 8, sizeof(trt_pointgrid) / 8 (asserted in trt_rounds.hpp)

#ifdef TRT_UNIT_RENDER // kernels that are not templates have ONE home among the library's translation units: trt_render.hip
// TRT.c:1063-1066 for frames rendered with samples as work units: pixel = (((0 + s0) + s1) + ...) * (1/spp),
// samples in index order.  The scratch is sample-major, samples[(k*pixels + pixel)*3 + channel], so that for every k
// consecutive threads read consecutive doubles (a pure streaming kernel: spp*24 B read + 24 B written per pixel).
// every word of a launch's queue behind the first chunks of its workgroups (see kQueueStride)
__global__ void start_queue_kernel(unsigned int *queue, unsigned grid, unsigned waves_per_group, unsigned shift)
{
    const unsigned x = threadIdx.x, words = 1u << shift;
    if (x < words)
        queue[x * kQueueStride] = (grid > x ? (grid - x + words - 1) >> shift : 0u) * waves_per_group;
}
// ... and, being the last kernel of a frame, it leaves the frame's queue ready for a launch of the same shape (start_queue_kernel's job:
// the next frame of this context then has no kernel in front of its render kernel)
__global__ __launch_bounds__(256) void reduce_samples_kernel(const double *samples, double *out, long values, int spp, double inv_spp,
                                                             unsigned int *queue, unsigned grid, unsigned waves_per_group, unsigned shift)
{
    if (blockIdx.x == 0 && threadIdx.x < (1u << shift)) // the render kernel that used the queue has finished
        queue[threadIdx.x * kQueueStride] = (grid > threadIdx.x ? (grid - threadIdx.x + (1u << shift) - 1) >> shift : 0u) * waves_per_group;
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x; // one thread per colour channel of a pixel
    if (i >= values)
        return;
    double mean = 0.0;
    for (int k = 0; k < spp; k++)
        mean += samples[(long)k * values + i];
    out[i] = mean * inv_spp;
}

// (int)(c*255) per channel, TRT.c:1157-1163
__global__ void quantize_kernel(const double *px, long n_values, unsigned char *rgb)
{
    const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n_values)
        rgb[i] = (unsigned char)d2i(px[i] * 255);
}

#endif // TRT_UNIT_RENDER

} // namespace trt
