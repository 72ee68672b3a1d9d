// trt_tables.hip -- the candidate tables of a scene (csrc/trt_lightgrid.h, csrc/trt_raygrid.h): marking and packing kernels, builders.
// Compiled for gfx950 only, with -ffp-contract=off (see trt_device.hpp).
#include "trt_context.hpp"

using namespace trt_impl;

namespace
{

// Marking kernels of the light-space tables: one thread per cell, every sphere tested with the predicates of
// trt_lightgrid.h (+ - * / sqrt only: the host reference builders in the tests produce the same bits).
__global__ void build_dirgrid_kernel(const trt_dirgrid_disc *discs, int n, int g, int slabs, int words, unsigned long long *masks)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= (long)slabs * g * g)
        return;
    const int c = (int)(cell % g), j = (int)((cell / g) % g), slab = (int)(cell / ((long)g * g));
    for (int w = 0; w < words; w++)
    {
        unsigned long long m = 0;
        for (int b = 0; b < 64 && w * 64 + b < n; b++)
            if (trt_dirgrid_in_slab(discs + w * 64 + b, slab) && trt_dirgrid_reaches(discs + w * 64 + b, c, j))
                m |= 0x8000000000000000ull >> b;
        masks[cell * words + w] = m;
    }
}

__global__ void build_pointgrid_kernel(const trt_pointgrid_cone *cones, int n, int g, int shells, int words, unsigned long long *masks)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    if (cell >= 6L * shells * g * g)
        return;
    const int shell = (int)(cell / (6L * g * g)), face = (int)((cell / ((long)g * g)) % 6), j = (int)((cell / g) % g), c = (int)(cell % g);
    for (int w = 0; w < words; w++)
    {
        unsigned long long m = 0;
        for (int b = 0; b < 64 && w * 64 + b < n; b++)
            if (trt_pointgrid_in_shell(cones + w * 64 + b, shell, shells) && trt_pointgrid_reaches(cones + w * 64 + b, face, c, j, g))
                m |= 0x8000000000000000ull >> b;
        masks[cell * words + w] = m;
    }
}

// Mask words of a table -> list cells (trt_raygrid.h).  Lists longer than seven entries take words from the pool; when
// the pool's part is exhausted the cell says TRT_LIST_NONE and its rays sweep.
// The counter has 64 bits: it keeps counting after the pool is exhausted (exhaustion is a normal mode: the cell then says "no
// list" and its rays sweep), and a 32-bit one would wrap after 2^32 words' worth of requests and hand out words that earlier
// cells already point to.
// Called by EVERY lane of a wave (`valid`: the lane has a cell): the lanes' requests are summed and the wave takes its words
// with ONE atomic -- a request per cell on the one counter serialises in L2 (the eye's tables at 256 spheres: 49 152 cells
// with pooled lists, ~0.1 ms of nothing but that).
__device__ unsigned long long pack_cell(const unsigned long long *mask, int words, unsigned long long *pool, unsigned long long *pool_used,
                                        unsigned pool_limit, int bits, bool valid = true)
{
    const int count = valid ? trt_list_count(mask, words) : 0;
    const unsigned need = valid ? trt_list_pool_words(count, bits) : 0u;
    const int lane = (int)(threadIdx.x & 63);
    unsigned upto = need; // inclusive prefix sum over the wave
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
        const unsigned below = __shfl_up(upto, d);
        upto += lane >= d ? below : 0u;
    }
    const unsigned total = __shfl(upto, 63);
    unsigned long long base = 0;
    if (total)
    {
        if (lane == 63)
            base = atomicAdd(pool_used, (unsigned long long)total);
        base = __shfl(base, 63);
    }
    if (!valid)
        return 0ull;
    if (need == 0)
        return trt_list_pack(mask, words, count, nullptr, 0u, bits);
    const unsigned long long at = base + (upto - need);
    if (count > 0xffff || at + need > (unsigned long long)pool_limit)
        return (unsigned long long)TRT_LIST_NONE << 56;
    return trt_list_pack(mask, words, count, pool, (unsigned)at, bits);
}

__global__ void set_pool_counter_kernel(unsigned long long *counter, unsigned long long value) { *counter = value; }

__global__ void pack_lists_kernel(const unsigned long long *masks, long cells, int words, unsigned long long *lists, unsigned long long *pool,
                                  unsigned long long *pool_used, unsigned pool_limit, int bits)
{
    const long cell = (long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool valid = cell < cells;
    const unsigned long long packed = pack_cell(masks + (valid ? cell : 0) * words, words, pool, pool_used, pool_limit, bits, valid);
    if (valid)
        lists[cell] = packed;
}

// Direction tables of path-ray families (trt_raygrid.h): blockIdx.y = family.  When the table's side is a multiple of 8 a
// workgroup builds one TILE of 8 x 8 cells: every block first forms the cones of its family's apex (the same + - * / sqrt as
// the host reference builder: the same bits), thread i asks whether sphere i's cone reaches the TILE at all (trt_raygrid.h
// "marking the cells": the table's bit is tile AND cell), and then the four waves share the spheres that do -- wave q takes
// every fourth one -- each lane marking ITS cell with the cone / cell predicate of trt_lightgrid.h; the waves' masks are
// OR-ed in LDS and the first wave packs the lists.  (Every cell asking every sphere, 256 cells per block, took 0.95 ms per
// camera for the eye's two tables at 256 spheres: 192 blocks of long dependent FP64 chains.  Tiles of 16 x 16: 0.16 ms.)
// Otherwise: one thread per cell, 256 consecutive cells per block, every sphere.  `by_value`: the two families of the eye come
// as kernel arguments (they change with the camera), otherwise family blockIdx.y of `families`.
static_assert(TRT_FAMILY_TILE * TRT_FAMILY_TILE == 64, "a tile per wave");
constexpr int kSceneTilesPerBlock = 16; // the scene's sphere families: thousands of tables, 16 tiles from one set of cones
unsigned family_grid_blocks(int g, int tiles_per_block)
{
    const unsigned cells = 6u * (unsigned)g * (unsigned)g;
    return trt_family_tiled(g) ? (cells / 64u + (unsigned)tiles_per_block - 1u) / (unsigned)tiles_per_block : (cells + 255u) / 256u;
}

// kWords: 64-sphere mask words a cell can have -- 4 for scenes of up to 256 spheres (8-bit list entries), 16 for scenes of up to
// TRT_PATH_MAX_SPHERES = 1024 (16-bit entries; round 5: before, the path rays of a scene of more than 256 spheres swept).  The LDS
// is the launch's dynamic allocation (family_lds_bytes): the family's cones, the tile's reach words, the four waves' marks.
template <int kWords>
constexpr size_t family_lds_bytes()
{
    return sizeof(trt_pointgrid_cone) * 64 * kWords + sizeof(unsigned long long) * kWords + sizeof(unsigned long long) * 4 * 64 * kWords;
}

template <int kWords>
__global__ __launch_bounds__(256) void build_family_lists_kernel(const double *spheres, int n, const trt_rayfamily *families, trt_rayfamily f0,
                                                                 trt_rayfamily f1, int by_value, int g, unsigned long long *lists,
                                                                 unsigned long long *pool, unsigned long long *pool_used, unsigned pool_limit,
                                                                 int tiles_per_block)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char family_lds[];
    trt_pointgrid_cone *const cones = (trt_pointgrid_cone *)family_lds;                  // [64 kWords]
    unsigned long long *const reach = (unsigned long long *)(cones + 64 * kWords);       // [kWords]: bit k of word w: sphere 64 w + k reaches this block's tile
    unsigned long long(*const part)[64][kWords] = (unsigned long long(*)[64][kWords])(reach + kWords); // [4]: wave q's marks of the tile's cells
    const int bits = kWords > 4 ? 16 : 8; // entry width of the list cells (trt_raygrid.h)
    const trt_rayfamily F = by_value ? (blockIdx.y == 0 ? f0 : f1) : families[blockIdx.y];
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        trt_rayfamily_cone(&F, spheres + 9 * i, &cones[i]);
    __syncthreads();
    const unsigned cells = 6u * (unsigned)g * (unsigned)g;
    const int words = (n + 63) / 64 > 0 ? (n + 63) / 64 : 1;
    if (!trt_family_tiled(g))
    {
        const unsigned cell = blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = cell < cells;
        const int face = (int)(cell / ((unsigned)g * (unsigned)g)), j = (int)((cell / (unsigned)g) % (unsigned)g), c = (int)(cell % (unsigned)g);
        unsigned long long m[kWords];
#pragma unroll
        for (int w = 0; w < kWords; w++)
            m[w] = 0;
        for (int i = 0; valid && i < n; i++)
            if (trt_pointgrid_reaches(&cones[i], face, c, j, g))
                m[i >> 6] |= 0x8000000000000000ull >> (i & 63);
        const unsigned long long packed = pack_cell(m, words, pool, pool_used, pool_limit, bits, valid);
        if (valid)
            lists[(size_t)blockIdx.y * cells + cell] = packed;
        return;
    }
    // block -> tiles_per_block consecutive tiles (the scene's tables: thousands of families, a block builds a face's worth of
    // tiles from ONE set of cones; the eye's two tables: a tile per block, for the parallelism)
    const unsigned gt = (unsigned)g / TRT_FAMILY_TILE, tiles = 6u * gt * gt;
    const int lane = (int)(threadIdx.x & 63), q = (int)(threadIdx.x >> 6);
    for (unsigned tile = blockIdx.x * (unsigned)tiles_per_block; tile < tiles && tile < (blockIdx.x + 1u) * (unsigned)tiles_per_block; tile++)
    {
        const int face = (int)(tile / (gt * gt)), tj = (int)((tile / gt) % gt), tc = (int)(tile % gt);
        __syncthreads(); // the previous tile's reach[] and part[] have been read
#pragma unroll
        for (int chunk = 0; chunk < kWords / 4; chunk++) // thread t asks for spheres t, 256 + t, ...: wave q's ballot is word 4 chunk + q
        {
            const int i = 256 * chunk + (int)threadIdx.x;
            const bool reaches = i < n && trt_pointgrid_reaches(&cones[i], face, tc, tj, (int)gt);
            const unsigned long long word = __ballot(reaches); // the 64 spheres of this wave
            if (lane == 0)
                reach[4 * chunk + q] = word;
        }
        __syncthreads();
        const int j = tj * TRT_FAMILY_TILE + lane / TRT_FAMILY_TILE, c = tc * TRT_FAMILY_TILE + lane % TRT_FAMILY_TILE;
#pragma unroll
        for (int w = 0; w < kWords; w++)
        {
            unsigned long long m = 0;
            unsigned long long todo = 64 * w < n ? reach[w] & (0x1111111111111111ull << q) : 0ull; // the same in every lane of the wave: a scalar loop
            while (todo)
            {
                const int k = __builtin_ctzll(todo);
                todo &= todo - 1;
                if (trt_pointgrid_reaches(&cones[64 * w + k], face, c, j, g))
                    m |= 0x8000000000000000ull >> k;
            }
            part[q][lane][w] = m;
        }
        __syncthreads();
        if (q == 0)
        {
            unsigned long long m[kWords];
#pragma unroll
            for (int w = 0; w < kWords; w++)
                m[w] = part[0][lane][w] | part[1][lane][w] | part[2][lane][w] | part[3][lane][w];
            const unsigned cell = ((unsigned)face * (unsigned)g + (unsigned)j) * (unsigned)g + (unsigned)c;
            lists[(size_t)blockIdx.y * cells + cell] = pack_cell(m, words, pool, pool_used, pool_limit, bits);
        }
    }
}

// the builder for a scene of n spheres: 4 mask words (8-bit entries) up to 256 spheres, 16 (16-bit entries) up to 1024
void launch_family_builder(int n, dim3 grid, hipStream_t stream, const double *spheres, const trt_rayfamily *families, trt_rayfamily f0, trt_rayfamily f1,
                           int by_value, int g, unsigned long long *lists, unsigned long long *pool, unsigned long long *pool_used, unsigned pool_limit,
                           int tiles_per_block)
{
    if (n <= TRT_LIST_MAX_SPHERES)
        hipLaunchKernelGGL(build_family_lists_kernel<4>, grid, dim3(256), family_lds_bytes<4>(), stream, spheres, n, families, f0, f1, by_value, g, lists, pool,
                           pool_used, pool_limit, tiles_per_block);
    else
        hipLaunchKernelGGL(build_family_lists_kernel<TRT_PATH_MAX_SPHERES / 64>, grid, dim3(256), family_lds_bytes<TRT_PATH_MAX_SPHERES / 64>(), stream, spheres, n,
                           families, f0, f1, by_value, g, lists, pool, pool_used, pool_limit, tiles_per_block);
}

// Light-space candidate masks of every light (trt_lightgrid.h), from the context's host copy of the primitives: the
} // namespace

namespace trt_impl
{

void allow_large_lds_tables(const trt_context *ctx)
{
    (void)hipFuncSetAttribute((const void *)build_family_lists_kernel<TRT_PATH_MAX_SPHERES / 64>, hipFuncAttributeMaxDynamicSharedMemorySize, ctx->lds_limit);
}

static int build_light_grids(trt_context *ctx, const trt_cull_scene &cs)
{
    const int n = (int)(ctx->T->h_spheres.size() / 9), nd = (int)(ctx->T->h_dir.size() / 6), np = (int)(ctx->T->h_point.size() / 7);
    const int gd = ctx->dirgrid_cells, gp = ctx->pointgrid_cells, sd = std::max(ctx->dirgrid_slabs, 1), sp = std::max(ctx->pointgrid_shells, 1);
    trt::GridView &g = ctx->grids;
    g.enabled = 0;
    ctx->T->grids_built_for[0] = gd;
    ctx->T->grids_built_for[1] = gp;
    ctx->T->grids_built_for[2] = ctx->dirgrid_slabs;
    ctx->T->grids_built_for[3] = ctx->pointgrid_shells;
    if (gd < 8 || gp < 2 || nd + np == 0 || n > TRT_LIST_MAX_SPHERES_WIDE)
        return TRT_OK; // enabled = 0: the kernel sweeps
    const int bits = n > TRT_LIST_MAX_SPHERES ? 16 : 8; // entry width of the list cells
    g.list_bits = bits;
    const size_t words = (size_t)std::max((n + 63) / 64, 1), slots = (size_t)std::max(n, 1);
    const size_t dir_stride = (size_t)sd * gd * gd * words, point_stride = 6 * (size_t)sp * gp * gp * words;
    std::vector<trt_dirgrid> dg(nd);
    std::vector<trt_pointgrid> pg(np);
    std::vector<trt_dirgrid_disc> discs(slots * nd);
    std::vector<trt_pointgrid_cone> cones(slots * np);
    for (int i = 0; i < nd; i++)
    {
        const double *li = ctx->T->h_dir.data() + 6 * i;
        const double to_light[3] = {-li[0], -li[1], -li[2]}; // TRT.c:903; prepare normalises
        const double len2 = to_light[0] * to_light[0] + to_light[1] * to_light[1] + to_light[2] * to_light[2];
        if (!(len2 > 0.0) || !(len2 < 1e300))
            return TRT_OK; // a light without a direction: leave the tables off
        trt_dirgrid_prepare(ctx->T->h_spheres.data(), n, &cs, to_light, gd, sd, &dg[i], discs.data() + slots * i);
    }
    for (int i = 0; i < np; i++)
        trt_pointgrid_prepare(ctx->T->h_spheres.data(), n, &cs, ctx->T->h_point.data() + 7 * i, gp, sp, &pg[i], cones.data() + slots * i);
    HIP_TRY(ctx->T->d_dir_masks.reserve(dir_stride * nd));
    HIP_TRY(ctx->T->d_point_masks.reserve(point_stride * np));
    HIP_TRY(ctx->T->d_dirgrids.reserve(nd));
    HIP_TRY(ctx->T->d_pointgrids.reserve(np));
    HIP_TRY(ctx->T->d_discs.reserve(discs.size()));
    HIP_TRY(ctx->T->d_cones.reserve(cones.size()));
    if (nd)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_dirgrids.ptr, dg.data(), dg.size() * sizeof(trt_dirgrid), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_discs.ptr, discs.data(), discs.size() * sizeof(trt_dirgrid_disc), hipMemcpyHostToDevice));
    }
    if (np)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_pointgrids.ptr, pg.data(), pg.size() * sizeof(trt_pointgrid), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_cones.ptr, cones.data(), cones.size() * sizeof(trt_pointgrid_cone), hipMemcpyHostToDevice));
    }
    const int block = 256;
    for (int i = 0; i < nd; i++)
        hipLaunchKernelGGL(build_dirgrid_kernel, dim3((unsigned)(((size_t)sd * gd * gd + block - 1) / block)), dim3(block), 0, ctx->stream,
                           ctx->T->d_discs.ptr + slots * i, n, gd, sd, (int)words, ctx->T->d_dir_masks.ptr + dir_stride * i);
    for (int i = 0; i < np; i++)
        hipLaunchKernelGGL(build_pointgrid_kernel, dim3((unsigned)((6 * (size_t)sp * gp * gp + block - 1) / block)), dim3(block), 0, ctx->stream,
                           ctx->T->d_cones.ptr + slots * i, n, gp, sp, (int)words, ctx->T->d_point_masks.ptr + point_stride * i);
    // the kernel reads list cells: pack every table (the mask words stay for trt_read_light_grid)
    const size_t dir_cells = (size_t)sd * gd * gd, point_cells = 6 * (size_t)sp * gp * gp;
    HIP_TRY(ctx->T->d_dir_lists.reserve(dir_cells * nd));
    HIP_TRY(ctx->T->d_point_lists.reserve(point_cells * np));
    HIP_TRY(hipMemsetAsync(ctx->T->d_dir_lists.ptr, 0xFF, std::max<size_t>(dir_cells * nd, 1) * sizeof(unsigned long long), ctx->stream)); // unwritten = "no list"
    HIP_TRY(hipMemsetAsync(ctx->T->d_point_lists.ptr, 0xFF, std::max<size_t>(point_cells * np, 1) * sizeof(unsigned long long), ctx->stream));
    if (nd)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((dir_cells * nd + block - 1) / block)), dim3(block), 0, ctx->stream, ctx->T->d_dir_masks.ptr,
                           (long)(dir_cells * nd), (int)words, ctx->T->d_dir_lists.ptr, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words, bits);
    if (np)
        hipLaunchKernelGGL(pack_lists_kernel, dim3((unsigned)((point_cells * np + block - 1) / block)), dim3(block), 0, ctx->stream, ctx->T->d_point_masks.ptr,
                           (long)(point_cells * np), (int)words, ctx->T->d_point_lists.ptr, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words, bits);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipStreamSynchronize(ctx->stream)); // the caller may hand the context another stream before it renders
    g.dir = ctx->T->d_dirgrids.ptr;
    g.point = ctx->T->d_pointgrids.ptr;
    g.dir_lists = ctx->T->d_dir_lists.ptr;
    g.point_lists = ctx->T->d_point_lists.ptr;
    g.dir_stride = (unsigned)dir_cells;
    g.point_stride = (unsigned)point_cells;
    g.pool = ctx->T->d_pool.ptr;
    g.enabled = 1;
    return TRT_OK;
}

// m of the spheres' sub-families (trt_raygrid.h) for a scene of n spheres
// The automatic policy (path_patches = -1) also looks at what the tables would weigh: 6 m^2 tables per sphere and side -- 604 MB of
// cells and a 302 MB pool at 256 spheres, 32 cells, m = 2 -- and steps m down (2 -> 1 -> 0) until cells and pool fit a budget
// instead of failing in hipMalloc or on the 2^32-cell limit; an m asked for by number is taken as it is.
constexpr unsigned long long kAutoPatchBudgetBytes = 4ull << 30;
int patches_for(const trt_context *ctx, int n)
{
    if (ctx->path_patches >= 0)
        return std::min(ctx->path_patches, TRT_PATCH_MAX_M);
    if (ctx->moving_scene) // tables that live for one frame: the 24-fold cells of the patches cost more to build than they save
        return 0;
    int m = n >= TRT_PATCHES_FROM_SPHERES ? 2 : 0;
    const unsigned long long per_table = 6ull * (unsigned long long)ctx->path_g_sph * (unsigned long long)ctx->path_g_sph;
    while (m > 0 && 2ull * (unsigned long long)n * (6ull * m * m) * per_table * 12ull > kAutoPatchBudgetBytes) // 8 B a cell + half a pool word
        m--;
    return m;
}

// Direction tables of the 2NP families of the spheres of the path rays (trt_raygrid.h: P patches per sphere and their mirror
// images); the two families of the eye follow per camera (ensure_eye_tables).  The host places the families (O(NP)), the
// device forms the cones and marks and packs the cells.
static int build_path_tables(trt_context *ctx, const trt_cull_scene &cs, const double *ground)
{
    const int n = (int)(ctx->T->h_spheres.size() / 9);
    trt::GridView &g = ctx->grids;
    g.path_enabled = 0;
    g.patch_m = g.patch_count = 0;
    ctx->eye_tables_valid = false;
    ctx->T->path_built_for[0] = ctx->path_g_eye;
    ctx->T->path_built_for[1] = ctx->path_g_sph;
    ctx->T->path_built_for[2] = ctx->path_min_spheres;
    ctx->T->path_built_for[3] = ctx->path_patches;
    ctx->T->cull_scene = cs;
    memcpy(ctx->T->ground_built, ground, sizeof ctx->T->ground_built);
    const int ge = ctx->path_g_eye, gs = ctx->path_g_sph;
    if (ge < 2 || gs < 2 || n > TRT_PATH_MAX_SPHERES || n < ctx->path_min_spheres)
        return TRT_OK; // path_enabled = 0: every path ray sweeps
    trt_patchset patches;
    trt_patchset_init(&patches, patches_for(ctx, n));
    const size_t P = (size_t)patches.count, families = 2 * (size_t)n * P;
    const size_t eye_cells = 6 * (size_t)ge * ge, sph_cells = 6 * (size_t)gs * gs;
    const size_t eye_part = (size_t)kEyeSlots * 2 * eye_cells; // the eye's two tables of every slot first
    if (eye_part + families * sph_cells >= 0xffffffffull)
        return fail(TRT_ERR_CAPACITY, "path tables of %zu families x %zu cells", families, sph_cells);
    HIP_TRY(ctx->T->d_path_lists.reserve(eye_part + families * sph_cells));
    // A cell that no builder has written must not look like a list: 0xFF.. is TRT_LIST_NONE ("no list": the ray's wave sweeps),
    // whereas the bytes hipMalloc hands out could read as a pooled list at any offset of the pool -- a wild read in the render
    // kernel (round 4's "nopack" experiment).  Every cell IS written before a kernel reads it; this is the belt.
    HIP_TRY(hipMemsetAsync(ctx->T->d_path_lists.ptr, 0xFF, (eye_part + families * sph_cells) * sizeof(unsigned long long), ctx->stream));
    HIP_TRY(ctx->T->d_families.reserve(std::max<size_t>(families, 1)));
    HIP_TRY(ctx->T->d_sphere_fam.reserve(4 * (size_t)std::max(n, 1)));
    HIP_TRY(ctx->T->d_patch_rec.reserve(P * TRT_PATCH_RECORD));
    std::vector<trt_rayfamily> fam(std::max<size_t>(families, 1));
    std::vector<double> rec(4 * (size_t)std::max(n, 1)), prec(P * TRT_PATCH_RECORD);
    trt_family_consts consts;
    trt_sphere_families(ctx->T->h_spheres.data(), n, ground, &cs, &patches, fam.data(), rec.data(), &consts);
    trt_patch_records(&patches, ground, prec.data());
    HIP_TRY(hipMemcpy(ctx->T->d_patch_rec.ptr, prec.data(), prec.size() * sizeof(double), hipMemcpyHostToDevice));
    if (n)
    {
        HIP_TRY(hipMemcpy(ctx->T->d_families.ptr, fam.data(), families * sizeof(trt_rayfamily), hipMemcpyHostToDevice));
        HIP_TRY(hipMemcpy(ctx->T->d_sphere_fam.ptr, rec.data(), 4 * (size_t)n * sizeof(double), hipMemcpyHostToDevice));
        for (size_t first = 0; first < families; first += 32768) // grid.y is limited to 65535
        {
            const unsigned batch = (unsigned)std::min<size_t>(32768, families - first);
            launch_family_builder(n, dim3(family_grid_blocks(gs, kSceneTilesPerBlock), batch), ctx->stream, (const double *)ctx->T->d_spheres.ptr,
                                  (const trt_rayfamily *)ctx->T->d_families.ptr + first, trt_rayfamily{}, trt_rayfamily{}, 0, gs,
                                  ctx->T->d_path_lists.ptr + eye_part + first * sph_cells, ctx->T->d_pool.ptr, ctx->T->d_pool_used.ptr, (unsigned)ctx->T->pool_scene_words,
                                  kSceneTilesPerBlock);
        }
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipStreamSynchronize(ctx->stream));
    }
    g.path_lists = ctx->T->d_path_lists.ptr;
    g.eye_at = (unsigned)((size_t)ctx->eye_slot * 2 * eye_cells);
    g.sph_at = (unsigned)eye_part;
    g.pool = ctx->T->d_pool.ptr;
    g.sphere_fam = ctx->T->d_sphere_fam.ptr;
    g.patch_rec = ctx->T->d_patch_rec.ptr;
    g.patch_m = patches.m;
    g.patch_count = patches.count;
    g.rg2_sph = consts.rg * consts.rg;
    g.slack0 = consts.slack;
    g.g_eye = ge;
    g.g_sph = gs;
    g.path_enabled = 1;
    return TRT_OK;
}

// Every candidate table of the scene: the pool of long lists is laid out first (one part for the scene's tables, one for
// the eye's, which are rebuilt per camera), then the light tables, then the sphere families.
int build_tables(trt_context *ctx, const trt_cull_scene &cs, const double *ground)
{
    const size_t n = ctx->T->h_spheres.size() / 9, nd = ctx->T->h_dir.size() / 6, np = ctx->T->h_point.size() / 7;
    const size_t gd = (size_t)ctx->dirgrid_cells, gp = (size_t)ctx->pointgrid_cells, ge = (size_t)ctx->path_g_eye, gs = (size_t)ctx->path_g_sph;
    const size_t sd = (size_t)std::max(ctx->dirgrid_slabs, 1), sp = (size_t)std::max(ctx->pointgrid_shells, 1);
    // one pool word per cell; the many small tables of sub-families (their lists are short: that is what they are for) get
    // half a word per cell -- a list that finds no room leaves its cell TRT_LIST_NONE and its rays sweep
    const int m = patches_for(ctx, (int)n);
    const size_t sphere_cells = 2 * n * (m ? 6 * (size_t)m * m : 1) * 6 * gs * gs;
    // (scenes of more than 256 spheres: 16-bit entries, a long list takes twice the words; beyond TRT_PATH_MAX_SPHERES no sphere families)
    const size_t wide = n > TRT_LIST_MAX_SPHERES ? 2 : 1, sphere_part = n > TRT_PATH_MAX_SPHERES ? 0 : (m ? sphere_cells / 2 : sphere_cells);
    ctx->T->pool_scene_words = std::max<size_t>(1024, wide * (nd * sd * gd * gd + np * 6 * sp * gp * gp + sphere_part));
    // The eye's two tables are rebuilt for every camera, on the frame's stream: nobody can look at their counter and grow their part
    // afterwards, so it holds the longest lists there can be -- every sphere in every cell, up to 64 words (512 / 256 entries) a cell.
    // (Round 4 gave them one word per cell: enough at 64 and 256 spheres, not in a scene of 700, whose primary rays then swept.)
    const size_t per_word = wide == 2 ? 4 : 8;
    ctx->T->pool_eye_words = std::max<size_t>(1024, 2 * 6 * ge * ge * std::min<size_t>((n + per_word - 1) / per_word, 64));
    if (ctx->list_pool_cap) // trt_set_list_pool_words (tests: the pool's exhaustion -- of the scene's part and of every eye slot's)
    {
        ctx->T->pool_scene_words = std::min(ctx->T->pool_scene_words, std::max<size_t>(ctx->list_pool_cap, 1));
        ctx->T->pool_eye_words = std::min(ctx->T->pool_eye_words, std::max<size_t>(ctx->list_pool_cap, 1));
    }
    const double t0 = host_seconds();
    int rc = TRT_OK;
    // The scene's part is sized by a guess (a word per cell) and GROWN to what the builders asked for if that was more: the
    // counter keeps counting after the part is exhausted (pack_cell), so one more pass with a part of that size has room for
    // every list.  Dense scenes of many spheres need it (700 spheres: ~7 words a cell); the BASELINE configs never do.
    for (int pass = 0; pass < 2 && !rc; pass++)
    {
        if (ctx->T->pool_scene_words + kEyeSlots * ctx->T->pool_eye_words >= 0xffffffffull)
            return fail(TRT_ERR_CAPACITY, "candidate tables too large");
        ctx->grids = trt::GridView{};
        ctx->grids.list_bits = n > TRT_LIST_MAX_SPHERES ? 16 : 8; // every table of the scene: the light tables' and the families' lists alike
        HIP_TRY(ctx->T->d_pool.reserve(ctx->T->pool_scene_words + kEyeSlots * ctx->T->pool_eye_words));
        HIP_TRY(ctx->T->d_pool_used.reserve(16 * (1 + kEyeSlots)));
        HIP_TRY(hipMemsetAsync(ctx->T->d_pool_used.ptr, 0, 16 * (1 + kEyeSlots) * sizeof(unsigned long long), ctx->stream));
        rc = build_light_grids(ctx, cs);
        if (!rc)
            rc = build_path_tables(ctx, cs, ground);
        if (rc || ctx->list_pool_cap)
            break; // a capped pool (tests) stays capped
        unsigned long long asked = 0;
        HIP_TRY(hipMemcpy(&asked, ctx->T->d_pool_used.ptr, sizeof asked, hipMemcpyDeviceToHost)); // the builders have been synchronised
        if (asked <= ctx->T->pool_scene_words)
            break;
        ctx->T->pool_scene_words = (size_t)asked + 1024;
    }
    ctx->T->build_seconds = host_seconds() - t0;
    return rc;
}

// The two families of the eye (trt_raygrid.h): rebuilt on `stream` whenever the eye (or the scene) changed since they were built.
int ensure_eye_tables(trt_context *ctx, const Camera *camera, hipStream_t stream)
{
    trt::GridView &g = ctx->grids;
    if (!g.path_enabled)
        return TRT_OK;
    const double eye[3] = {camera->frame.origin.x, camera->frame.origin.y, camera->frame.origin.z};
    if (ctx->eye_tables_valid && !memcmp(eye, ctx->eye_built, sizeof eye))
        return TRT_OK;
    trt_eye_families(eye, ctx->T->ground_built, &ctx->T->cull_scene, g.eye);
    const int n = (int)(ctx->T->h_spheres.size() / 9), ge = g.g_eye;
    // this context's part of the pool: behind the scene's part and the parts of the slots before it
    const size_t pool_from = ctx->T->pool_scene_words + (size_t)ctx->eye_slot * ctx->T->pool_eye_words;
    unsigned long long *const counter = ctx->T->d_pool_used.ptr + 16 * (1 + ctx->eye_slot);
    hipLaunchKernelGGL(set_pool_counter_kernel, dim3(1), dim3(1), 0, stream, counter, (unsigned long long)pool_from);
    launch_family_builder(n, dim3(family_grid_blocks(ge, 1), 2u), stream, (const double *)ctx->T->d_spheres.ptr, (const trt_rayfamily *)nullptr, g.eye[0], g.eye[1], 1,
                          ge, ctx->T->d_path_lists.ptr + g.eye_at, ctx->T->d_pool.ptr, counter, (unsigned)(pool_from + ctx->T->pool_eye_words), 1);
    HIP_TRY(hipGetLastError());
    memcpy(ctx->eye_built, eye, sizeof eye);
    ctx->eye_tables_valid = true;
    return TRT_OK;
}

} // namespace trt_impl
