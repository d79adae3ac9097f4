// ------------------------------------------------------------------------------------------
// Tall-K weight gradients: C [M,N] (+ row sums) = A^T-like product with M, N <= 304 and K in the
// hundred thousands (dW = dY^T X over all rows of the batch, both operands row-major with the
// contracted index on the rows).  A workgroup of 8 waves owns a K chunk and one HALF of the output
// columns (up to 304 x 160) and keeps it in registers: 4 x 2 waves, each 5 x 5 tiles of
// v_mfma_f32_16x16x4_f32 = 100 accumulator registers per lane, two waves per SIMD.  The A operand is
// therefore read twice (by the two halves of a chunk, placed on the same XCD so that the second read
// is an L2 hit) and B once - the panel kernel reads B once per 64-row panel of the output, five times
// for M = 300.  Operand rows go straight to LDS (16 bytes per lane, no staging registers) through a
// ring of three stages; row strides of 304 and 176 floats (= 16 mod 32) put the four k groups of a
// wave on disjoint banks.  The product is taken as (X^T dY)^T: a lane is an output row m and its four
// registers four consecutive columns n, so the slab is written 16 bytes per lane.  Column N of the
// B tile is preset to 1: the row sums (bias gradient) come out of the same MFMAs.  Slabs are summed
// by gemm_splitk_reduce_kernel, which also adds the last K % 16 rows.
// ------------------------------------------------------------------------------------------
#include <mutex>

#include "gemm_common.h"

namespace psvae_gemm {

__global__ __launch_bounds__(TK_THREADS, 2) void gemm_tallk_kernel(TallKArgs g) {
    extern __shared__ __align__(16) float tk_lds[];   // 3 stages x (A tile [16][304] | B tile [16][176])
    constexpr int TILE_A = TK_BK * TK_LDA, TILE_B = TK_BK * TK_LDB, STAGE = TILE_A + TILE_B;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int jl = lane & 15, kq = lane >> 4;
    // blocks b and b + 8 (the two halves of one chunk) run on the same XCD
    int z = blockIdx.x, half = 0;
    if (g.halves == 2) half = (z >> 3) & 1, z = (z >> 4) * 8 + (z & 7);
    if (z >= g.chunks) return;   // (the grid is padded to whole groups of 16 blocks)
    const int k_begin = z * g.k_chunk, k_end = min(g.K, k_begin + g.k_chunk);
    const int n_lo = half * TK_HALF;                                  // first output column of this half
    const int n_cols = min(g.N, n_lo + TK_HALF) - n_lo;               // columns of B it loads (may be <= 0)
    const int ntm = (g.M + 15) >> 4, ntn = (min(g.n_ext, n_lo + TK_HALF) - n_lo + 15) >> 4;
    // Every wave multiplies a full 5 x 5 block of tiles, so the k loop has no branch: the last wave
    // row / column slides back over its neighbour's tiles (19 tiles: 0-4, 5-9, 10-14, 14-18) and the
    // duplicates are dropped at the store.
    const int own_m = (wave & 3) * TK_T, own_n = (wave >> 2) * TK_T;
    const int mt0 = max(0, min(own_m, ntm - TK_T)), nt0 = max(0, min(own_n, ntn - TK_T));

    // columns the loads never write: zero, and 1 at column N of the B tile when row sums are asked for
    for (int i = tid; i < TK_STAGES * STAGE; i += TK_THREADS) {
        const int r = i % STAGE;
        if (r < TILE_A) {
            if (r % TK_LDA >= g.M) tk_lds[i] = 0.f;
        } else {
            const int col = (r - TILE_A) % TK_LDB;
            if (col >= n_cols) tk_lds[i] = (n_lo + col == g.N && g.n_ext > g.N) ? 1.f : 0.f;
        }
    }
    // One k-tile of both operands -> stage, each row in pieces of up to 1 KiB (64 lanes x 16 B) straight
    // to LDS: a wave takes one piece index of rows r0, r0+4, r0+8, r0+12 of A and rows wave, wave+8 of B.
    const int piece = wave & 1, r0 = wave >> 1, ca = piece * 256 + lane * 4, cb = lane * 4;
    const bool a_on = ca < g.M, b_on = cb < n_cols;
    auto issue = [&](int stage, int k0) {
        float *dst = tk_lds + stage * STAGE;
#pragma unroll
        for (int n = 0; n < 4; ++n) {   // (row base: wave-uniform; lane part: ca)
            const float *ra = g.A + (long long)(k0 + r0 + 4 * n) * g.lda;
            if (a_on)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(ra + ca),
                                                 (__attribute__((address_space(3))) void *)(dst + (r0 + 4 * n) * TK_LDA + piece * 256),
                                                 16, 0, 0);
        }
#pragma unroll
        for (int n = 0; n < 2; ++n) {
            const float *rb = g.B + (long long)(k0 + wave + 8 * n) * g.ldb + n_lo;
            if (b_on)
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(rb + cb),
                                                 (__attribute__((address_space(3))) void *)(dst + TILE_A + (wave + 8 * n) * TK_LDB),
                                                 16, 0, 0);
        }
    };
    f32x4 acc[TK_T][TK_T];
#pragma unroll
    for (int i = 0; i < TK_T; ++i)
#pragma unroll
        for (int j = 0; j < TK_T; ++j) acc[i][j] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // operand fragments of one k-step (4 k rows): 5 + 5 registers, two sets in flight
    const int a_off = kq * TK_LDA + mt0 * 16 + jl, b_off = TILE_A + kq * TK_LDB + nt0 * 16 + jl;
    auto fetch = [&](float (&av)[TK_T], float (&bv)[TK_T], const float *st, int ks) {
#pragma unroll
        for (int i = 0; i < TK_T; ++i) av[i] = st[a_off + ks * 4 * TK_LDA + 16 * i];
#pragma unroll
        for (int j = 0; j < TK_T; ++j) bv[j] = st[b_off + ks * 4 * TK_LDB + 16 * j];
    };
    auto multiply = [&](const float (&av)[TK_T], const float (&bv)[TK_T]) {
#pragma unroll
        for (int i = 0; i < TK_T; ++i)
#pragma unroll
            for (int j = 0; j < TK_T; ++j)
                acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bv[j], av[i], acc[i][j], 0, 0, 0);
    };

    // Ring of three stages with ONE barrier per k-tile, placed in the middle of the tile: the tile
    // after this one was requested a whole tile earlier, so the last k-step of a tile can already
    // fetch the first fragments of the next one, and the stage two tiles ahead - last read before
    // every wave reached this barrier - is free to be refilled.
    // A load that goes straight to LDS is counted by the issuing wave's vector-memory counter only and
    // the compiler does not wait for it at a barrier (a workgroup-scope fence covers LDS instructions,
    // not this path), so every wave waits for its own outstanding loads - exactly the pieces of the
    // next tile - before it signals the barrier; the other waves read them right after it.
    const int n_tiles = max(0, k_end - k_begin) / TK_BK;
    if (n_tiles > 0) issue(0, k_begin);
    if (n_tiles > 1) issue(1, k_begin + TK_BK);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float a0[TK_T], b0[TK_T], a1[TK_T], b1[TK_T];
    fetch(a0, b0, tk_lds, 0);
    int cur = 0;
    for (int t = 0; t < n_tiles; ++t) {
        const float *st = tk_lds + cur * STAGE;
        const int nxt = cur == TK_STAGES - 1 ? 0 : cur + 1, nxt2 = nxt == TK_STAGES - 1 ? 0 : nxt + 1;
        fetch(a1, b1, st, 1);
        multiply(a0, b0);
        fetch(a0, b0, st, 2);
        multiply(a1, b1);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (t + 2 < n_tiles) issue(nxt2, k_begin + (t + 2) * TK_BK);
        fetch(a1, b1, st, 3);
        multiply(a0, b0);
        fetch(a0, b0, tk_lds + nxt * STAGE, 0);   // (after the last tile: a stage nobody writes any more)
        multiply(a1, b1);
        cur = nxt;
    }
    // slab: lane = row m of the tile, registers = columns n0 .. n0+3
    float *slab = g.ws + (long long)z * g.M * g.ws_ld;
#pragma unroll
    for (int i = 0; i < TK_T; ++i) {
        const int m = (mt0 + i) * 16 + jl;
        if (mt0 + i < own_m || m >= g.M) continue;
#pragma unroll
        for (int j = 0; j < TK_T; ++j) {
            const int n0 = n_lo + (nt0 + j) * 16 + 4 * kq;
            if (nt0 + j >= own_n && nt0 + j < ntn && n0 < g.ws_ld)
                *reinterpret_cast<float4 *>(slab + (long long)m * g.ws_ld + n0) =
                    make_float4(acc[i][j][0], acc[i][j][1], acc[i][j][2], acc[i][j][3]);
        }
    }
}


// K chunks of the tall-K kernel: one workgroup per CU when K allows it (two workgroups per chunk when
// the output has more than 160 columns), at least 8 k-tiles each
int tallk_split(int K, int n_ext) {
    const int cap = n_ext > TK_HALF ? 128 : 256;
    int s = K / (8 * TK_BK);
    return s < 1 ? 1 : (s > cap ? cap : s);
}

// 92 160 bytes of dynamic LDS: above the default limit, so the attribute is raised once per device
int launch_tallk(const TallKArgs &t, hipStream_t s) {
    const size_t lds = (size_t)TK_STAGES * TK_BK * (TK_LDA + TK_LDB) * sizeof(float);
    static bool attr_set[64] = {};   // per device, under attr_mu (forward and autograd threads both launch)
    static std::mutex attr_mu;
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess) {
        psvae_set_error("psvae_gemm_f32 (tall-K): hipGetDevice: %s", hipGetErrorString(e));
        return (int)e;
    }
    {
        std::lock_guard<std::mutex> lock(attr_mu);
        if (device < 0 || device >= 64 || !attr_set[device]) {
            e = hipFuncSetAttribute((const void *)gemm_tallk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            if (e != hipSuccess) {   // not marked: the next call tries again
                psvae_set_error("psvae_gemm_f32 (tall-K): raising the dynamic LDS limit to %zu bytes on device %d: %s", lds,
                                device, hipGetErrorString(e));
                return (int)e;
            }
            if (device >= 0 && device < 64) attr_set[device] = true;
        }
    }
    // with two halves the chunk count is padded to a multiple of 8 (blocks b, b + 8 pair up)
    const int chunks_launched = t.halves == 2 ? psvae_cdiv(t.chunks, 8) * 8 : t.chunks;
    hipLaunchKernelGGL(gemm_tallk_kernel, dim3(chunks_launched * t.halves), dim3(TK_THREADS), lds, s, t);
    PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (tall-K)");
    return PSVAE_OK;
}

}  // namespace psvae_gemm
