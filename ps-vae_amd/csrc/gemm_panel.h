// The panel kernel of psvae_gemm_f32 (see gemm.hip for the overview) and its launcher template.
#pragma once
#include "gemm_common.h"

#include <type_traits>

namespace psvae_gemm {

// ------------------------------------------------------------------------------------------
// Fast path: 128 x (32*TN) panel per workgroup, vector loads.
// ------------------------------------------------------------------------------------------
template <int BX, int BK, bool KCONTIG, int NT = psvae_gemm::THREADS>
struct TileLoader {
    static constexpr int THREADS = NT;                                // threads that share the tile
    static constexpr int ITEMS = BX * BK / 4;                         // float4 items per tile
    static constexpr int PER_THREAD = (ITEMS + THREADS - 1) / THREADS;
    static constexpr int STRIDE = KCONTIG ? BX + 1 : BX + 4;          // LDS row stride (floats)

    // element (x, k) of the operand lives at p[x*sx + k*sk]; KCONTIG: sk == 1, else sx == 1.
    // MASK: the element counts as 0 where pm[same offset] <= 0.  ONES: a virtual column of
    // ones at x == ones_x (only for !KCONTIG).
    //
    // load() ONLY issues the global loads (unconditionally, from a clamped address) and
    // store() applies bounds / mask / ones while writing LDS one k-tile later.  Anything that
    // consumes a loaded value before the MFMA loop — or a branch around a load — makes the
    // compiler wait for the loads right there and the prefetch no longer overlaps the MFMAs.
    struct Coord {
        int x, k;
        bool in_tile;
    };
    __device__ static __forceinline__ Coord coord(int i, int tid, int x0, int k0) {
        const int item = tid + THREADS * i;
        Coord c;
        if (KCONTIG) {
            c.x = x0 + item / (BK / 4), c.k = k0 + 4 * (item % (BK / 4));
        } else {
            c.x = x0 + 4 * (item % (BX / 4)), c.k = k0 + item / (BX / 4);
        }
        c.in_tile = (ITEMS % THREADS == 0) || item < ITEMS;
        return c;
    }

    template <bool MASK>
    __device__ static __forceinline__ void load(float4 (&r)[PER_THREAD], float4 (&rm)[MASK ? PER_THREAD : 1],
                                                const float *__restrict__ p, const float *__restrict__ pm,
                                                long long sx, long long sk, int x0, int k0, int X, int k_end,
                                                int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const Coord c = coord(i, tid, x0, k0);
            const bool ok = c.in_tile && c.x < X && c.k < k_end;
            const long long off = ok ? (KCONTIG ? (long long)c.x * sx + c.k : (long long)c.k * sk + c.x) : 0;
            r[i] = *reinterpret_cast<const float4 *>(p + off);
            if (MASK) rm[i] = *reinterpret_cast<const float4 *>(pm + off);
        }
    }

    // INTERIOR: the whole tile lies inside the operand (no bounds to apply, no ones column): the
    // staging then costs its LDS writes only, which matters because vector instructions beside
    // other waves' MFMA loops issue several times slower than alone.
    template <bool MASK, bool ONES, bool INTERIOR = false>
    __device__ static __forceinline__ void store(float *T, const float4 (&r)[PER_THREAD],
                                                 const float4 (&rm)[MASK ? PER_THREAD : 1], int x0, int k0, int X,
                                                 int k_end, int ones_x, int tid) {
#pragma unroll
        for (int i = 0; i < PER_THREAD; ++i) {
            const Coord c = coord(i, tid, x0, k0);
            if (!c.in_tile) continue;
            const bool ok = INTERIOR || (c.x < X && c.k < k_end);
            float4 v = r[i];
            if (MASK) {
                const float4 m = rm[i];
                v.x = m.x > 0.f ? v.x : 0.f, v.y = m.y > 0.f ? v.y : 0.f;
                v.z = m.z > 0.f ? v.z : 0.f, v.w = m.w > 0.f ? v.w : 0.f;
            }
            v.x = ok ? v.x : 0.f, v.y = ok ? v.y : 0.f, v.z = ok ? v.z : 0.f, v.w = ok ? v.w : 0.f;
            if (ONES && !KCONTIG && !INTERIOR) v.x = (c.k < k_end && c.x == ones_x) ? 1.f : v.x;
            const int item = tid + THREADS * i;
            if (KCONTIG) {
                const int kq = item % (BK / 4), x = item / (BK / 4);
                float *d = T + (4 * kq) * STRIDE + x;
                d[0] = v.x, d[STRIDE] = v.y, d[2 * STRIDE] = v.z, d[3 * STRIDE] = v.w;
            } else {
                const int xq = item % (BX / 4), k = item / (BX / 4);
                *reinterpret_cast<float4 *>(T + k * STRIDE + 4 * xq) = v;
            }
        }
    }
};

// WN = 1: the 4 waves stack along M (128-row panel, 32*TN columns).  WN = 2: a 2 x 2 wave grid
// (64 rows x 64*TN columns), for products whose M fits 64-row panels with less padding and, with
// TN = 1 (64 x 64 tiles, tile64_inst.hip), for products too small to fill the chip with panels.
//
// KG > 1: the latency form for the products of a small batch - a 32 x 32 tile per workgroup, its four waves
// splitting K.  Every kernel starts with cold caches and a CU has a bounded number of requests in flight, so a
// short product on few CUs takes as long as its operand bytes per CU take to arrive: small tiles put the same
// bytes on four times as many CUs.  All threads stage the 64-deep k-tiles (five of them requested ahead, which
// is all of K = 320 at once), wave q multiplies k-steps 16 q .. 16 q + 15 of every tile, the four partial tiles
// are summed through LDS in a fixed order and wave 0 runs the epilogue.
template <int TN, int BK, bool A_KC, bool B_KC, bool MASK, bool ONES, int WN = 1, bool CLS = false, int KG = 1>
__global__ __launch_bounds__(THREADS, KG > 1 ? 1 : ((MASK || WN > 1 || TN > 5) ? 2 : PSVAE_PANEL_OCC)) void gemm_panel_kernel(GemmArgs g) {
    constexpr int WM = KG > 1 ? 1 : 4 / WN, BM = 32 * WM, BN = 32 * TN * WN;
    using LA = TileLoader<BM, BK, A_KC>;
    using LB = TileLoader<BN, BK, B_KC>;
    constexpr int A_FLOATS = BK * LA::STRIDE, B_FLOATS = BK * LB::STRIDE;
    constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    // two LDS stages where two workgroups of them still fit a CU's 160 KiB: always for the 64 x 64
    // tiles of small products (one workgroup per CU at best, so nothing else hides a k-tile's loads)
    constexpr int NBUF = ((PSVAE_PANEL_NBUF == 2 || (WN == 2 && TN == 1)) && 2 * STAGE_FLOATS * 4 <= 76 * 1024) ? 2 : 1;
    constexpr int EPI_STRIDE = 36, EPI_FLOATS = 4 * 32 * EPI_STRIDE;   // one 32x32 tile per wave
    constexpr int RED_FLOATS = (KG - 1) * 16 * 64;   // partial tiles of the waves 1 .. KG-1
    constexpr int LDS_FLOATS = KG > 1 ? (EPI_FLOATS + RED_FLOATS > 2 * STAGE_FLOATS ? EPI_FLOATS + RED_FLOATS : 2 * STAGE_FLOATS)
                                      : (NBUF * STAGE_FLOATS > EPI_FLOATS ? NBUF * STAGE_FLOATS : EPI_FLOATS);
    __shared__ __align__(16) float smem[LDS_FLOATS];
    float *As = smem, *Bs = smem + A_FLOATS;

    int tile_m, tile_n, z;
    if (!tile_of_block(g, tile_m, tile_n, z)) return;
    const int m0 = tile_m * BM, n0 = tile_n * g.bn;
    const int k_begin = z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = KG > 1 ? 0 : tid >> 6;
    const int wave_m = wave % WM, wave_n = wave / WM;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ones_x = g.N;
    const int n_lim = min(g.N, n0 + g.bn);   // columns of B this tile may read
    // accumulator tiles of this wave that hold at least one real column (wave-uniform)
    const int tn_live = (min(g.n_ext, n0 + g.bn) - n0 - wave_n * TN * 32 + 31) >> 5;

    f32x16 acc[TN];
#pragma unroll
    for (int j = 0; j < TN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;

    if constexpr (KG == 1) {
    float4 ra[LA::PER_THREAD], rb[LB::PER_THREAD], rma[MASK ? LA::PER_THREAD : 1], rmb[1];
    if (k_begin < k_end) {
        LA::template load<MASK>(ra, rma, g.A, g.amask, g.sa_m, g.sa_k, m0, k_begin, g.M, k_end, tid);
        LB::template load<false>(rb, rmb, g.B, nullptr, g.sb_n, g.sb_k, n0, k_begin, n_lim, k_end, tid);
    }
    // operand tiles without an edge in x (rows of A / columns of B incl. the ones column)
    const bool a_inside = m0 + BM <= g.M, b_inside = n0 + BN <= n_lim;
    const float *a_rd = As + lhi * LA::STRIDE + wave_m * 32 + l31;
    const float *b_rd = Bs + lhi * LB::STRIDE + wave_n * TN * 32 + l31;

    // registers -> LDS stage `buf` for the k-tile that starts at k0
    auto stage = [&](int buf, int k0) {
        float *as = As + buf * STAGE_FLOATS, *bs = Bs + buf * STAGE_FLOATS;
        if (k0 + BK <= k_end && a_inside)   // wave-uniform
            LA::template store<MASK, false, true>(as, ra, rma, m0, k0, g.M, k_end, -1, tid);
        else
            LA::template store<MASK, false>(as, ra, rma, m0, k0, g.M, k_end, -1, tid);
        if (k0 + BK <= k_end && b_inside)
            LB::template store<false, ONES, true>(bs, rb, rmb, n0, k0, n_lim, k_end, ones_x, tid);
        else
            LB::template store<false, ONES>(bs, rb, rmb, n0, k0, n_lim, k_end, ones_x, tid);
    };
    // the MFMAs of one k-tile from stage `buf`: operand registers for k-step s+1 are fetched from
    // LDS while the MFMAs of step s run; the scheduling barriers keep the compiler from sinking the
    // reads next to their use
    // The 64 x 64 tiles sum each k-tile into a fresh accumulator and add it to the running total
    // (two levels: chains of BK terms instead of K), which keeps their rounding error at that of a
    // blocked CPU product; 16 vector adds per k-tile beside its 32 MFMAs.
    constexpr bool TWO_LEVEL = (WN == 2 && TN == 1);
    auto multiply = [&](int buf, int k0) {
        f32x16 total0;
        if constexpr (TWO_LEVEL) {
            total0 = acc[0];
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] = 0.f;
        }
        const float *ar = a_rd + buf * STAGE_FLOATS, *br = b_rd + buf * STAGE_FLOATS;
        float a_cur = ar[0], b_cur[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_cur[j] = br[j * 32];
        const int k_live = k_end - k0;   // the last k-tile of a chunk may be partly empty
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            if (kk > 0 && kk >= k_live) break;   // wave-uniform
            float a_nxt = 0.f, b_nxt[TN];
            if (kk + 2 < BK) {
                a_nxt = ar[(kk + 2) * LA::STRIDE];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_nxt[j] = br[(kk + 2) * LB::STRIDE + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j)
                // the last column tile of a wide product may leave up to two tiles empty (N = 328 as
                // 2 x 192, 600 as 3 x 224); tiles of 5 or fewer never do for this model's widths
                if (TN < 6 || j + 2 < TN || j < tn_live)
                    acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if (kk + 2 < BK) {
                a_cur = a_nxt;
#pragma unroll
                for (int j = 0; j < TN; ++j) b_cur[j] = b_nxt[j];
            }
        }
        if constexpr (TWO_LEVEL) {
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[0][r] += total0[r];
        }
    };
    if constexpr (NBUF == 1) {
        // one LDS stage: write it, barrier, request the next k-tile, multiply, barrier
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            stage(0, k0);
            __syncthreads();
            if (k0 + BK < k_end) {
                LA::template load<MASK>(ra, rma, g.A, g.amask, g.sa_m, g.sa_k, m0, k0 + BK, g.M, k_end, tid);
                LB::template load<false>(rb, rmb, g.B, nullptr, g.sb_n, g.sb_k, n0, k0 + BK, n_lim, k_end, tid);
            }
            multiply(0, k0);
            __syncthreads();
        }
    } else {
        // two LDS stages: the next k-tile is requested before the current one is multiplied and
        // written to the other stage right after, so a k-tile costs ONE barrier and the staging
        // stores of a wave run beside the other waves' MFMAs
        if (k_begin < k_end) stage(0, k_begin);
        __syncthreads();
        int cur = 0;
        for (int k0 = k_begin; k0 < k_end; k0 += BK) {
            const bool more = k0 + BK < k_end;   // block-uniform
            if (more) {
                LA::template load<MASK>(ra, rma, g.A, g.amask, g.sa_m, g.sa_k, m0, k0 + BK, g.M, k_end, tid);
                LB::template load<false>(rb, rmb, g.B, nullptr, g.sb_n, g.sb_k, n0, k0 + BK, n_lim, k_end, tid);
            }
            multiply(cur, k0);
            if (more) stage(cur ^ 1, k0 + BK);
            __syncthreads();
            cur ^= 1;
        }
    }
    } else {
        static_assert(KG == 1 || (KG == 4 && TN == 1 && WN == 1 && !CLS), "latency form: 32 x 32 tiles, four waves");
        constexpr int PA = LA::PER_THREAD, PB = LB::PER_THREAD;
        const int grp = tid >> 6;
        const bool a_inside = m0 + BM <= g.M, b_inside = n0 + BN <= n_lim;
        constexpr int D = 5, KSTEPS = BK / KG;
        float4 fa[D][PA], fb[D][PB], fm[D][MASK ? PA : 1], none[1];
        const int n_tiles = (k_end - k_begin + BK - 1) / BK;
        if (k_begin < k_end) {   // (block-uniform; an empty K range may come with null operands)
#pragma unroll
            for (int d = 0; d < D; ++d) {
                LA::template load<MASK>(fa[d], fm[d], g.A, g.amask, g.sa_m, g.sa_k, m0, k_begin + d * BK, g.M, k_end, tid);
                LB::template load<false>(fb[d], none, g.B, nullptr, g.sb_n, g.sb_k, n0, k_begin + d * BK, n_lim, k_end, tid);
            }
        }
        const float *a_rd = As + (lhi + grp * KSTEPS) * LA::STRIDE + l31;
        const float *b_rd = Bs + (lhi + grp * KSTEPS) * LB::STRIDE + l31;
        for (int t0 = 0; t0 < n_tiles; t0 += D) {
#pragma unroll
            for (int d = 0; d < D; ++d) {
                const int t = t0 + d, k0 = k_begin + t * BK;
                if (t >= n_tiles) break;   // block-uniform
                // a tile goes to the LDS stage of its parity: one barrier per tile
                float *as = As + (t & 1) * STAGE_FLOATS, *bs = Bs + (t & 1) * STAGE_FLOATS;
                if (k0 + BK <= k_end && a_inside)
                    LA::template store<MASK, false, true>(as, fa[d], fm[d], m0, k0, g.M, k_end, -1, tid);
                else
                    LA::template store<MASK, false>(as, fa[d], fm[d], m0, k0, g.M, k_end, -1, tid);
                if (k0 + BK <= k_end && b_inside)
                    LB::template store<false, ONES, true>(bs, fb[d], none, n0, k0, n_lim, k_end, ones_x, tid);
                else
                    LB::template store<false, ONES>(bs, fb[d], none, n0, k0, n_lim, k_end, ones_x, tid);
                if (k0 + D * BK < k_end) {   // block-uniform
                    LA::template load<MASK>(fa[d], fm[d], g.A, g.amask, g.sa_m, g.sa_k, m0, k0 + D * BK, g.M, k_end, tid);
                    LB::template load<false>(fb[d], none, g.B, nullptr, g.sb_n, g.sb_k, n0, k0 + D * BK, n_lim, k_end, tid);
                }
                __syncthreads();
                const float *ar = a_rd + (t & 1) * STAGE_FLOATS, *br = b_rd + (t & 1) * STAGE_FLOATS;
                float av[KSTEPS / 2], bv[KSTEPS / 2];
#pragma unroll
                for (int q = 0; q < KSTEPS / 2; ++q) av[q] = ar[2 * q * LA::STRIDE], bv[q] = br[2 * q * LB::STRIDE];
#pragma unroll
                for (int q = 0; q < KSTEPS / 2; ++q)   // (rows past the end of K hold zeros)
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[q], acc[0], 0, 0, 0);
            }
        }
        __syncthreads();   // the partial tiles below reuse the operand stages
        float *red = smem + EPI_FLOATS;
        if (grp > 0) {
            float *r = red + (grp - 1) * 16 * 64 + lane;
#pragma unroll
            for (int i = 0; i < 16; ++i) r[i * 64] = acc[0][i];
        }
        __syncthreads();
        if (grp > 0) return;
#pragma unroll
        for (int q = 0; q < KG - 1; ++q)
#pragma unroll
            for (int i = 0; i < 16; ++i) acc[0][i] += red[(q * 16 + i) * 64 + lane];
    }

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) +
    // 4 * (lane >> 5): a direct store would be 16 dword stores per tile, each touching two
    // 128-B row segments.  Instead every wave transposes one tile at a time through its own
    // 32 x 36 LDS patch and writes float4 per lane: 4 store instructions of 8 rows x 128 B.
    const int col_lim = min(g.n_ext, n0 + g.bn);
    if constexpr (CLS) {
        // The result tile is not written: relu(acc + bias) is multiplied into the 4-class layer on the spot.
        // After a tile's transposition through the patch lane (prow, pc4) holds columns pc4 .. pc4 + 3 of rows
        // prow + 8 i; it adds their products with the four class rows to 16 running sums (row group x class),
        // which are then summed over the 8 lanes of a row (fixed order) and written as one float4 per row.
        static_assert(WN == 1, "class epilogue: 4 x 1 wave grid");
        float *patch = smem + wave * 32 * EPI_STRIDE;
        const int prow = lane >> 3, pc4 = (lane & 7) * 4;
        float part[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int c = 0; c < 4; ++c) part[i][c] = 0.f;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cbase = n0 + j * 32;
            if (cbase >= col_lim) continue;   // wave-uniform
            const int col = cbase + pc4;
            const bool in = col < g.N;        // (N % 4 == 0: a lane's four columns are inside together)
            float4 b4 = make_float4(0.f, 0.f, 0.f, 0.f), w4[4];
            if (in && g.bias != nullptr) b4 = *reinterpret_cast<const float4 *>(g.bias + col);
#pragma unroll
            for (int c = 0; c < 4; ++c)
                w4[c] = in ? *reinterpret_cast<const float4 *>(g.cls_w + (long long)c * g.N + col) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * EPI_STRIDE + l31] = acc[j][r];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float4 v = *reinterpret_cast<const float4 *>(patch + (prow + 8 * i) * EPI_STRIDE + pc4);
                v.x = fmaxf(v.x + b4.x, 0.f), v.y = fmaxf(v.y + b4.y, 0.f), v.z = fmaxf(v.z + b4.z, 0.f), v.w = fmaxf(v.w + b4.w, 0.f);
#pragma unroll
                for (int c = 0; c < 4; ++c)
                    part[i][c] += (v.x * w4[c].x + v.y * w4[c].y) + (v.z * w4[c].z + v.w * w4[c].w);
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch is rewritten by the next tile
        }
#pragma unroll
        for (int i = 0; i < 4; ++i) {
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                float x = part[i][c];
                x += __shfl_xor(x, 1);
                x += __shfl_xor(x, 2);
                x += __shfl_xor(x, 4);
                part[i][c] = x;
            }
            const int row = m0 + wave_m * 32 + prow + 8 * i;
            if ((lane & 7) == 0 && row < g.M)
                *reinterpret_cast<float4 *>(g.cls_out + ((long long)row * g.nt_n + tile_n) * 4) =
                    make_float4(part[i][0], part[i][1], part[i][2], part[i][3]);
        }
        return;
    }
    const bool split = g.split_k > 1;
    float *out = split ? g.ws + (long long)z * g.M * g.ws_ld : g.C;
    const long long ld = split ? g.ws_ld : g.ldc;
    if (g.c_vec) {
        // A wait for a load also waits for every store issued before it (one in-order counter), and a
        // store takes microseconds to be acknowledged while the chip is writing C.  So every
        // value the epilogue reads from memory is requested BEFORE the stores it would queue
        // behind: all bias vectors ahead of the first store, and a tile's old C / gate rows ahead
        // of that tile's stores.  The empty asm statements consume the loaded registers, which
        // pins the wait to that point instead of one wait per store further down.
        float *patch = smem + wave * 32 * EPI_STRIDE;
        const int prow = lane >> 3, pc4 = (lane & 7) * 4;
        const bool use_bias = !split && g.bias != nullptr;
        const bool use_old = !split && g.accumulate, use_gate = !split && g.cmask != nullptr;
        float4 bias4[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wave_n * TN + j) * 32 + pc4;
            bias4[j] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (use_bias && col < g.N) bias4[j] = *reinterpret_cast<const float4 *>(g.bias + col);
        }
        if (use_bias) {
#pragma unroll
            for (int j = 0; j < TN; ++j)
                asm volatile("" : "+v"(bias4[j].x), "+v"(bias4[j].y), "+v"(bias4[j].z), "+v"(bias4[j].w));
        }
        // Instructions of a wave that shares its SIMD with MFMA loops issue an order of magnitude
        // slower than alone (stamps: 0.16 us for a tile's LDS transposition, 1.3-2 us for the ~160
        // vector instructions around its four stores), so the common case - a tile that lies fully
        // inside C - runs on row pointers computed once, immediate column offsets and no bounds tests.
        const bool rows_full = g.c_vec == 2 && m0 + wave_m * 32 + 32 <= g.M;
        unsigned ro[4];   // byte offsets of this lane's four rows (c_vec == 2: they fit 32 bits)
#pragma unroll
        for (int i = 0; i < 4; ++i)
            ro[i] = rows_full ? (unsigned)(((long long)(m0 + wave_m * 32 + prow + 8 * i) * ld + n0 + wave_n * TN * 32 + pc4) * 4)
                              : 0u;
        char *out_b = reinterpret_cast<char *>(out);
        // Sign bits (never with a K split): 16 lane masks per 32 x 32 tile, indexed [row group of 32]
        // [column group of 32][i][component] - lane (prow, pc4) of row group i owns row prow + 8 i,
        // columns pc4 .. pc4 + 3, in the full-tile path and in the edge path alike.
        const bool wave_has_rows = m0 + wave_m * 32 < g.M;   // (a wave below the last row owns no row group of the bit array)
        const bool put_bits = !split && g.relu_bits != nullptr && wave_has_rows;
        const bool get_bits = !split && g.gate_bits != nullptr && wave_has_rows;
        // (the index is made wave-uniform for the compiler, so that the masks are fetched by scalar
        // loads: no vector registers, and no queueing behind the vector stores of the tile before)
        const int bits_row = ((m0 + wave_m * 32) >> 5) * g.bits_ng;
        auto tile_bits = [&](int cbase) { return __builtin_amdgcn_readfirstlane((bits_row + (cbase >> 5)) * 16); };
        auto uniform_mask = [&](const unsigned long long *p) { return *p; };
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int cbase = n0 + (wave_n * TN + j) * 32;
            if (cbase >= col_lim) continue;   // wave-uniform
#pragma unroll
            for (int r = 0; r < 16; ++r)
                patch[((r & 3) + 8 * (r >> 2) + 4 * lhi) * EPI_STRIDE + l31] = acc[j][r];
            if (rows_full && !use_old && !use_gate && cbase + 32 <= g.N) {   // wave-uniform: the whole tile is inside C
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
                const float4 b4 = bias4[j];
                const float inf = __builtin_inff();
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    float4 v = *reinterpret_cast<const float4 *>(patch + (prow + 8 * i) * EPI_STRIDE + pc4);
                    if (!split) {
                        v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
                        if (g.act == PSVAE_ACT_RELU)   // one v_med3_f32 per element
                            v.x = __builtin_amdgcn_fmed3f(v.x, 0.f, inf), v.y = __builtin_amdgcn_fmed3f(v.y, 0.f, inf),
                            v.z = __builtin_amdgcn_fmed3f(v.z, 0.f, inf), v.w = __builtin_amdgcn_fmed3f(v.w, 0.f, inf);
                    }
                    if (get_bits) {
                        const unsigned long long *gp = g.gate_bits + tile_bits(cbase) + 4 * i;
                        v.x = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp)) ? v.x : 0.f;
                        v.y = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 1)) ? v.y : 0.f;
                        v.z = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 2)) ? v.z : 0.f;
                        v.w = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 3)) ? v.w : 0.f;
                    }
                    if (put_bits) {
                        const unsigned long long bx = __ballot(v.x > 0.f), by = __ballot(v.y > 0.f);
                        const unsigned long long bz = __ballot(v.z > 0.f), bw = __ballot(v.w > 0.f);
                        if (lane == 0) {
                            unsigned long long *bp = g.relu_bits + tile_bits(cbase) + 4 * i;
                            bp[0] = bx, bp[1] = by, bp[2] = bz, bp[3] = bw;
                        }
                    }
                    *reinterpret_cast<float4 *>(out_b + ro[i] + 128 * j) = v;
                }
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch is rewritten by the next tile
                continue;
            }
            const int col = cbase + pc4;
            float4 old4[4], gate4[4];
            if (use_old || use_gate) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int row = m0 + wave_m * 32 + prow + 8 * i;
                    const bool in = row < g.M && col < g.N;
                    const long long off = in ? (long long)row * ld + col : 0;
                    old4[i] = gate4[i] = make_float4(0.f, 0.f, 0.f, 0.f);
                    if (use_old) old4[i] = *reinterpret_cast<const float4 *>(out + off);
                    if (use_gate) gate4[i] = *reinterpret_cast<const float4 *>(g.cmask + off);
                }
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    asm volatile("" : "+v"(old4[i].x), "+v"(old4[i].y), "+v"(old4[i].z), "+v"(old4[i].w));
                    asm volatile("" : "+v"(gate4[i].x), "+v"(gate4[i].y), "+v"(gate4[i].z), "+v"(gate4[i].w));
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int row = m0 + wave_m * 32 + prow + 8 * i;
                float4 v = *reinterpret_cast<const float4 *>(patch + (prow + 8 * i) * EPI_STRIDE + pc4);
                if (put_bits || get_bits) {   // wave-uniform; every lane takes part in the lane masks
                    const bool in = row < g.M && col < g.N;
                    const float4 b4 = bias4[j];
                    v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
                    if (g.act == PSVAE_ACT_RELU)
                        v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                    if (use_old) {
                        const float4 o = old4[i];
                        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                    }
                    if (get_bits) {
                        const unsigned long long *gp = g.gate_bits + tile_bits(cbase) + 4 * i;
                        v.x = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp)) ? v.x : 0.f;
                        v.y = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 1)) ? v.y : 0.f;
                        v.z = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 2)) ? v.z : 0.f;
                        v.w = __builtin_amdgcn_inverse_ballot_w64(uniform_mask(gp + 3)) ? v.w : 0.f;
                    }
                    if (use_gate) {
                        const float4 mk = gate4[i];
                        v.x = mk.x > 0.f ? v.x : 0.f, v.y = mk.y > 0.f ? v.y : 0.f;
                        v.z = mk.z > 0.f ? v.z : 0.f, v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    if (put_bits) {
                        const unsigned long long bx = __ballot(in && v.x > 0.f), by = __ballot(in && v.y > 0.f);
                        const unsigned long long bz = __ballot(in && v.z > 0.f), bw = __ballot(in && v.w > 0.f);
                        if (lane == 0) {
                            unsigned long long *bp = g.relu_bits + tile_bits(cbase) + 4 * i;
                            bp[0] = bx, bp[1] = by, bp[2] = bz, bp[3] = bw;
                        }
                    }
                    if (in) *reinterpret_cast<float4 *>(out + (long long)row * ld + col) = v;
                    continue;
                }
                if (row >= g.M || col >= col_lim) continue;
                if (split) {
                    *reinterpret_cast<float4 *>(out + (long long)row * ld + col) = v;
                } else if (col == g.N) {          // the virtual ones column (N % 4 == 0 here)
                    float *rs = g.rowsum + row;
                    *rs = g.accumulate ? *rs + v.x : v.x;
                } else {
                    const float4 b4 = bias4[j];
                    v.x += b4.x, v.y += b4.y, v.z += b4.z, v.w += b4.w;
                    if (g.act == PSVAE_ACT_RELU)
                        v.x = fmaxf(v.x, 0.f), v.y = fmaxf(v.y, 0.f), v.z = fmaxf(v.z, 0.f), v.w = fmaxf(v.w, 0.f);
                    if (use_old) {
                        const float4 o = old4[i];
                        v.x += o.x, v.y += o.y, v.z += o.z, v.w += o.w;
                    }
                    if (use_gate) {
                        const float4 mk = gate4[i];
                        v.x = mk.x > 0.f ? v.x : 0.f, v.y = mk.y > 0.f ? v.y : 0.f;
                        v.z = mk.z > 0.f ? v.z : 0.f, v.w = mk.w > 0.f ? v.w : 0.f;
                    }
                    *reinterpret_cast<float4 *>(out + (long long)row * ld + col) = v;
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // patch is rewritten by the next tile
        }
    } else {
        const int row0 = m0 + wave_m * 32 + 4 * lhi;
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wave_n * TN + j) * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = row0 + (r & 3) + 8 * (r >> 2);
                if (col < col_lim && row < g.M) store_result(g, z, row, col, acc[j][r]);
            }
        }
    }
}

template <int TN, int BK>
void launch_panel(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    const bool mask = g.amask != nullptr, ones = g.rowsum != nullptr;
    const dim3 blk(THREADS);
#define PSVAE_PANEL(AKC, BKC, MASK, ONES) \
    hipLaunchKernelGGL((gemm_panel_kernel<TN, BK, AKC, BKC, MASK, ONES>), grid, blk, 0, s, g)
    if (a_kc && b_kc) {                 // forward X W^T
        if (mask) PSVAE_PANEL(true, true, true, false); else PSVAE_PANEL(true, true, false, false);
    } else if (a_kc) {                  // input gradient dY W (rowsum: the ones column of the n-contiguous B)
        if (mask && ones) PSVAE_PANEL(true, false, true, true);
        else if (mask) PSVAE_PANEL(true, false, true, false);
        else if (ones) PSVAE_PANEL(true, false, false, true);
        else PSVAE_PANEL(true, false, false, false);
    } else if (b_kc) {
        if (mask) PSVAE_PANEL(false, true, true, false); else PSVAE_PANEL(false, true, false, false);
    } else if (g.wn == 2) {             // the same on the 2 x 2 wave grid (TN == 5 only)
        if constexpr (TN == 5) {
#define PSVAE_PANEL22(MASK, ONES) \
    hipLaunchKernelGGL((gemm_panel_kernel<5, BK, false, false, MASK, ONES, 2>), grid, blk, 0, s, g)
            if (mask && ones) PSVAE_PANEL22(true, true);
            else if (mask) PSVAE_PANEL22(true, false);
            else if (ones) PSVAE_PANEL22(false, true);
            else PSVAE_PANEL22(false, false);
#undef PSVAE_PANEL22
        }
    } else {                            // weight (+ bias) gradient dY^T [X | 1]
        if (mask && ones) PSVAE_PANEL(false, false, true, true);
        else if (mask) PSVAE_PANEL(false, false, true, false);
        else if (ones) PSVAE_PANEL(false, false, false, true);
        else PSVAE_PANEL(false, false, false, false);
    }
#undef PSVAE_PANEL
}

// (instantiated in panel_cls_inst.hip only)
template <int TN>
void launch_panel_classes_tn(const GemmArgs &g, dim3 grid, hipStream_t s) {
    hipLaunchKernelGGL((gemm_panel_kernel<TN, PSVAE_PANEL_BK, true, true, false, false, 1, true>), grid, dim3(THREADS), 0, s, g);
}

}  // namespace psvae_gemm
