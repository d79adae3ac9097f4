// f32 matrix products on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One entry point serves the three products of a Linear layer (reference:
// src/modules/common_nn.py:8-23 and its autograd): operands are addressed through element
// strides, so X W^T, dY W and dY^T X differ only in the strides passed.  The MFMA is an exact
// f32 fused-multiply-add chain in k order, so results match a plain f32 dot product to
// rounding of the summation order.
//
// Two kernels:
//  * gemm_panel_kernel<TN, BK, A_KC, B_KC> — the fast path.  A workgroup (4 waves of 64 lanes)
//    owns a 128-row panel of C and ALL its columns (or half of them when N > 352): wave w
//    holds rows 32w..32w+31 x 32*TN columns in TN accumulator tiles.  The layer widths of
//    this model (200, 300, 328, 400, 656, 56, 302) fit 32*TN with < 7 % padding, the A panel
//    is read from HBM exactly once, and per k-step a wave issues TN MFMAs for TN+1 LDS reads.
//    Global loads are 16 B per lane and the next k-tile is in flight (registers) while the
//    current one is multiplied.  Needs 16-byte aligned operands (checked on the host).
//  * gemm_generic_kernel — any strides / alignment, 64x64 tiles, scalar loads.
//
// LDS tiles are k-major (T[k][x]): an MFMA operand fetch is one conflict-free ds_read_b32 per
// lane (lanes 0-31 read consecutive x at k, lanes 32-63 at k+1 — the 32x32x2 operand layout).
// Row strides are chosen per loader so the staging stores are conflict-free too.
// Workgroups that share an A row-panel sit on one XCD (blockIdx % 8).
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));


namespace {

constexpr int THREADS = 256;
// Tuning constants of the panel kernel (the -D overrides exist for A/B builds only).
#ifndef PSVAE_PANEL_BK
#define PSVAE_PANEL_BK 32      // k-tile depth of the panels up to 192 columns wide
#endif
#ifndef PSVAE_PANEL_OCC
#define PSVAE_PANEL_OCC 3      // workgroups per CU the register budget of those panels is set for
#endif
#ifndef PSVAE_PANEL_NBUF
#define PSVAE_PANEL_NBUF 1     // LDS stages of the operand tiles
#endif

struct GemmArgs {
    int M, N, K;
    int n_ext;            // N + 1 when rowsum is requested (a virtual all-ones column of B)
    const float *A;
    const float *amask;   // optional, addressed like A: A(m,k) counts as 0 where amask(m,k) <= 0
    long long sa_m, sa_k;
    const float *B;
    long long sb_k, sb_n;
    float *C;
    long long ldc;
    const float *bias;
    float *rowsum;        // optional [M]: sum_k A(m,k)
    const float *cmask;   // optional, addressed like C: C(m,n) = 0 where cmask(m,n) <= 0
    int act;
    int accumulate;
    int split_k;
    int k_chunk;          // multiple of 32
    float *ws;            // [split_k][M][ws_ld] when split_k > 1
    int ws_ld;            // n_ext rounded up to a multiple of 4
    int c_vec;            // C (and ws) rows may be written 16 bytes at a time; 2: and byte offsets fit 32 bits
    int nt_m, nt_n, bn;   // tiles and column-tile width
    int wn;               // wave grid of the panel kernel: 1 = 4 x 1, 2 = 2 x 2
    int tail_begin;       // rows tail_begin .. K-1 of the contraction are added by the reduce kernel (K: none)
};

__device__ __forceinline__ void epilogue_store(const GemmArgs &g, int row, int col, float v) {
    if (col == g.N) {   // the virtual ones column: row sums of A
        float *r = g.rowsum + row;
        *r = g.accumulate ? *r + v : v;
        return;
    }
    if (g.bias) v += g.bias[col];
    if (g.act == PSVAE_ACT_RELU) v = fmaxf(v, 0.f);
    float *c = g.C + (long long)row * g.ldc + col;
    if (g.accumulate) v += *c;
    if (g.cmask && !(g.cmask[(long long)row * g.ldc + col] > 0.f)) v = 0.f;
    *c = v;
}

__device__ __forceinline__ void store_result(const GemmArgs &g, int z, int row, int col, float v) {
    if (g.split_k > 1)
        g.ws[((long long)z * g.M + row) * g.ws_ld + col] = v;
    else
        epilogue_store(g, row, col, v);
}

// XCD-aware work assignment on a 1-D grid.  A unit = (row panel, K split); its nt_n column
// tiles get consecutive local ids on one XCD (blockIdx % 8 picks the XCD), so the A panel
// chunk they share is fetched into one L2, and the units are dealt round-robin over the
// XCDs so that a product with few row panels but many splits still fills the whole chip.
// Returns false for the padding blocks of the last group of 8 units.
__device__ __forceinline__ bool tile_of_block(const GemmArgs &g, int &tile_m, int &tile_n, int &z) {
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    tile_n = local % g.nt_n;
    const int unit = (local / g.nt_n) * 8 + xcd;
    tile_m = unit % g.nt_m;
    z = unit / g.nt_m;
    return z < g.split_k;
}

// ------------------------------------------------------------------------------------------
// Fast path: 128 x (32*TN) panel per workgroup, vector loads.
// ------------------------------------------------------------------------------------------
template <int BX, int BK, bool KCONTIG>
struct TileLoader {
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
// (64 rows x 64*TN columns), for products whose M fits 64-row panels with less padding.
template <int TN, int BK, bool A_KC, bool B_KC, bool MASK, bool ONES, int WN = 1>
__global__ __launch_bounds__(THREADS, (MASK || WN > 1 || TN > 5) ? 2 : PSVAE_PANEL_OCC) void gemm_panel_kernel(GemmArgs g) {
    constexpr int WM = 4 / WN, BM = 32 * WM, BN = 32 * TN * WN;
    using LA = TileLoader<BM, BK, A_KC>;
    using LB = TileLoader<BN, BK, B_KC>;
    constexpr int A_FLOATS = BK * LA::STRIDE, B_FLOATS = BK * LB::STRIDE;
    constexpr int STAGE_FLOATS = A_FLOATS + B_FLOATS;
    // two LDS stages where two workgroups of them still fit a CU's 160 KiB
    constexpr int NBUF = (PSVAE_PANEL_NBUF == 2 && 2 * STAGE_FLOATS * 4 <= 76 * 1024) ? 2 : 1;
    constexpr int EPI_STRIDE = 36, EPI_FLOATS = 4 * 32 * EPI_STRIDE;   // one 32x32 tile per wave
    constexpr int LDS_FLOATS = NBUF * STAGE_FLOATS > EPI_FLOATS ? NBUF * STAGE_FLOATS : EPI_FLOATS;
    __shared__ __align__(16) float smem[LDS_FLOATS];
    float *As = smem, *Bs = smem + A_FLOATS;

    int tile_m, tile_n, z;
    if (!tile_of_block(g, tile_m, tile_n, z)) return;
    const int m0 = tile_m * BM, n0 = tile_n * g.bn;
    const int k_begin = z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    auto multiply = [&](int buf, int k0) {
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

    // Epilogue.  C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) +
    // 4 * (lane >> 5): a direct store would be 16 dword stores per tile, each touching two
    // 128-B row segments.  Instead every wave transposes one tile at a time through its own
    // 32 x 36 LDS patch and writes float4 per lane: 4 store instructions of 8 rows x 128 B.
    const int col_lim = min(g.n_ext, n0 + g.bn);
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

// ------------------------------------------------------------------------------------------
// Generic path: 64 x 64 tiles, scalar loads, any strides.
// ------------------------------------------------------------------------------------------
constexpr int GBK = 32, GB = 64;

template <bool KCONTIG>
__device__ __forceinline__ void generic_load(float (&r)[GB / 8], const float *__restrict__ p,
                                             const float *__restrict__ pm, long long sx, long long sk, int x0,
                                             int k0, int X, int k_end, int ones_x, int tid) {
#pragma unroll
    for (int i = 0; i < GB / 8; ++i) {
        int x, k;
        if (KCONTIG) {
            k = k0 + (tid & 31), x = x0 + (tid >> 5) + 8 * i;
        } else {
            x = x0 + (tid % GB), k = k0 + (tid / GB) + (THREADS / GB) * i;
        }
        float v = 0.f;
        if (k < k_end) {
            if (x < X) {
                const long long off = (long long)x * sx + (long long)k * sk;
                v = p[off];
                if (pm != nullptr && !(pm[off] > 0.f)) v = 0.f;
            } else if (x == ones_x) {
                v = 1.f;
            }
        }
        r[i] = v;
    }
}

template <bool KCONTIG>
__device__ __forceinline__ void generic_store(float (*T)[GB + 1], const float (&r)[GB / 8], int tid) {
#pragma unroll
    for (int i = 0; i < GB / 8; ++i) {
        if (KCONTIG)
            T[tid & 31][(tid >> 5) + 8 * i] = r[i];
        else
            T[(tid / GB) + (THREADS / GB) * i][tid % GB] = r[i];
    }
}

template <bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS) void gemm_generic_kernel(GemmArgs g) {
    __shared__ float As[GBK][GB + 1];
    __shared__ float Bs[GBK][GB + 1];
    int tile_m, tile_n, z;
    if (!tile_of_block(g, tile_m, tile_n, z)) return;
    const int m0 = tile_m * GB, n0 = tile_n * GB;
    const int k_begin = z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lhi = lane >> 5;
    const int ones_x = g.rowsum ? g.N : -1;

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    float ra[GB / 8], rb[GB / 8];
    if (k_begin < k_end) {
        generic_load<A_KC>(ra, g.A, g.amask, g.sa_m, g.sa_k, m0, k_begin, g.M, k_end, -1, tid);
        generic_load<B_KC>(rb, g.B, nullptr, g.sb_n, g.sb_k, n0, k_begin, g.N, k_end, ones_x, tid);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += GBK) {
        generic_store<A_KC>(As, ra, tid);
        generic_store<B_KC>(Bs, rb, tid);
        __syncthreads();
        if (k0 + GBK < k_end) {
            generic_load<A_KC>(ra, g.A, g.amask, g.sa_m, g.sa_k, m0, k0 + GBK, g.M, k_end, -1, tid);
            generic_load<B_KC>(rb, g.B, nullptr, g.sb_n, g.sb_k, n0, k0 + GBK, g.N, k_end, ones_x, tid);
        }
#pragma unroll
        for (int kk = 0; kk < GBK; kk += 2)
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(As[kk + lhi][wm * 32 + l31], Bs[kk + lhi][wn * 32 + l31], acc,
                                                       0, 0, 0);
        __syncthreads();
    }
    const int col = n0 + wn * 32 + l31;
    if (col >= g.n_ext) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
        const int row = m0 + wm * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
        if (row < g.M) store_result(g, z, row, col, acc[r]);
    }
}

// Sum the split-K slabs and apply the epilogue.  A block of 256 threads takes 64 groups of four
// consecutive columns; its four waves each sum a quarter of the slabs (16-byte loads, eight in
// flight), and wave 0 adds the four partial sums in wave order: the result does not depend on the
// grid.  Needs a 16-byte aligned workspace (ws_ld is a multiple of 4 by construction).
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(GemmArgs g) {
    __shared__ float4 part[3][64];
    const int ld4 = g.ws_ld >> 2;
    const long long quads = (long long)g.M * ld4;
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int per = (g.split_k + 3) >> 2, z0 = q * per, z1 = min(g.split_k, z0 + per);
    const float4 *ws4 = reinterpret_cast<const float4 *>(g.ws);
    for (long long base = (long long)blockIdx.x * 64; base < quads; base += (long long)gridDim.x * 64) {
        const long long idx = base + l;
        float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
        if (idx < quads) {
#pragma unroll 8
            for (int z = z0; z < z1; ++z) {
                const float4 t = ws4[(long long)z * quads + idx];
                v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
            }
        }
        if (q) part[q - 1][l] = v;
        __syncthreads();
        if (q == 0 && idx < quads) {
#pragma unroll
            for (int p = 0; p < 3; ++p) {
                const float4 t = part[p][l];
                v.x += t.x, v.y += t.y, v.z += t.z, v.w += t.w;
            }
            const int row = (int)(idx / ld4), col = (int)(idx % ld4) * 4;
            const float e[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int c = 0; c < 4; ++c) {
                if (col + c >= g.n_ext) continue;
                float r = e[c];
                for (int k = g.tail_begin; k < g.K; ++k)   // ragged end of a tall-K product (fewer than 16 rows)
                    r += g.A[row * g.sa_m + k * g.sa_k] * (col + c < g.N ? g.B[k * g.sb_k + (col + c) * g.sb_n] : 1.f);
                epilogue_store(g, row, col + c, r);
            }
        }
        __syncthreads();
    }
}

// The same for a workspace that is not 16-byte aligned: one element per thread, slabs in order.
__global__ void gemm_splitk_reduce_scalar_kernel(GemmArgs g) {
    const long long total = (long long)g.M * g.ws_ld;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(idx / g.ws_ld), col = (int)(idx % g.ws_ld);
        if (col >= g.n_ext) continue;
        float v = 0.f;
        for (int z = 0; z < g.split_k; ++z) v += g.ws[(long long)z * total + idx];
        epilogue_store(g, row, col, v);
    }
}

void launch_splitk_reduce(const GemmArgs &g, hipStream_t s) {
    const long long total = (long long)g.M * g.ws_ld;
    if (psvae_aligned16(g.ws)) {
        const long long nb = (total / 4 + 63) / 64;
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3((unsigned)(nb < 8192 ? nb : 8192)), dim3(256), 0, s, g);
    } else {
        const long long nb = (total + 255) / 256;
        hipLaunchKernelGGL(gemm_splitk_reduce_scalar_kernel, dim3((unsigned)(nb < 4096 ? nb : 4096)), dim3(256), 0, s, g);
    }
}

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
constexpr int TK_THREADS = 512, TK_BK = 16, TK_LDA = 304, TK_LDB = 176, TK_T = 5, TK_MAX = 304, TK_STAGES = 3;
constexpr int TK_HALF = 160;   // output columns per workgroup
typedef float f32x4 __attribute__((ext_vector_type(4)));

struct TallKArgs {
    int M, N, K, n_ext;   // K: a multiple of TK_BK
    const float *A;       // A(m,k) = A[k * lda + m]
    long long lda;
    const float *B;       // B(k,n) = B[k * ldb + n]
    long long ldb;
    float *ws;            // [chunks][M][ws_ld]
    int ws_ld, k_chunk, halves, chunks;
};

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

    // Ring of three stages with ONE barrier per k-tile, placed in the middle of the tile: by then the
    // tile after this one has landed (it was requested a whole tile earlier), so the last k-step of a
    // tile can already fetch the first fragments of the next one, and the stage two tiles ahead -
    // last read before every wave reached this barrier - is free to be refilled.
    const int n_tiles = max(0, k_end - k_begin) / TK_BK;
    if (n_tiles > 0) issue(0, k_begin);
    if (n_tiles > 1) issue(1, k_begin + TK_BK);
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

constexpr int TALLK_MIN_K = 16384;   // below this the panel kernel's split-K does as well

// K chunks of the tall-K kernel: one workgroup per CU when K allows it (two workgroups per chunk when
// the output has more than 160 columns), at least 8 k-tiles each
int tallk_split(int K, int n_ext) {
    const int cap = n_ext > TK_HALF ? 128 : 256;
    int s = K / (8 * TK_BK);
    return s < 1 ? 1 : (s > cap ? cap : s);
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

// Widest column tile: 224 columns = 7 accumulator tiles per wave (wider tiles were measured
// slower: they leave two workgroups per CU).
constexpr int PANEL_MAX_BN = 224;
int panel_max_bn() { return PANEL_MAX_BN; }

// Column tiling of the panel kernel: nc tiles of width bn (a multiple of 32), and the
// instantiated TN that covers bn.
void panel_shape(int n_ext, int &nc, int &bn, int &tn) {
    nc = psvae_cdiv(n_ext, panel_max_bn());
    bn = psvae_cdiv(psvae_cdiv(n_ext, nc), 32) * 32;
    const int need = bn / 32;
    static const int kTn[] = {1, 2, 3, 4, 5, 6, 7, 10, 11};
    tn = 11;
    for (int t : kTn)
        if (t >= need) {
            tn = t;
            break;
        }
}

// split_k == 0: pick a split that fills the chip when the output has few tiles: the
// weight-gradient products (M, N ~ 300, K = number of rows in the batch) and the per-step
// products of the GRU (4096 rows, K = 600).
// The 2 x 2 wave grid (64-row panels, one column tile of 320) pays when it cuts the padded rows:
// M = 300 is 5 panels of 64 instead of 3 of 128.
bool prefer_2x2(int M, int n_ext) {
    return n_ext > 224 && n_ext <= 320 && (M - 1) % 128 < 64;
}

int resolve_split(int M, int N, int K, int split_k) {
    if (split_k >= 1) return split_k;
    int nc, bn, tn;
    panel_shape(N + 1, nc, bn, tn);
    // (assumes the strides of a weight-gradient product when the 2 x 2 grid could apply; a
    // wrong guess only changes how many K chunks are cut)
    const long long tiles = prefer_2x2(M, N + 1) ? psvae_cdiv(M, 64) : (long long)psvae_cdiv(M, 128) * nc;
    if (tiles >= 192) return 1;
    long long s = 512 / tiles;
    const long long by_k = K / 128;   // at least four k-tiles of work per split
    if (s > by_k) s = by_k;
    return s < 1 ? 1 : (int)s;
}

bool vec_ok(const float *p, long long s_major) {
    return p == nullptr || (psvae_aligned16(p) && s_major % 4 == 0);
}

}  // namespace

extern "C" size_t psvae_gemm_f32_workspace_bytes(int M, int N, int K, int split_k) {
    if (M <= 0 || N <= 0) return 0;
    size_t tall = 0;   // the tall-K kernel may take the product (decided from the strides at the call)
    if (split_k == 0 && M <= TK_MAX && N + 1 <= TK_MAX && K >= TALLK_MIN_K)
        tall = (size_t)tallk_split(K, N + 1) * (size_t)M * (size_t)(N + 4) * sizeof(float);
    split_k = resolve_split(M, N, K, split_k);
    const size_t panel = split_k <= 1 ? 0 : (size_t)split_k * (size_t)M * (size_t)(N + 4) * sizeof(float);
    return tall > panel ? tall : panel;
}

extern "C" int psvae_gemm_f32(int M, int N, int K, const float *A, const float *amask, int64_t sa_m,
                              int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_n, float *C, int64_t ldc,
                              const float *bias, float *rowsum, const float *cmask, int act, int accumulate,
                              int split_k, void *workspace, size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(M >= 0 && N >= 0 && K >= 0, PSVAE_EINVAL, "psvae_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return PSVAE_OK;
    PSVAE_REQUIRE(C != nullptr && (K == 0 || (A != nullptr && B != nullptr)), PSVAE_EINVAL,
                  "psvae_gemm_f32: null operand");
    PSVAE_REQUIRE(ldc >= N, PSVAE_EINVAL, "psvae_gemm_f32: ldc %lld < N %d", (long long)ldc, N);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "psvae_gemm_f32: bad act %d", act);
    hipStream_t s = (hipStream_t)stream;
    const int n_ext0 = N + (rowsum ? 1 : 0);
    if (split_k == 0 && sa_m == 1 && sb_n == 1 && amask == nullptr && M <= TK_MAX && n_ext0 <= TK_MAX &&
        K >= TALLK_MIN_K && M % 4 == 0 && N % 4 == 0 && sa_k % 4 == 0 && sb_k % 4 == 0 && sa_k >= M && sb_k >= N &&
        psvae_aligned16(A) && psvae_aligned16(B) && psvae_aligned16(workspace)) {
        const int split = tallk_split(K, n_ext0);
        GemmArgs g;
        g.M = M, g.N = N, g.K = K, g.n_ext = n_ext0, g.ws_ld = (n_ext0 + 3) / 4 * 4;
        PSVAE_REQUIRE(workspace != nullptr &&
                          workspace_bytes >= (size_t)split * (size_t)M * (size_t)g.ws_ld * sizeof(float),
                      PSVAE_EWORKSPACE, "psvae_gemm_f32: workspace too small for the tall-K product");
        g.A = A, g.amask = nullptr, g.sa_m = sa_m, g.sa_k = sa_k, g.B = B, g.sb_k = sb_k, g.sb_n = sb_n;
        g.C = C, g.ldc = ldc, g.bias = bias, g.rowsum = rowsum, g.cmask = cmask, g.act = act, g.accumulate = accumulate;
        g.split_k = split, g.ws = (float *)workspace, g.c_vec = 0, g.nt_m = g.nt_n = 1, g.bn = 0, g.wn = 1;
        g.tail_begin = K / TK_BK * TK_BK;   // the kernel takes whole k-tiles, the reduce kernel the rows left over
        g.k_chunk = psvae_cdiv(psvae_cdiv(g.tail_begin, split), TK_BK) * TK_BK;
        g.split_k = psvae_cdiv(g.tail_begin, g.k_chunk);
        const int halves = n_ext0 > TK_HALF ? 2 : 1;
        TallKArgs t{M, N, g.tail_begin, n_ext0, A, (long long)sa_k, B, (long long)sb_k, g.ws, g.ws_ld, g.k_chunk, halves, g.split_k};
        const size_t lds = (size_t)TK_STAGES * TK_BK * (TK_LDA + TK_LDB) * sizeof(float);
        static bool attr_set = false;   // 92 160 bytes of dynamic LDS: above the default limit
        if (!attr_set) {
            (void)hipFuncSetAttribute((const void *)gemm_tallk_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
            attr_set = true;
        }
        // with two halves the chunk count is padded to a multiple of 8 (blocks b, b + 8 pair up)
        const int chunks_launched = halves == 2 ? psvae_cdiv(g.split_k, 8) * 8 : g.split_k;
        hipLaunchKernelGGL(gemm_tallk_kernel, dim3(chunks_launched * halves), dim3(TK_THREADS), lds, s, t);
        PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (tall-K)");
        launch_splitk_reduce(g, s);
        PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (tall-K reduce)");
        return PSVAE_OK;
    }
    split_k = resolve_split(M, N, K, split_k);
    int k_chunk = psvae_cdiv(psvae_cdiv(K, split_k), 32) * 32;
    if (k_chunk == 0) k_chunk = 32;
    split_k = K > 0 ? psvae_cdiv(K, k_chunk) : 1;
    PSVAE_REQUIRE(split_k <= 65535, PSVAE_ETOOLARGE, "psvae_gemm_f32: split_k %d too large", split_k);
    if (split_k > 1) {
        PSVAE_REQUIRE(workspace != nullptr && workspace_bytes >= psvae_gemm_f32_workspace_bytes(M, N, K, split_k),
                      PSVAE_EWORKSPACE, "psvae_gemm_f32: workspace too small for split_k %d", split_k);
    }
    GemmArgs g;
    g.M = M, g.N = N, g.K = K, g.n_ext = N + (rowsum ? 1 : 0);
    g.ws_ld = (g.n_ext + 3) / 4 * 4;
    g.c_vec = (split_k > 1 ? psvae_aligned16(workspace) : (psvae_aligned16(C) && ldc % 4 == 0)) && N % 4 == 0 &&
              (bias == nullptr || psvae_aligned16(bias)) && (cmask == nullptr || psvae_aligned16(cmask));
    if (g.c_vec && (long long)M * (split_k > 1 ? g.ws_ld : ldc) * 4 < (1ll << 32)) g.c_vec = 2;   // 32-bit byte offsets
    g.A = A, g.amask = amask, g.sa_m = sa_m, g.sa_k = sa_k;
    g.B = B, g.sb_k = sb_k, g.sb_n = sb_n;
    g.C = C, g.ldc = ldc, g.bias = bias, g.rowsum = rowsum, g.cmask = cmask, g.act = act;
    g.accumulate = accumulate, g.split_k = split_k, g.k_chunk = k_chunk, g.ws = (float *)workspace, g.tail_begin = K;
    const bool a_kc = (sa_k == 1), b_kc = (sb_k == 1);

    // The panel kernel needs unit stride along one axis of each operand, 16-byte aligned
    // bases / major strides, and the vectorised extent to be a multiple of 4.
    const bool a_vec = (a_kc ? (K % 4 == 0 && vec_ok(A, sa_m)) : (sa_m == 1 && M % 4 == 0 && vec_ok(A, sa_k))) &&
                       (amask == nullptr || psvae_aligned16(amask));
    const bool b_vec = b_kc ? (K % 4 == 0 && vec_ok(B, sb_n)) : (sb_n == 1 && N % 4 == 0 && vec_ok(B, sb_k));
    bool panel = a_vec && b_vec && (!rowsum || !b_kc) && M >= 32;
    int nc = 1, bn = 32, tn = 1;
    if (panel) panel_shape(g.n_ext, nc, bn, tn);
#ifdef PSVAE_FAST_BUILD   // A/B builds instantiate the 160-column panels only: everything else goes generic
    panel = panel && (tn == 5 || (!a_kc && !b_kc && prefer_2x2(M, g.n_ext)));
#endif

    if (panel) {
        g.nt_m = psvae_cdiv(M, 128), g.nt_n = nc, g.bn = bn, g.wn = 1;
        if (!a_kc && !b_kc && prefer_2x2(M, g.n_ext)) {
            g.wn = 2, g.nt_m = psvae_cdiv(M, 64), g.nt_n = 1, g.bn = 320;
            tn = 5;
        }
        const long long blocks = (long long)psvae_cdiv((long long)g.nt_m * split_k, 8) * 8 * g.nt_n;
        PSVAE_REQUIRE(blocks < (1ll << 31), PSVAE_ETOOLARGE, "psvae_gemm_f32: grid too large");
        dim3 grid((unsigned)blocks);
        switch (tn) {
#ifndef PSVAE_FAST_BUILD
            case 1: launch_panel<1, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 2: launch_panel<2, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 3: launch_panel<3, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 4: launch_panel<4, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 6: launch_panel<6, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 7: launch_panel<7, 16>(g, a_kc, b_kc, grid, s); break;
            case 10: launch_panel<10, 16>(g, a_kc, b_kc, grid, s); break;
            case 11: launch_panel<11, 16>(g, a_kc, b_kc, grid, s); break;
#endif
            default: launch_panel<5, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
        }
    } else {
        g.nt_m = psvae_cdiv(M, GB), g.nt_n = psvae_cdiv(g.n_ext, GB), g.bn = GB;
        const long long blocks = (long long)psvae_cdiv((long long)g.nt_m * split_k, 8) * 8 * g.nt_n;
        PSVAE_REQUIRE(blocks < (1ll << 31), PSVAE_ETOOLARGE, "psvae_gemm_f32: grid too large");
        dim3 grid((unsigned)blocks);
        if (a_kc && b_kc)
            hipLaunchKernelGGL((gemm_generic_kernel<true, true>), grid, dim3(THREADS), 0, s, g);
        else if (a_kc)
            hipLaunchKernelGGL((gemm_generic_kernel<true, false>), grid, dim3(THREADS), 0, s, g);
        else if (b_kc)
            hipLaunchKernelGGL((gemm_generic_kernel<false, true>), grid, dim3(THREADS), 0, s, g);
        else
            hipLaunchKernelGGL((gemm_generic_kernel<false, false>), grid, dim3(THREADS), 0, s, g);
    }
    PSVAE_CHECK_LAUNCH("psvae_gemm_f32");
    if (split_k > 1) {
        launch_splitk_reduce(g, s);
        PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (split-K reduce)");
    }
    return PSVAE_OK;
}
