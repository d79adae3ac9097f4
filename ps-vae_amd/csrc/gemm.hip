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
//    Products too small to fill the chip with panels take 64 x 64 tiles of the same kernel, and the
//    products of a small batch (at most 512 tiles of 32 x 32) its latency form: one 32 x 32 tile per
//    workgroup, the four waves splitting K, all of a short K requested at once (gemm_panel.h).
//  * gemm_generic_kernel — any strides / alignment, 64x64 tiles, scalar loads.
//
// LDS tiles are k-major (T[k][x]): an MFMA operand fetch is one conflict-free ds_read_b32 per
// lane (lanes 0-31 read consecutive x at k, lanes 32-63 at k+1 — the 32x32x2 operand layout).
// Row strides are chosen per loader so the staging stores are conflict-free too.
// Workgroups that share an A row-panel sit on one XCD (blockIdx % 8).
#include "gemm_common.h"

namespace psvae_gemm {

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

// Products below this many panel workgroups take the 64 x 64 tiles; their K split leaves at least
// four 64-deep k-tiles per workgroup and stops at one workgroup per CU (never more splits than
// resolve_split grants: the workspace is sized from that).
#ifndef PSVAE_TILE64_BELOW
#define PSVAE_TILE64_BELOW 256   // (the -D override exists for A/B builds only)
#endif
constexpr int TILE64_BELOW_BLOCKS = PSVAE_TILE64_BELOW;
#ifndef PSVAE_2X2_MIN_BLOCKS
#define PSVAE_2X2_MIN_BLOCKS 256
#endif
constexpr int TWO_BY_TWO_MIN_BLOCKS = PSVAE_2X2_MIN_BLOCKS;
int tile64_split(int M, int n_ext, int K) {
    const long long tiles = (long long)psvae_cdiv(M, 64) * psvae_cdiv(n_ext, 64);
    long long s = 256 / tiles;
    const long long by_k = K / 256;
    if (s > by_k) s = by_k;
    return s < 1 ? 1 : (int)s;
}

// Products of at most this many 32 x 32 tiles (two small workgroups per CU) take the latency form.
#ifndef PSVAE_TILE32_MAX_TILES
#define PSVAE_TILE32_MAX_TILES 512   // (the -D override exists for A/B builds only; 0: never)
#endif
constexpr int TILE32_MAX_TILES = PSVAE_TILE32_MAX_TILES;

bool vec_ok(const float *p, long long s_major) {
    return p == nullptr || (psvae_aligned16(p) && s_major % 4 == 0);
}

}  // namespace psvae_gemm

using namespace psvae_gemm;

extern "C" size_t psvae_gemm_f32_workspace_bytes(int M, int N, int K, int split_k) {
    if (M <= 0 || N <= 0) return 0;
    size_t tall = 0;   // the tall-K kernel may take the product (decided from the strides at the call)
    if (split_k == 0 && M <= TK_MAX && N <= TK_MAX && K >= TALLK_MIN_K) {
        // (with or without the row-sum column: the chunk cap differs at N == 160)
        const int a = tallk_split(K, N), b = tallk_split(K, N + 1);
        tall = (size_t)(a > b ? a : b) * (size_t)M * (size_t)(N + 4) * sizeof(float);
    }
    split_k = resolve_split(M, N, K, split_k);
    const size_t panel = split_k <= 1 ? 0 : (size_t)split_k * (size_t)M * (size_t)(N + 4) * sizeof(float);
    return tall > panel ? tall : panel;
}

static int gemm_impl(int M, int N, int K, const float *A, const float *amask, int64_t sa_m,
                     int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_n, float *C, int64_t ldc,
                     const float *bias, float *rowsum, const float *cmask, int act, int accumulate,
                     int split_k, void *workspace, size_t workspace_bytes, void *stream,
                     unsigned long long *relu_bits, const unsigned long long *gate_bits) {
    PSVAE_REQUIRE(M >= 0 && N >= 0 && K >= 0, PSVAE_EINVAL, "psvae_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return PSVAE_OK;
    PSVAE_REQUIRE(C != nullptr && (K == 0 || (A != nullptr && B != nullptr)), PSVAE_EINVAL,
                  "psvae_gemm_f32: null operand");
    PSVAE_REQUIRE(ldc >= N, PSVAE_EINVAL, "psvae_gemm_f32: ldc %lld < N %d", (long long)ldc, N);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "psvae_gemm_f32: bad act %d", act);
    hipStream_t s = (hipStream_t)stream;
    const int n_ext0 = N + (rowsum ? 1 : 0);
    if (split_k == 0 && !relu_bits && !gate_bits && sa_m == 1 && sb_n == 1 && amask == nullptr && M <= TK_MAX && n_ext0 <= TK_MAX &&
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
        g.relu_bits = nullptr, g.gate_bits = nullptr, g.bits_ng = 0;
        g.tail_begin = K / TK_BK * TK_BK;   // the kernel takes whole k-tiles, the reduce kernel the rows left over
        g.k_chunk = psvae_cdiv(psvae_cdiv(g.tail_begin, split), TK_BK) * TK_BK;
        g.split_k = psvae_cdiv(g.tail_begin, g.k_chunk);
        const int halves = n_ext0 > TK_HALF ? 2 : 1;
        TallKArgs t{M, N, g.tail_begin, n_ext0, A, (long long)sa_k, B, (long long)sb_k, g.ws, g.ws_ld, g.k_chunk, halves, g.split_k};
        if (int rc = launch_tallk(t, s)) return rc;
        launch_splitk_reduce(g, s);
        PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (tall-K reduce)");
        return PSVAE_OK;
    }
    const int n_ext = N + (rowsum ? 1 : 0);
    const bool a_kc = (sa_k == 1), b_kc = (sb_k == 1);
    // The panel kernel needs unit stride along one axis of each operand, 16-byte aligned
    // bases / major strides, and the vectorised extent to be a multiple of 4.
    const bool a_vec = (a_kc ? (K % 4 == 0 && vec_ok(A, sa_m)) : (sa_m == 1 && M % 4 == 0 && vec_ok(A, sa_k))) &&
                       (amask == nullptr || psvae_aligned16(amask));
    const bool b_vec = b_kc ? (K % 4 == 0 && vec_ok(B, sb_n)) : (sb_n == 1 && N % 4 == 0 && vec_ok(B, sb_k));
    const bool panel = a_vec && b_vec && (!rowsum || !b_kc) && M >= 32;
    int nc = 1, bn = 32, tn = 1;
    if (panel) panel_shape(n_ext, nc, bn, tn);
    const int split_asked = split_k;
    split_k = resolve_split(M, N, K, split_k);
    // (a short K leaves the 64 x 320 panels on a handful of CUs with long serial k loops: 64 x 64 tiles then)
    const bool two_by_two = panel && !a_kc && !b_kc && prefer_2x2(M, n_ext) &&
                            (long long)psvae_cdiv(M, 64) * split_k >= TWO_BY_TWO_MIN_BLOCKS;
    // 64 x 64 tiles where the panels (with their K split) would leave the chip mostly idle
    const bool tile64 = panel && !two_by_two && (long long)psvae_cdiv(M, 128) * nc * split_k < TILE64_BELOW_BLOCKS;
    if (tile64 && split_asked == 0) split_k = tile64_split(M, n_ext, K);
    const int kq = tile64 ? TILE64_BK : 32;
    int k_chunk = psvae_cdiv(psvae_cdiv(K, split_k), kq) * kq;
    if (k_chunk == 0) k_chunk = kq;
    split_k = K > 0 ? psvae_cdiv(K, k_chunk) : 1;
    PSVAE_REQUIRE(split_k <= 65535, PSVAE_ETOOLARGE, "psvae_gemm_f32: split_k %d too large", split_k);
    if (split_k > 1) {
        PSVAE_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)split_k * (size_t)M * (size_t)(N + 4) * sizeof(float),
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
    g.relu_bits = relu_bits, g.gate_bits = gate_bits, g.bits_ng = psvae_cdiv(N, 32);
    g.cls_w = nullptr, g.cls_out = nullptr;
    if (relu_bits || gate_bits) {
        PSVAE_REQUIRE((tile64 || panel) && split_k == 1 && g.c_vec != 0 && rowsum == nullptr, PSVAE_EINVAL,
                      "psvae_gemm_f32_bits: needs the 16-byte-aligned kernels, no K split and no rowsum "
                      "(M=%d N=%d K=%d split_k=%d)", M, N, K, split_k);
    }
    if (tile64) {
        g.nt_m = psvae_cdiv(M, 64), g.nt_n = psvae_cdiv(g.n_ext, 64), g.bn = 64, g.wn = 2;
        const long long blocks = (long long)psvae_cdiv((long long)g.nt_m * split_k, 8) * 8 * g.nt_n;
        if ((long long)psvae_cdiv(M, 32) * psvae_cdiv(g.n_ext, 32) * split_k <= TILE32_MAX_TILES) {
            g.nt_m = psvae_cdiv(M, 32), g.nt_n = psvae_cdiv(g.n_ext, 32), g.bn = 32, g.wn = 1;
            const long long blocks32 = (long long)psvae_cdiv((long long)g.nt_m * split_k, 8) * 8 * g.nt_n;
            launch_tile32(g, a_kc, b_kc, dim3((unsigned)blocks32), s);
        } else {
            launch_tile64(g, a_kc, b_kc, dim3((unsigned)blocks), s);
        }
    } else if (panel) {
        g.nt_m = psvae_cdiv(M, 128), g.nt_n = nc, g.bn = bn, g.wn = 1;
        if (two_by_two) {
            g.wn = 2, g.nt_m = psvae_cdiv(M, 64), g.nt_n = 1, g.bn = 320;
            tn = 5;
        }
        const long long blocks = (long long)psvae_cdiv((long long)g.nt_m * split_k, 8) * 8 * g.nt_n;
        PSVAE_REQUIRE(blocks < (1ll << 31), PSVAE_ETOOLARGE, "psvae_gemm_f32: grid too large");
        dim3 grid((unsigned)blocks);
        switch (tn) {
            case 1: launch_panel<1, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 2: launch_panel<2, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 3: launch_panel<3, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 4: launch_panel<4, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 6: launch_panel<6, PSVAE_PANEL_BK>(g, a_kc, b_kc, grid, s); break;
            case 7: launch_panel<7, 16>(g, a_kc, b_kc, grid, s); break;
            case 10: launch_panel<10, 16>(g, a_kc, b_kc, grid, s); break;
            case 11: launch_panel<11, 16>(g, a_kc, b_kc, grid, s); break;
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

extern "C" int psvae_gemm_f32(int M, int N, int K, const float *A, const float *amask, int64_t sa_m,
                              int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_n, float *C, int64_t ldc,
                              const float *bias, float *rowsum, const float *cmask, int act, int accumulate,
                              int split_k, void *workspace, size_t workspace_bytes, void *stream) {
    return gemm_impl(M, N, K, A, amask, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, rowsum, cmask, act, accumulate, split_k,
                     workspace, workspace_bytes, stream, nullptr, nullptr);
}

// ---- X W^T + b, ReLU, and a 4-class layer on top, the [M,N] intermediate never written
extern "C" int psvae_gemm_f32_relu_classes_tiles(int N) {
    if (N <= 0 || N % 4 != 0) return 0;
    int nc, bn, tn;
    panel_shape(N, nc, bn, tn);
    return (tn == 5 || tn == 6) ? nc : 0;
}

extern "C" int psvae_gemm_f32_relu_classes(int M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw,
                                           const float *bias, const float *cls_w, float *partials, void *stream) {
    const char *fn = "psvae_gemm_f32_relu_classes";
    PSVAE_REQUIRE(M >= 0 && N > 0 && K > 0, PSVAE_EINVAL, "%s: bad sizes M=%d N=%d K=%d", fn, M, N, K);
    if (M == 0) return PSVAE_OK;
    const int tiles = psvae_gemm_f32_relu_classes_tiles(N);
    PSVAE_REQUIRE(tiles > 0, PSVAE_EINVAL, "%s: N=%d is not covered (column tiles of 129..192 columns, N a multiple of 4)", fn, N);
    PSVAE_REQUIRE(A && W && cls_w && partials && K % 4 == 0 && lda % 4 == 0 && ldw % 4 == 0 && lda >= K && ldw >= K &&
                      psvae_aligned16(A) && psvae_aligned16(W) && psvae_aligned16(cls_w) && psvae_aligned16(partials) &&
                      (bias == nullptr || psvae_aligned16(bias)), PSVAE_EINVAL,
                  "%s: operands must be 16-byte aligned with row strides and K multiples of 4", fn);
    int nc, bn, tn;
    panel_shape(N, nc, bn, tn);
    GemmArgs g;
    g.M = M, g.N = N, g.K = K, g.n_ext = N, g.ws_ld = (N + 3) / 4 * 4;
    g.A = A, g.amask = nullptr, g.sa_m = lda, g.sa_k = 1, g.B = W, g.sb_k = 1, g.sb_n = ldw;
    g.C = nullptr, g.ldc = N, g.bias = bias, g.rowsum = nullptr, g.cmask = nullptr, g.act = PSVAE_ACT_RELU, g.accumulate = 0;
    g.split_k = 1, g.k_chunk = psvae_cdiv(K, PSVAE_PANEL_BK) * PSVAE_PANEL_BK, g.ws = nullptr, g.c_vec = 2, g.tail_begin = K;
    g.nt_m = psvae_cdiv(M, 128), g.nt_n = nc, g.bn = bn, g.wn = 1;
    g.relu_bits = nullptr, g.gate_bits = nullptr, g.bits_ng = 0;
    g.cls_w = cls_w, g.cls_out = partials;
    const long long blocks = (long long)psvae_cdiv((long long)g.nt_m, 8) * 8 * g.nt_n;
    PSVAE_REQUIRE(blocks < (1ll << 31), PSVAE_ETOOLARGE, "%s: grid too large", fn);
    launch_panel_classes(g, tn, dim3((unsigned)blocks), (hipStream_t)stream);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" size_t psvae_gemm_f32_bits_words(int M, int N) {
    return M <= 0 || N <= 0 ? 0 : (size_t)psvae_cdiv(M, 32) * (size_t)psvae_cdiv(N, 32) * 16;
}

extern "C" int psvae_gemm_f32_bits(int M, int N, int K, const float *A, int64_t sa_m, int64_t sa_k, const float *B,
                                   int64_t sb_k, int64_t sb_n, float *C, int64_t ldc, const float *bias, int act,
                                   uint64_t *relu_bits, const uint64_t *gate_bits, void *stream) {
    PSVAE_REQUIRE(relu_bits != nullptr || gate_bits != nullptr, PSVAE_EINVAL, "psvae_gemm_f32_bits: no bit array given");
    return gemm_impl(M, N, K, A, nullptr, sa_m, sa_k, B, sb_k, sb_n, C, ldc, bias, nullptr, nullptr, act, 0, 1, nullptr, 0,
                     stream, (unsigned long long *)relu_bits, (const unsigned long long *)gate_bits);
}
