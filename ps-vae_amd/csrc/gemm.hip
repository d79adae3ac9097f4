// f32 matrix products on the gfx950 matrix cores (v_mfma_f32_32x32x2_f32).
//
// One kernel serves the three products of a Linear layer (reference:
// src/modules/common_nn.py:8-23 and its autograd): operands are addressed through element
// strides, so X W^T, dY W and dY^T X differ only in the strides passed.  The MFMA is an exact
// f32 fused-multiply-add chain in k order, so results match a plain f32 dot product to
// rounding of the summation order.
//
// Tiling: 256 threads = 4 waves (64 lanes each) per workgroup, BK = 32.  Tiles are staged in
// LDS k-major (T[k][x], row stride BX+1) so that an MFMA operand fetch is one conflict-free
// ds_read_b32 per lane (lanes 0-31 read consecutive x at k, lanes 32-63 at k+1 — exactly the
// 32x32x2 operand layout).  The next tile's global loads are issued before the current
// tile's MFMAs (register prefetch).  Workgroups that share an A row-panel are placed on one
// XCD (blockIdx % 8) so the panel is fetched into one L2 only.
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

namespace {

constexpr int BK = 32;
constexpr int THREADS = 256;

struct GemmArgs {
    int M, N, K;
    const float *A;
    const float *amask;   // optional, same strides as A: A(m,k) counts as 0 where amask(m,k) <= 0
    long long sa_m, sa_k;
    const float *B;
    long long sb_k, sb_n;
    float *C;
    long long ldc;
    const float *bias;
    int act;
    int accumulate;
    int split_k;
    int k_chunk;     // multiple of BK
    float *ws;       // [split_k][M][N] when split_k > 1
    int nt_m, nt_n;
};

__device__ __forceinline__ void epilogue_store(const GemmArgs &g, int row, int col, float v) {
    if (g.bias) v += g.bias[col];
    if (g.act == PSVAE_ACT_RELU) v = fmaxf(v, 0.f);
    float *c = g.C + (long long)row * g.ldc + col;
    if (g.accumulate) v += *c;
    *c = v;
}

// Global -> registers for one BK x BX tile of a strided operand; element (x, k) lives at
// p[x*sx + k*sk].  KCONTIG: lanes run along k (coalesced when sk == 1); otherwise along x.
// `pm` (optional) gates the element: it counts as 0 where pm[same offset] <= 0 (the ReLU
// derivative applied to a gradient operand while it is loaded).
template <int BX, bool KCONTIG>
__device__ __forceinline__ void load_tile(float (&r)[BX / 8], const float *__restrict__ p,
                                          const float *__restrict__ pm, long long sx, long long sk, int x0,
                                          int k0, int X, int k_end, int tid) {
    if (KCONTIG) {
        const int k = k0 + (tid & 31), xr = x0 + (tid >> 5);
        const bool kok = k < k_end;
#pragma unroll
        for (int i = 0; i < BX / 8; ++i) {
            const int x = xr + 8 * i;
            const long long off = (long long)x * sx + (long long)k * sk;
            float v = (kok && x < X) ? p[off] : 0.f;
            if (pm != nullptr && kok && x < X && !(pm[off] > 0.f)) v = 0.f;
            r[i] = v;
        }
    } else {
        constexpr int KS = THREADS / BX;
        const int x = x0 + (tid % BX), kb = k0 + (tid / BX);
        const bool xok = x < X;
#pragma unroll
        for (int i = 0; i < BX / 8; ++i) {
            const int k = kb + KS * i;
            const long long off = (long long)x * sx + (long long)k * sk;
            float v = (xok && k < k_end) ? p[off] : 0.f;
            if (pm != nullptr && xok && k < k_end && !(pm[off] > 0.f)) v = 0.f;
            r[i] = v;
        }
    }
}

template <int BX, bool KCONTIG>
__device__ __forceinline__ void store_tile(float (*T)[BX + 1], const float (&r)[BX / 8], int tid) {
    if (KCONTIG) {
        const int k = tid & 31, xr = tid >> 5;
#pragma unroll
        for (int i = 0; i < BX / 8; ++i) T[k][xr + 8 * i] = r[i];
    } else {
        constexpr int KS = THREADS / BX;
        const int x = tid % BX, kb = tid / BX;
#pragma unroll
        for (int i = 0; i < BX / 8; ++i) T[kb + KS * i][x] = r[i];
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N, bool A_KC, bool B_KC>
__global__ __launch_bounds__(THREADS) void gemm_f32_kernel(GemmArgs g) {
    static_assert(WAVES_M * WAVES_N == 4, "4 waves per workgroup");
    constexpr int WM = BM / WAVES_M, WN = BN / WAVES_N;
    constexpr int TM = WM / 32, TN = WN / 32;
    __shared__ float As[BK][BM + 1];
    __shared__ float Bs[BK][BN + 1];

    // XCD-aware tile assignment: the nt_n column tiles of one row panel get consecutive
    // local ids on one XCD.
    const int bid = blockIdx.x;
    const int xcd = bid & 7, local = bid >> 3;
    const int tile_n = local % g.nt_n;
    const int tile_m = (local / g.nt_n) * 8 + xcd;
    if (tile_m >= g.nt_m) return;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int z = blockIdx.y;
    const int k_begin = z * g.k_chunk;
    const int k_end = min(g.K, k_begin + g.k_chunk);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WAVES_N, wn = wave % WAVES_N;
    const int l31 = lane & 31, lhi = lane >> 5;

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    float ra[BM / 8], rb[BN / 8];
    if (k_begin < k_end) {
        load_tile<BM, A_KC>(ra, g.A, g.amask, g.sa_m, g.sa_k, m0, k_begin, g.M, k_end, tid);
        load_tile<BN, B_KC>(rb, g.B, nullptr, g.sb_n, g.sb_k, n0, k_begin, g.N, k_end, tid);
    }
    for (int k0 = k_begin; k0 < k_end; k0 += BK) {
        store_tile<BM, A_KC>(As, ra, tid);
        store_tile<BN, B_KC>(Bs, rb, tid);
        __syncthreads();
        if (k0 + BK < k_end) {
            load_tile<BM, A_KC>(ra, g.A, g.amask, g.sa_m, g.sa_k, m0, k0 + BK, g.M, k_end, tid);
            load_tile<BN, B_KC>(rb, g.B, nullptr, g.sb_n, g.sb_k, n0, k0 + BK, g.N, k_end, tid);
        }
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a[TM], b[TN];
#pragma unroll
            for (int i = 0; i < TM; ++i) a[i] = As[kk + lhi][wm * WM + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < TN; ++j) b[j] = Bs[kk + lhi][wn * WN + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < TM; ++i)
#pragma unroll
                for (int j = 0; j < TN; ++j)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        __syncthreads();
    }

    // C/D layout of the 32x32 MFMA: col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5).
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + wn * WN + j * 32 + l31;
            if (col >= g.N) continue;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + wm * WM + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row >= g.M) continue;
                if (g.split_k > 1)
                    g.ws[((long long)z * g.M + row) * g.N + col] = acc[i][j][r];
                else
                    epilogue_store(g, row, col, acc[i][j][r]);
            }
        }
}

// Sum the split-K slabs in slab order and apply the epilogue.
__global__ void gemm_splitk_reduce_kernel(GemmArgs g) {
    const long long total = (long long)g.M * g.N;
    for (long long idx = (long long)blockIdx.x * blockDim.x + threadIdx.x; idx < total;
         idx += (long long)gridDim.x * blockDim.x) {
        float v = 0.f;
        for (int z = 0; z < g.split_k; ++z) v += g.ws[(long long)z * total + idx];
        epilogue_store(g, (int)(idx / g.N), (int)(idx % g.N), v);
    }
}

template <int BM, int BN, int WAVES_M, int WAVES_N>
void launch_cfg(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s) {
    if (a_kc && b_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, true, true>), grid, dim3(THREADS), 0, s, g);
    else if (a_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, true, false>), grid, dim3(THREADS), 0, s, g);
    else if (b_kc)
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, true>), grid, dim3(THREADS), 0, s, g);
    else
        hipLaunchKernelGGL((gemm_f32_kernel<BM, BN, WAVES_M, WAVES_N, false, false>), grid, dim3(THREADS), 0, s, g);
}

}  // namespace

// split_k == 0: pick a split that fills the chip when the output has few tiles and K is long
// (the weight-gradient products: M, N ~ 300, K = number of rows in the batch).
static int resolve_split(int M, int N, int K, int split_k) {
    if (split_k >= 1) return split_k;
    const long long tiles = (long long)psvae_cdiv(M, 64) * psvae_cdiv(N, 64);
    if (tiles >= 256 || K <= 2048) return 1;
    long long s = 1024 / tiles;
    const long long by_k = K / 512;
    if (s > by_k) s = by_k;
    return s < 1 ? 1 : (int)s;
}

extern "C" size_t psvae_gemm_f32_workspace_bytes(int M, int N, int K, int split_k) {
    split_k = resolve_split(M, N, K, split_k);
    if (split_k <= 1 || M <= 0 || N <= 0) return 0;
    return (size_t)split_k * (size_t)M * (size_t)N * sizeof(float);
}

extern "C" int psvae_gemm_f32(int M, int N, int K, const float *A, const float *amask, int64_t sa_m,
                              int64_t sa_k, const float *B, int64_t sb_k, int64_t sb_n, float *C, int64_t ldc,
                              const float *bias, int act, int accumulate, int split_k, void *workspace,
                              size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(M >= 0 && N >= 0 && K >= 0, PSVAE_EINVAL, "psvae_gemm_f32: negative dimension");
    if (M == 0 || N == 0) return PSVAE_OK;
    PSVAE_REQUIRE(C != nullptr && (K == 0 || (A != nullptr && B != nullptr)), PSVAE_EINVAL,
                  "psvae_gemm_f32: null operand");
    PSVAE_REQUIRE(ldc >= N, PSVAE_EINVAL, "psvae_gemm_f32: ldc %lld < N %d", (long long)ldc, N);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "psvae_gemm_f32: bad act %d", act);
    split_k = resolve_split(M, N, K, split_k);
    int k_chunk = psvae_cdiv(psvae_cdiv(K, split_k), BK) * BK;
    if (k_chunk == 0) k_chunk = BK;
    split_k = K > 0 ? psvae_cdiv(K, k_chunk) : 1;
    PSVAE_REQUIRE(split_k <= 65535, PSVAE_ETOOLARGE, "psvae_gemm_f32: split_k %d too large", split_k);
    if (split_k > 1) {
        PSVAE_REQUIRE(workspace != nullptr && workspace_bytes >= psvae_gemm_f32_workspace_bytes(M, N, K, split_k),
                      PSVAE_EWORKSPACE, "psvae_gemm_f32: workspace too small for split_k %d", split_k);
    }
    GemmArgs g;
    g.M = M, g.N = N, g.K = K;
    g.A = A, g.amask = amask, g.sa_m = sa_m, g.sa_k = sa_k;
    g.B = B, g.sb_k = sb_k, g.sb_n = sb_n;
    g.C = C, g.ldc = ldc, g.bias = bias, g.act = act;
    g.accumulate = accumulate, g.split_k = split_k, g.k_chunk = k_chunk, g.ws = (float *)workspace;
    const bool a_kc = (sa_k == 1), b_kc = (sb_k == 1);

    // Tile shape with the least padded work; ties go to the larger tile.
    const long long w0 = (long long)psvae_cdiv(M, 128) * 128 * psvae_cdiv(N, 128) * 128;
    const long long w1 = (long long)psvae_cdiv(M, 128) * 128 * psvae_cdiv(N, 64) * 64;
    const long long w2 = (long long)psvae_cdiv(M, 64) * 64 * psvae_cdiv(N, 64) * 64;
    int cfg = 0;
    long long best = w0;
    if (w1 < best) cfg = 1, best = w1;
    if (w2 < best) cfg = 2, best = w2;
    const int bm = cfg == 2 ? 64 : 128, bn = cfg == 0 ? 128 : 64;
    g.nt_m = psvae_cdiv(M, bm), g.nt_n = psvae_cdiv(N, bn);
    const long long blocks = (long long)psvae_cdiv(g.nt_m, 8) * 8 * g.nt_n;
    PSVAE_REQUIRE(blocks < (1ll << 31), PSVAE_ETOOLARGE, "psvae_gemm_f32: grid too large");
    dim3 grid((unsigned)blocks, (unsigned)split_k);
    hipStream_t s = (hipStream_t)stream;
    if (cfg == 0)
        launch_cfg<128, 128, 2, 2>(g, a_kc, b_kc, grid, s);
    else if (cfg == 1)
        launch_cfg<128, 64, 2, 2>(g, a_kc, b_kc, grid, s);
    else
        launch_cfg<64, 64, 2, 2>(g, a_kc, b_kc, grid, s);
    PSVAE_CHECK_LAUNCH("psvae_gemm_f32");
    if (split_k > 1) {
        const long long total = (long long)M * N;
        const int rb = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(rb), dim3(256), 0, s, g);
        PSVAE_CHECK_LAUNCH("psvae_gemm_f32 (split-K reduce)");
    }
    return PSVAE_OK;
}
