// Shared declarations of the f32 matrix-product kernels (gemm.hip, gemm_panel.h, gemm_tallk.hip).
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace psvae_gemm {

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
    int k_chunk;          // multiple of the kernel's k-tile depth
    float *ws;            // [split_k][M][ws_ld] when split_k > 1
    int ws_ld;            // n_ext rounded up to a multiple of 4
    int c_vec;            // C (and ws) rows may be written 16 bytes at a time; 2: and byte offsets fit 32 bits
    int nt_m, nt_n, bn;   // tiles and column-tile width
    int wn;               // wave grid of the panel kernel: 1 = 4 x 1, 2 = 2 x 2
    int tail_begin;       // rows tail_begin .. K-1 of the contraction are added by the reduce kernel (K: none)
    // Sign bits of the result, one 64-bit lane mask per (32 x 32 tile, row group i, component): see
    // psvae_gemm_f32_bits.  relu_bits: written (result > 0); gate_bits: read, the result is zeroed
    // where its bit is clear.  Vector panel path without K split only.
    unsigned long long *relu_bits;
    const unsigned long long *gate_bits;
    int bits_ng;          // 32-column groups per row of C: ceil(N / 32)
    // Class layer in the epilogue (psvae_gemm_f32_relu_classes): C is not written; every column tile adds its
    // part of relu(result + bias) . cls_w[c] for the 4 classes to cls_out[row][tile_n][c].
    const float *cls_w;   // [4][N]
    float *cls_out;       // [M][nt_n][4]
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

// Instantiated once per column-tile count in panel_inst.hip (one object file per TN, so that the
// variants compile in parallel).
template <int TN, int BK>
void launch_panel(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);
// the forward form X W^T with the class layer in the epilogue, TN = 5 or 6 (panel_cls_inst.hip)
void launch_panel_classes(const GemmArgs &g, int tn, dim3 grid, hipStream_t s);

// ---- 64 x 64 tiles for small products (tile64_inst.hip)
#ifndef PSVAE_TILE64_BK
#define PSVAE_TILE64_BK 64
#endif
constexpr int TILE64_BK = PSVAE_TILE64_BK;
void launch_tile64(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);
// 32 x 32 tiles whose TILE64_KG waves split K, the latency form (tile32_inst.hip)
constexpr int TILE64_KG = 4;
void launch_tile32(const GemmArgs &g, bool a_kc, bool b_kc, dim3 grid, hipStream_t s);

// ---- tall-K weight-gradient kernel (gemm_tallk.hip)
constexpr int TK_THREADS = 512, TK_BK = 16, TK_LDA = 304, TK_LDB = 176, TK_T = 5, TK_MAX = 304, TK_STAGES = 3;
constexpr int TK_HALF = 160;   // output columns per workgroup

struct TallKArgs {
    int M, N, K, n_ext;   // K: a multiple of TK_BK
    const float *A;       // A(m,k) = A[k * lda + m]
    long long lda;
    const float *B;       // B(k,n) = B[k * ldb + n]
    long long ldb;
    float *ws;            // [chunks][M][ws_ld]
    int ws_ld, k_chunk, halves, chunks;
};

constexpr int TALLK_MIN_K = 16384;   // below this the panel kernel's split-K does as well
int tallk_split(int K, int n_ext);
int launch_tallk(const TallKArgs &t, hipStream_t s);

}  // namespace psvae_gemm
