// GINE message passing: gather + ReLU + segmented (CSR) sum, forward and backward.
//
// Restates the aggregation of torch_geometric.nn.GINEConv as the reference calls it
// (src/modules/encoder.py:57-59, eps = 0, aggr = add):
//     out[i] = x[i] + sum_{e : dst(e) = i} ReLU(x[src(e)] + table[cls(e)])
// The reference materialises x[edge_index[0]] and an [E,H] edge tensor and scatter-adds;
// here the bond class indexes a C x H table and the sum runs over the CSR row of the
// destination node, so nothing of size E x H touches memory and no atomics are needed
// (the sum order is the CSR order: fixed, run-to-run reproducible).
//
// Thread mapping: a row of H floats is H/4 float4 columns; a workgroup holds SLOTS rows at
// a time, thread (slot, c) owns column c of the rows slot, slot+SLOTS, ... of the block's
// node range.  Each wave therefore reads whole contiguous row segments (16 B per lane).
// Molecules are small and contiguous in node order, so the neighbour rows a block gathers
// are the rows it (or the adjacent block) streams anyway: they are served by L1/L2.
#include "common.h"

namespace {

__device__ __forceinline__ float4 relu_add(float4 a, float4 b) {
    return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

__global__ void gine_aggregate_fwd_kernel(int V, int cols, int slots, int nodes_per_block,
                                          const float4 *__restrict__ x, int ldx4,
                                          const float4 *__restrict__ table,
                                          const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                          const uint8_t *__restrict__ cls, float4 *__restrict__ out, int ldo4) {
    const int c = threadIdx.x % cols, slot = threadIdx.x / cols;
    if (slot >= slots) return;
    const int begin = blockIdx.x * nodes_per_block;
    const int end = min(V, begin + nodes_per_block);
    for (int i = begin + slot; i < end; i += slots) {
        const int e0 = row_ptr[i], e1 = row_ptr[i + 1];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int e = e0; e < e1; ++e) {
            const float4 xj = x[(long long)col[e] * ldx4 + c];
            const float4 t = table[(int)cls[e] * cols + c];
            acc = add4(acc, relu_add(xj, t));
        }
        out[(long long)i * ldo4 + c] = add4(acc, x[(long long)i * ldx4 + c]);
    }
}

// Backward over the CSR by SOURCE node.  Each thread keeps the table-gradient partial of
// its column for every class; they are reduced over the slots of the block through LDS and
// written as one [C,H] slab per block.
template <int MAXC>
__global__ void gine_aggregate_bwd_kernel(int V, int C, int cols, int slots, int nodes_per_block,
                                          const float4 *__restrict__ x, int ldx4,
                                          const float4 *__restrict__ table,
                                          const float4 *__restrict__ g, int ldg4,
                                          const int32_t *__restrict__ t_row_ptr,
                                          const int32_t *__restrict__ t_col, const uint8_t *__restrict__ t_cls,
                                          const float4 *__restrict__ add, int ldadd4, float4 *__restrict__ g_x,
                                          int ldgx4, float4 *__restrict__ partial) {
    extern __shared__ float4 sh[];   // [slots][C][cols]
    const int c = threadIdx.x % cols, slot = threadIdx.x / cols;
    const bool active = slot < slots;
    float4 gt[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) gt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        const int begin = blockIdx.x * nodes_per_block;
        const int end = min(V, begin + nodes_per_block);
        for (int j = begin + slot; j < end; j += slots) {
            const float4 xj = x[(long long)j * ldx4 + c];
            float4 acc = g[(long long)j * ldg4 + c];
            const int e0 = t_row_ptr[j], e1 = t_row_ptr[j + 1];
            for (int e = e0; e < e1; ++e) {
                const int k = t_cls[e];
                const float4 gd = g[(long long)t_col[e] * ldg4 + c];
                const float4 t = table[k * cols + c];
                float4 m;
                m.x = xj.x + t.x > 0.f ? gd.x : 0.f;
                m.y = xj.y + t.y > 0.f ? gd.y : 0.f;
                m.z = xj.z + t.z > 0.f ? gd.z : 0.f;
                m.w = xj.w + t.w > 0.f ? gd.w : 0.f;
                acc = add4(acc, m);
#pragma unroll
                for (int kk = 0; kk < MAXC; ++kk)
                    if (kk == k) gt[kk] = add4(gt[kk], m);
            }
            if (add != nullptr) acc = add4(acc, add[(long long)j * ldadd4 + c]);
            g_x[(long long)j * ldgx4 + c] = acc;
        }
#pragma unroll
        for (int kk = 0; kk < MAXC; ++kk)
            if (kk < C) sh[(slot * C + kk) * cols + c] = gt[kk];
    }
    __syncthreads();
    if (active && slot == 0) {
        for (int kk = 0; kk < C; ++kk) {
            float4 s = sh[kk * cols + c];
            for (int sl = 1; sl < slots; ++sl) s = add4(s, sh[(sl * C + kk) * cols + c]);
            partial[((long long)blockIdx.x * C + kk) * cols + c] = s;
        }
    }
}

// g_table[i] (+)= sum over blocks b of partial[b][i] in a fixed order: a workgroup owns 32
// consecutive outputs; 8 thread groups take blocks b = q, q+8, ...; their sums are added in
// q order.
__global__ void gine_table_reduce_kernel(int n_out, int n_blocks, const float *__restrict__ partial,
                                         float *__restrict__ g_table, int accumulate) {
    __shared__ float sh[8][32];
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int i = blockIdx.x * 32 + l;
    float s = 0.f;
    if (i < n_out)
        for (int b = q; b < n_blocks; b += 8) s += partial[(long long)b * n_out + i];
    sh[q][l] = s;
    __syncthreads();
    if (q == 0 && i < n_out) {
        float t = sh[0][l];
#pragma unroll
        for (int k = 1; k < 8; ++k) t += sh[k][l];
        g_table[i] = accumulate ? g_table[i] + t : t;
    }
}

struct Shape {
    int cols, slots, threads, nodes_per_block, blocks;
};

Shape shape_for(int V, int H, int max_blocks) {
    Shape s;
    s.cols = H / 4;
    s.slots = s.cols >= 256 ? 1 : psvae_cdiv(256, s.cols);
    s.threads = psvae_cdiv(s.cols * s.slots, 64) * 64;
    s.nodes_per_block = 8 * s.slots;
    s.blocks = psvae_cdiv(V, s.nodes_per_block);
    if (max_blocks > 0 && s.blocks > max_blocks) {
        s.nodes_per_block = psvae_cdiv(psvae_cdiv(V, max_blocks), s.slots) * s.slots;
        s.blocks = psvae_cdiv(V, s.nodes_per_block);
    }
    return s;
}

constexpr int BWD_MAX_BLOCKS = 1024;

int check_common(const char *fn, int V, int H, int C, const void *x, int ldx) {
    PSVAE_REQUIRE(V >= 0 && H > 0 && C > 0, PSVAE_EINVAL, "%s: bad sizes V=%d H=%d C=%d", fn, V, H, C);
    PSVAE_REQUIRE(H % 4 == 0 && ldx % 4 == 0 && ldx >= H, PSVAE_EINVAL,
                  "%s: H and row strides must be multiples of 4 (H=%d ld=%d)", fn, H, ldx);
    PSVAE_REQUIRE(H / 4 <= 1024, PSVAE_ETOOLARGE, "%s: H=%d too wide", fn, H);
    PSVAE_REQUIRE(V == 0 || (x != nullptr && psvae_aligned16(x)), PSVAE_EINVAL, "%s: x null or not 16-byte aligned", fn);
    return PSVAE_OK;
}

}  // namespace

extern "C" int psvae_gine_aggregate_fwd(int V, int H, int C, const float *x, int ldx, const float *table,
                                        const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                                        float *out, int ldo, void *stream) {
    int rc = check_common("psvae_gine_aggregate_fwd", V, H, C, x, ldx);
    if (rc) return rc;
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(table && row_ptr && out && psvae_aligned16(table) && psvae_aligned16(out), PSVAE_EINVAL,
                  "psvae_gine_aggregate_fwd: null or misaligned operand");
    PSVAE_REQUIRE(ldo % 4 == 0 && ldo >= H, PSVAE_EINVAL, "psvae_gine_aggregate_fwd: bad ldo %d", ldo);
    const Shape s = shape_for(V, H, 0);
    hipLaunchKernelGGL(gine_aggregate_fwd_kernel, dim3(s.blocks), dim3(s.threads), 0, (hipStream_t)stream, V,
                       s.cols, s.slots, s.nodes_per_block, (const float4 *)x, ldx / 4, (const float4 *)table,
                       row_ptr, col, cls, (float4 *)out, ldo / 4);
    PSVAE_CHECK_LAUNCH("psvae_gine_aggregate_fwd");
    return PSVAE_OK;
}

extern "C" size_t psvae_gine_aggregate_bwd_workspace_bytes(int V, int H, int C) {
    if (V <= 0 || H <= 0 || C <= 0) return 0;
    const Shape s = shape_for(V, H, BWD_MAX_BLOCKS);
    return (size_t)s.blocks * C * H * sizeof(float);
}

extern "C" int psvae_gine_aggregate_bwd(int V, int H, int C, const float *x, int ldx, const float *table,
                                        const float *g, int ldg, const int32_t *t_row_ptr,
                                        const int32_t *t_col, const uint8_t *t_cls, const float *add, int ldadd,
                                        float *g_x, int ldgx, float *g_table, int accumulate_table,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    int rc = check_common("psvae_gine_aggregate_bwd", V, H, C, x, ldx);
    if (rc) return rc;
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "psvae_gine_aggregate_bwd: at most 8 edge classes (C=%d)", C);
    PSVAE_REQUIRE(g_table != nullptr, PSVAE_EINVAL, "psvae_gine_aggregate_bwd: g_table is null");
    hipStream_t s_ = (hipStream_t)stream;
    if (V == 0) return accumulate_table ? PSVAE_OK : (int)hipMemsetAsync(g_table, 0, (size_t)C * H * sizeof(float), s_);
    PSVAE_REQUIRE(add == nullptr || (psvae_aligned16(add) && ldadd % 4 == 0 && ldadd >= H), PSVAE_EINVAL,
                  "psvae_gine_aggregate_bwd: bad `add` operand");
    PSVAE_REQUIRE(table && t_row_ptr && g && g_x && psvae_aligned16(table) && psvae_aligned16(g) &&
                      psvae_aligned16(g_x), PSVAE_EINVAL, "psvae_gine_aggregate_bwd: null or misaligned operand");
    PSVAE_REQUIRE(ldg % 4 == 0 && ldg >= H && ldgx % 4 == 0 && ldgx >= H, PSVAE_EINVAL,
                  "psvae_gine_aggregate_bwd: bad row stride");
    PSVAE_REQUIRE(workspace && psvae_aligned16(workspace) &&
                      workspace_bytes >= psvae_gine_aggregate_bwd_workspace_bytes(V, H, C),
                  PSVAE_EWORKSPACE, "psvae_gine_aggregate_bwd: workspace too small");
    const Shape s = shape_for(V, H, BWD_MAX_BLOCKS);
    const size_t lds = (size_t)s.slots * C * s.cols * sizeof(float4);
    PSVAE_REQUIRE(lds <= 64 * 1024, PSVAE_ETOOLARGE, "psvae_gine_aggregate_bwd: LDS need %zu", lds);
    if (C <= 4)
        hipLaunchKernelGGL(gine_aggregate_bwd_kernel<4>, dim3(s.blocks), dim3(s.threads), lds, s_, V, C, s.cols,
                           s.slots, s.nodes_per_block, (const float4 *)x, ldx / 4, (const float4 *)table,
                           (const float4 *)g, ldg / 4, t_row_ptr, t_col, t_cls, (const float4 *)add, ldadd / 4, (float4 *)g_x,
                           ldgx / 4,
                           (float4 *)workspace);
    else
        hipLaunchKernelGGL(gine_aggregate_bwd_kernel<8>, dim3(s.blocks), dim3(s.threads), lds, s_, V, C, s.cols,
                           s.slots, s.nodes_per_block, (const float4 *)x, ldx / 4, (const float4 *)table,
                           (const float4 *)g, ldg / 4, t_row_ptr, t_col, t_cls, (const float4 *)add, ldadd / 4, (float4 *)g_x,
                           ldgx / 4,
                           (float4 *)workspace);
    PSVAE_CHECK_LAUNCH("psvae_gine_aggregate_bwd");
    const int n_out = C * H;
    hipLaunchKernelGGL(gine_table_reduce_kernel, dim3(psvae_cdiv(n_out, 32)), dim3(256), 0, s_, n_out, s.blocks,
                       (const float *)workspace, g_table, accumulate_table);
    PSVAE_CHECK_LAUNCH("psvae_gine_aggregate_bwd (table reduce)");
    return PSVAE_OK;
}
