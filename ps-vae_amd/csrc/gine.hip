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
#include <mutex>
#include <type_traits>

#include "common.h"

#ifndef PSVAE_MASK16
#define PSVAE_MASK16 uint16_t   // word of the message bits for C <= 4 (A/B builds: uint32_t)
#endif

namespace {

__device__ __forceinline__ float4 relu_add(float4 a, float4 b) {
    return make_float4(fmaxf(a.x + b.x, 0.f), fmaxf(a.y + b.y, 0.f), fmaxf(a.z + b.z, 0.f), fmaxf(a.w + b.w, 0.f));
}
__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

// Index data of a block's node range, staged in LDS by one coalesced pass so that a lane's
// neighbour-row loads do not wait behind two more dependent global loads (row_ptr, then col).
// Blocks whose range holds more than STAGE_EDGES edges read the indices from global memory.
constexpr int STAGE_NODES = 64, STAGE_EDGES = 768;

struct StagedCsr {
    int ptr[STAGE_NODES + 1];
    int col[STAGE_EDGES];
    unsigned char cls[STAGE_EDGES];
};

// returns true when the indices of nodes [begin, end) are in `sc` (edge ids relative to e_base)
__device__ __forceinline__ bool stage_csr(StagedCsr &sc, int begin, int end, const int32_t *__restrict__ row_ptr,
                                          const int32_t *__restrict__ col, const uint8_t *__restrict__ cls,
                                          int &e_base) {
    const int n = end - begin;
    e_base = row_ptr[begin];
    const int n_e = row_ptr[end] - e_base;
    const bool ok = n <= STAGE_NODES && n_e <= STAGE_EDGES;   // block-uniform
    if (ok) {
        for (int i = threadIdx.x; i <= n; i += blockDim.x) sc.ptr[i] = row_ptr[begin + i] - e_base;
        for (int e = threadIdx.x; e < n_e; e += blockDim.x) sc.col[e] = col[e_base + e], sc.cls[e] = cls[e_base + e];
    }
    __syncthreads();
    return ok;
}

// ReLU decisions of the messages a node SENDS, for the backward: the message of node j along an
// edge of class k is ReLU(x[j] + table[k]), so its derivative depends on (j, k) only - 4 C bits per
// float4 column of a node (bit 4 k + e: element e under class k), one 16-bit word for C <= 4 and one
// 32-bit word for C <= 8, against the 16 bytes of the x values themselves.
// (The table rows are read from memory where they are used - L1 hits - rather than held in registers per
// thread: the register-held form was measured 25 % slower, these kernels live on occupancy.)
template <int MAXC>
__device__ __forceinline__ unsigned message_bits(float4 self, const float4 *__restrict__ table, int C, int cols, int c) {
    unsigned w = 0;
#pragma unroll
    for (int k = 0; k < MAXC; ++k) {   // (rows beyond C: their bits are never read)
        const float4 t = table[(k < C ? k : 0) * cols + c];
        w |= (unsigned)(self.x + t.x > 0.f) << (4 * k) | (unsigned)(self.y + t.y > 0.f) << (4 * k + 1) |
             (unsigned)(self.z + t.z > 0.f) << (4 * k + 2) | (unsigned)(self.w + t.w > 0.f) << (4 * k + 3);
    }
    return w;
}

template <int MAXC, typename WORD>   // WORD = void: no mask bits are written
__global__ void gine_aggregate_fwd_kernel(int V, int C, int cols, int slots, int nodes_per_block,
                                          const float4 *__restrict__ x, int ldx4,
                                          const float4 *__restrict__ table,
                                          const int32_t *__restrict__ row_ptr, const int32_t *__restrict__ col,
                                          const uint8_t *__restrict__ cls, float4 *__restrict__ out, int ldo4,
                                          WORD *__restrict__ mask) {
    __shared__ StagedCsr sc;
    const int c = threadIdx.x % cols, slot = threadIdx.x / cols;
    const int begin = blockIdx.x * nodes_per_block;
    const int end = min(V, begin + nodes_per_block);
    int e_base;
    const bool staged = stage_csr(sc, begin, end, row_ptr, col, cls, e_base);
    if (slot >= slots) return;
    for (int i = begin + slot; i < end; i += slots) {
        const int e0 = staged ? sc.ptr[i - begin] : row_ptr[i], e1 = staged ? sc.ptr[i - begin + 1] : row_ptr[i + 1];
        const float4 self = x[(long long)i * ldx4 + c];
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // atoms have at most a handful of bonds: four neighbour rows are requested at once
        // (indices clamped, contributions predicated) instead of one dependent load at a time
        for (int e = e0; e < e1; e += 4) {
            int src[4], k[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int ee = min(e + u, e1 - 1);
                src[u] = staged ? sc.col[ee] : col[ee], k[u] = staged ? sc.cls[ee] : cls[ee];
            }
            float4 xj[4], t[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                xj[u] = x[(long long)src[u] * ldx4 + c];
                t[u] = table[k[u] * cols + c];
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (e + u < e1) acc = add4(acc, relu_add(xj[u], t[u]));
        }
        store_streamed(&out[(long long)i * ldo4 + c], add4(acc, self));
        // (the extra 2-byte store per lane costs the kernel ~15 %: 32-bit words, or collecting a block's words
        // in LDS and writing them as one run, were measured no faster; the backward gains more than this loses)
        if constexpr (!std::is_void<WORD>::value) mask[(long long)i * cols + c] = (WORD)message_bits<MAXC>(self, table, C, cols, c);
    }
}

// Backward over the CSR by SOURCE node.  Each thread keeps the table-gradient partial of
// its column for every class; they are reduced over the slots of the block through LDS and
// written as one [C,H] slab per block.  (Staging the indices in LDS as the forward does was
// measured slower here: the index reads are L1 hits shared by the lanes of a row.)
// WORD = void: the ReLU decisions are recomputed from x and the table; otherwise they are read from
// the mask bits the forward left behind (x and table are not touched).  seg_add / node_graph
// (optional): a per-GRAPH row added to every node of the graph (the gradient of a pooled sum over
// the graph's nodes, which is the same row for all of them).
template <int MAXC, typename WORD>
__global__ void gine_aggregate_bwd_kernel(int V, int C, int cols, int slots, int nodes_per_block,
                                          const float4 *__restrict__ x, int ldx4,
                                          const float4 *__restrict__ table,
                                          const float4 *__restrict__ g, int ldg4,
                                          const int32_t *__restrict__ t_row_ptr,
                                          const int32_t *__restrict__ t_col, const uint8_t *__restrict__ t_cls,
                                          const float4 *__restrict__ add, int ldadd4, float4 *__restrict__ g_x,
                                          int ldgx4, float4 *__restrict__ partial, const WORD *__restrict__ mask,
                                          const float4 *__restrict__ seg_add, int ldsa4,
                                          const int32_t *__restrict__ node_graph) {
    constexpr bool BITS = !std::is_void<WORD>::value;
    extern __shared__ float4 sh[];   // [slots][C][cols]
    const int c = threadIdx.x % cols, slot = threadIdx.x / cols;
    const bool active = slot < slots;
    float4 gt[MAXC];
#pragma unroll
    for (int k = 0; k < MAXC; ++k) gt[k] = make_float4(0.f, 0.f, 0.f, 0.f);
    const int begin = blockIdx.x * nodes_per_block;
    const int end = min(V, begin + nodes_per_block);
    if (active) {
        for (int j = begin + slot; j < end; j += slots) {
            float4 xj = make_float4(0.f, 0.f, 0.f, 0.f);
            unsigned bits = 0;
            if constexpr (BITS) bits = mask[(long long)j * cols + c]; else xj = x[(long long)j * ldx4 + c];
            float4 acc = g[(long long)j * ldg4 + c];
            float4 extra = make_float4(0.f, 0.f, 0.f, 0.f);
            if (seg_add != nullptr) extra = seg_add[(long long)node_graph[j] * ldsa4 + c];
            if (add != nullptr) extra = add4(extra, add[(long long)j * ldadd4 + c]);
            const int e0 = t_row_ptr[j], e1 = t_row_ptr[j + 1];
            for (int eb = e0; eb < e1; eb += 4) {   // four destination rows in flight, as in the forward
                int dst[4], kc[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int ee = min(eb + u, e1 - 1);
                    dst[u] = t_col[ee], kc[u] = t_cls[ee];
                }
                float4 gd4[4], t4[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    gd4[u] = g[(long long)dst[u] * ldg4 + c];
                    if constexpr (BITS) t4[u] = make_float4(0.f, 0.f, 0.f, 0.f); else t4[u] = table[kc[u] * cols + c];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    if (eb + u >= e1) continue;
                    const float4 gd = gd4[u];
                    const int k = kc[u];
                    float4 m;
                    if constexpr (BITS) {
                        const unsigned b = bits >> (4 * k);
                        m.x = (b & 1u) ? gd.x : 0.f, m.y = (b & 2u) ? gd.y : 0.f;
                        m.z = (b & 4u) ? gd.z : 0.f, m.w = (b & 8u) ? gd.w : 0.f;
                    } else {
                        const float4 t = t4[u];
                        m.x = xj.x + t.x > 0.f ? gd.x : 0.f;
                        m.y = xj.y + t.y > 0.f ? gd.y : 0.f;
                        m.z = xj.z + t.z > 0.f ? gd.z : 0.f;
                        m.w = xj.w + t.w > 0.f ? gd.w : 0.f;
                    }
                    acc = add4(acc, m);
#pragma unroll
                    for (int kk = 0; kk < MAXC; ++kk)
                        if (kk == k) gt[kk] = add4(gt[kk], m);
                }
            }
            store_streamed(&g_x[(long long)j * ldgx4 + c], add4(acc, extra));
        }
    }
    if (active) {
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
// consecutive outputs; 32 thread groups take blocks b = q, q+32, ...; their sums are added in
// q order.
constexpr int REDUCE_GROUPS = 32;

__global__ __launch_bounds__(32 * REDUCE_GROUPS) void gine_table_reduce_kernel(int n_out, int n_blocks,
                                                                               const float *__restrict__ partial,
                                                                               float *__restrict__ g_table,
                                                                               int accumulate) {
    __shared__ float sh[REDUCE_GROUPS][32];
    const int q = threadIdx.x >> 5, l = threadIdx.x & 31;
    const int i = blockIdx.x * 32 + l;
    float s = 0.f;
    if (i < n_out) {
        // eight slabs requested at once (a chain of dependent adds over one load each is latency-bound), summed
        // in slab order all the same
        int b = q;
        for (; b + 7 * REDUCE_GROUPS < n_blocks; b += 8 * REDUCE_GROUPS) {
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = partial[(long long)(b + u * REDUCE_GROUPS) * n_out + i];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        for (; b < n_blocks; b += REDUCE_GROUPS) s += partial[(long long)b * n_out + i];
    }
    sh[q][l] = s;
    __syncthreads();
    if (q == 0 && i < n_out) {
        float t = sh[0][l];
#pragma unroll
        for (int k = 1; k < REDUCE_GROUPS; ++k) t += sh[k][l];
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
    // eight rows per thread slot, fewer where that would leave most CUs without a workgroup (a batch of 32
    // molecules has ~750 nodes: 24 workgroups of 8 sequential row trips against 188 of one)
    int rows = 8;
    while (rows > 1 && (long long)psvae_cdiv(V, rows * s.slots) < 512) rows >>= 1;
    s.nodes_per_block = rows * s.slots;
    s.blocks = psvae_cdiv(V, s.nodes_per_block);
    if (max_blocks > 0 && s.blocks > max_blocks) {
        s.nodes_per_block = psvae_cdiv(psvae_cdiv(V, max_blocks), s.slots) * s.slots;
        s.blocks = psvae_cdiv(V, s.nodes_per_block);
    }
    return s;
}

// Most workgroups the backward is cut into (each writes one [C,H] slab of table-gradient
// partials; more workgroups were measured slower: the slab reduction grows with them).
constexpr int BWD_MAX_BLOCKS = 1024;
int bwd_max_blocks() { return BWD_MAX_BLOCKS; }

int check_common(const char *fn, int V, int H, int C, const void *x, int ldx) {
    PSVAE_REQUIRE(V >= 0 && H > 0 && C > 0, PSVAE_EINVAL, "%s: bad sizes V=%d H=%d C=%d", fn, V, H, C);
    PSVAE_REQUIRE(H % 4 == 0 && ldx % 4 == 0 && ldx >= H, PSVAE_EINVAL,
                  "%s: H and row strides must be multiples of 4 (H=%d ld=%d)", fn, H, ldx);
    PSVAE_REQUIRE(H / 4 <= 1024, PSVAE_ETOOLARGE, "%s: H=%d too wide", fn, H);
    PSVAE_REQUIRE(V == 0 || (x != nullptr && psvae_aligned16(x)), PSVAE_EINVAL, "%s: x null or not 16-byte aligned", fn);
    return PSVAE_OK;
}

}  // namespace

static int aggregate_fwd_impl(const char *fn, int V, int H, int C, const float *x, int ldx, const float *table,
                              const int32_t *row_ptr, const int32_t *col, const uint8_t *cls, float *out, int ldo,
                              void *mask, hipStream_t s_) {
    int rc = check_common(fn, V, H, C, x, ldx);
    if (rc) return rc;
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(table && row_ptr && out && psvae_aligned16(table) && psvae_aligned16(out), PSVAE_EINVAL,
                  "%s: null or misaligned operand", fn);
    PSVAE_REQUIRE(ldo % 4 == 0 && ldo >= H, PSVAE_EINVAL, "%s: bad ldo %d", fn, ldo);
    const Shape s = shape_for(V, H, 0);
#define PSVAE_AGG_FWD(MAXC, WORD)                                                                                    \
    hipLaunchKernelGGL((gine_aggregate_fwd_kernel<MAXC, WORD>), dim3(s.blocks), dim3(s.threads), 0, s_, V, C, s.cols,  \
                       s.slots, s.nodes_per_block, (const float4 *)x, ldx / 4, (const float4 *)table, row_ptr, col,  \
                       cls, (float4 *)out, ldo / 4, (WORD *)mask)
    if (mask == nullptr) {
        PSVAE_AGG_FWD(0, void);
    } else if (C <= 4) {
        PSVAE_AGG_FWD(4, PSVAE_MASK16);
    } else {
        PSVAE_AGG_FWD(8, uint32_t);
    }
#undef PSVAE_AGG_FWD
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_gine_aggregate_fwd(int V, int H, int C, const float *x, int ldx, const float *table,
                                        const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                                        float *out, int ldo, void *stream) {
    return aggregate_fwd_impl("psvae_gine_aggregate_fwd", V, H, C, x, ldx, table, row_ptr, col, cls, out, ldo, nullptr,
                              (hipStream_t)stream);
}

extern "C" size_t psvae_gine_mask_bytes(int V, int H, int C) {
    if (V <= 0 || H <= 0 || C <= 0) return 0;
    return (size_t)V * (size_t)(H / 4) * (C <= 4 ? sizeof(PSVAE_MASK16) : sizeof(uint32_t));
}

extern "C" int psvae_gine_aggregate_fwd_bits(int V, int H, int C, const float *x, int ldx, const float *table,
                                             const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                                             float *out, int ldo, void *mask_bits, void *stream) {
    const char *fn = "psvae_gine_aggregate_fwd_bits";
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "%s: at most 8 edge classes (C=%d)", fn, C);
    PSVAE_REQUIRE(V <= 0 || (mask_bits != nullptr && (((uintptr_t)mask_bits) & (C <= 4 ? 1u : 3u)) == 0), PSVAE_EINVAL,
                  "%s: mask_bits null or misaligned", fn);
    return aggregate_fwd_impl(fn, V, H, C, x, ldx, table, row_ptr, col, cls, out, ldo, mask_bits, (hipStream_t)stream);
}

extern "C" size_t psvae_gine_aggregate_bwd_workspace_bytes(int V, int H, int C) {
    if (V <= 0 || H <= 0 || C <= 0) return 0;
    const Shape s = shape_for(V, H, bwd_max_blocks());
    return (size_t)s.blocks * C * H * sizeof(float);
}

// Checks + launches shared by the two backward entry points; `partial` holds one [C,H] slab of
// table-gradient partials per workgroup (at least workspace_bytes(V, H, C) bytes).
static int aggregate_bwd_impl(const char *fn, int V, int H, int C, const float *x, int ldx, const float *table,
                              const float *g, int ldg, const int32_t *t_row_ptr, const int32_t *t_col,
                              const uint8_t *t_cls, const float *add, int ldadd, float *g_x, int ldgx,
                              float *g_table, int accumulate_table, float *partial, hipStream_t s_,
                              const void *mask = nullptr, const float *seg_add = nullptr, int ldsa = 0,
                              const int32_t *node_graph = nullptr) {
    if (V == 0) return accumulate_table ? PSVAE_OK : (int)hipMemsetAsync(g_table, 0, (size_t)C * H * sizeof(float), s_);
    PSVAE_REQUIRE(accumulate_table == 0 || accumulate_table == 1 || accumulate_table == PSVAE_TABLE_PARTIALS_ONLY, PSVAE_EINVAL,
                  "%s: accumulate_table must be 0, 1 or PSVAE_TABLE_PARTIALS_ONLY", fn);
    PSVAE_REQUIRE(add == nullptr || (psvae_aligned16(add) && ldadd % 4 == 0 && ldadd >= H), PSVAE_EINVAL,
                  "%s: bad `add` operand", fn);
    PSVAE_REQUIRE((mask != nullptr || (table && psvae_aligned16(table))) && t_row_ptr && g && g_x && psvae_aligned16(g) &&
                      psvae_aligned16(g_x), PSVAE_EINVAL, "%s: null or misaligned operand", fn);
    PSVAE_REQUIRE(seg_add == nullptr || (node_graph != nullptr && psvae_aligned16(seg_add) && ldsa % 4 == 0 && ldsa >= H),
                  PSVAE_EINVAL, "%s: bad per-graph add operand", fn);
    PSVAE_REQUIRE(ldg % 4 == 0 && ldg >= H && ldgx % 4 == 0 && ldgx >= H, PSVAE_EINVAL, "%s: bad row stride", fn);
    const Shape s = shape_for(V, H, bwd_max_blocks());
    const size_t lds = (size_t)s.slots * C * s.cols * sizeof(float4);
    PSVAE_REQUIRE(lds <= 64 * 1024, PSVAE_ETOOLARGE, "%s: LDS need %zu", fn, lds);
#define PSVAE_AGG_BWD(MAXC, WORD)                                                                                    \
    hipLaunchKernelGGL((gine_aggregate_bwd_kernel<MAXC, WORD>), dim3(s.blocks), dim3(s.threads), lds, s_, V, C, s.cols, \
                       s.slots, s.nodes_per_block, (const float4 *)x, ldx / 4, (const float4 *)table,                \
                       (const float4 *)g, ldg / 4, t_row_ptr, t_col, t_cls, (const float4 *)add, ldadd / 4,          \
                       (float4 *)g_x, ldgx / 4, (float4 *)partial, (const WORD *)mask, (const float4 *)seg_add,      \
                       ldsa / 4, node_graph)
    if (mask == nullptr) {
        if (C <= 4) PSVAE_AGG_BWD(4, void); else PSVAE_AGG_BWD(8, void);
    } else {
        if (C <= 4) PSVAE_AGG_BWD(4, PSVAE_MASK16); else PSVAE_AGG_BWD(8, uint32_t);
    }
#undef PSVAE_AGG_BWD
    PSVAE_CHECK_LAUNCH(fn);
    if (accumulate_table == PSVAE_TABLE_PARTIALS_ONLY) return PSVAE_OK;   // the caller reduces the slabs (psvae_gine_table_reduce)
    const int n_out = C * H;
    hipLaunchKernelGGL(gine_table_reduce_kernel, dim3(psvae_cdiv(n_out, 32)), dim3(32 * REDUCE_GROUPS), 0, s_, n_out,
                       s.blocks, (const float *)partial, g_table, accumulate_table);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_gine_aggregate_bwd(int V, int H, int C, const float *x, int ldx, const float *table,
                                        const float *g, int ldg, const int32_t *t_row_ptr,
                                        const int32_t *t_col, const uint8_t *t_cls, const float *add, int ldadd,
                                        float *g_x, int ldgx, float *g_table, int accumulate_table,
                                        void *workspace, size_t workspace_bytes, void *stream) {
    const char *fn = "psvae_gine_aggregate_bwd";
    int rc = check_common(fn, V, H, C, x, ldx);
    if (rc) return rc;
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "%s: at most 8 edge classes (C=%d)", fn, C);
    PSVAE_REQUIRE(g_table != nullptr, PSVAE_EINVAL, "%s: g_table is null", fn);
    PSVAE_REQUIRE(V == 0 || (workspace && psvae_aligned16(workspace) &&
                             workspace_bytes >= psvae_gine_aggregate_bwd_workspace_bytes(V, H, C)),
                  PSVAE_EWORKSPACE, "%s: workspace too small", fn);
    return aggregate_bwd_impl(fn, V, H, C, x, ldx, table, g, ldg, t_row_ptr, t_col, t_cls, add, ldadd, g_x, ldgx,
                              g_table, accumulate_table, (float *)workspace, (hipStream_t)stream);
}

// psvae_gine_aggregate_bwd with an optional per-graph row added to every node of its graph.
extern "C" int psvae_gine_aggregate_bwd_graph_add(int V, int H, int C, const float *x, int ldx, const float *table,
                                                  const float *g, int ldg, const int32_t *t_row_ptr,
                                                  const int32_t *t_col, const uint8_t *t_cls, const float *add, int ldadd,
                                                  const float *seg_add, int ld_seg_add, const int32_t *node_graph,
                                                  float *g_x, int ldgx, float *g_table, int accumulate_table,
                                                  void *workspace, size_t workspace_bytes, void *stream) {
    const char *fn = "psvae_gine_aggregate_bwd_graph_add";
    int rc = check_common(fn, V, H, C, x, ldx);
    if (rc) return rc;
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "%s: at most 8 edge classes (C=%d)", fn, C);
    PSVAE_REQUIRE(g_table != nullptr, PSVAE_EINVAL, "%s: g_table is null", fn);
    PSVAE_REQUIRE(V == 0 || (workspace && psvae_aligned16(workspace) &&
                             workspace_bytes >= psvae_gine_aggregate_bwd_workspace_bytes(V, H, C)),
                  PSVAE_EWORKSPACE, "%s: workspace too small", fn);
    return aggregate_bwd_impl(fn, V, H, C, x, ldx, table, g, ldg, t_row_ptr, t_col, t_cls, add, ldadd, g_x, ldgx, g_table,
                              accumulate_table, (float *)workspace, (hipStream_t)stream, nullptr, seg_add, ld_seg_add,
                              node_graph);
}

extern "C" int psvae_gine_aggregate_bwd_slabs(int V, int H, int C) {
    if (V <= 0 || H <= 0 || C <= 0) return 0;
    return shape_for(V, H, bwd_max_blocks()).blocks;
}

// g_table (+)= the sum of n_slabs [C,H] slabs of table-gradient partials, in slab order.
extern "C" int psvae_gine_table_reduce(int n_slabs, int C, int H, const float *partials, float *g_table, int accumulate,
                                       void *stream) {
    const char *fn = "psvae_gine_table_reduce";
    PSVAE_REQUIRE(n_slabs >= 0 && C > 0 && H > 0, PSVAE_EINVAL, "%s: bad sizes", fn);
    PSVAE_REQUIRE(g_table != nullptr && (n_slabs == 0 || partials != nullptr), PSVAE_EINVAL, "%s: null pointer", fn);
    const int n_out = C * H;
    if (n_slabs == 0) return accumulate ? PSVAE_OK : (int)hipMemsetAsync(g_table, 0, (size_t)n_out * sizeof(float), (hipStream_t)stream);
    hipLaunchKernelGGL(gine_table_reduce_kernel, dim3(psvae_cdiv(n_out, 32)), dim3(32 * REDUCE_GROUPS), 0, (hipStream_t)stream,
                       n_out, n_slabs, partials, g_table, accumulate ? 1 : 0);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

// The backward from the mask bits psvae_gine_aggregate_fwd_bits left behind (x and the table are not
// read), with an optional per-graph row added to every node of its graph.
extern "C" int psvae_gine_aggregate_bwd_bits(int V, int H, int C, const void *mask_bits, const float *g, int ldg,
                                             const int32_t *t_row_ptr, const int32_t *t_col, const uint8_t *t_cls,
                                             const float *add, int ldadd, const float *seg_add, int ld_seg_add,
                                             const int32_t *node_graph, float *g_x, int ldgx, float *g_table,
                                             int accumulate_table, void *workspace, size_t workspace_bytes,
                                             void *stream) {
    const char *fn = "psvae_gine_aggregate_bwd_bits";
    PSVAE_REQUIRE(V >= 0 && H > 0 && C > 0 && H % 4 == 0 && H / 4 <= 1024, PSVAE_EINVAL, "%s: bad sizes V=%d H=%d C=%d", fn, V,
                  H, C);
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "%s: at most 8 edge classes (C=%d)", fn, C);
    PSVAE_REQUIRE((g_table != nullptr || accumulate_table == PSVAE_TABLE_PARTIALS_ONLY) &&
                      (V == 0 || (mask_bits != nullptr && (((uintptr_t)mask_bits) & (C <= 4 ? 1u : 3u)) == 0)),
                  PSVAE_EINVAL, "%s: g_table null, or mask_bits null / misaligned", fn);
    if (V == 0 && accumulate_table == PSVAE_TABLE_PARTIALS_ONLY) return PSVAE_OK;   // no slab is written for no node
    PSVAE_REQUIRE(V == 0 || (workspace && psvae_aligned16(workspace) &&
                             workspace_bytes >= psvae_gine_aggregate_bwd_workspace_bytes(V, H, C)),
                  PSVAE_EWORKSPACE, "%s: workspace too small", fn);
    return aggregate_bwd_impl(fn, V, H, C, nullptr, H, nullptr, g, ldg, t_row_ptr, t_col, t_cls, add, ldadd, g_x, ldgx, g_table,
                              accumulate_table, (float *)workspace, (hipStream_t)stream, mask_bits, seg_add, ld_seg_add,
                              node_graph);
}

// ------------------------------------------------------------------------------------------
// The same backward for callers that cannot allocate (HIP graph capture, serving loops that
// forbid hipMalloc): the table-gradient partials go to a fixed device array of this library
// instead of a caller workspace.  Every GPU holds its own copy of the array at its own address,
// looked up once per device; all calls on one GPU share it, so calls that overlap on one GPU
// (two streams, two threads) race.  Nothing in it has to start at zero: the kernel writes every
// slab the reduction reads.
// ------------------------------------------------------------------------------------------
static __device__ __align__(256) float scratch[(size_t)BWD_MAX_BLOCKS * PSVAE_NOALLOC_MAX_CH];

namespace {
constexpr int MAX_DEVICES = 64;
float *g_scratch_ptr[MAX_DEVICES];   // device address of `scratch` on each GPU (null: not looked up yet)
std::mutex g_scratch_mu;

int scratch_of_current_device(const char *fn, float **out) {
    int dev = -1;
    hipError_t e = hipGetDevice(&dev);
    if (e != hipSuccess) {
        psvae_set_error("%s: hipGetDevice: %s", fn, hipGetErrorString(e));
        return (int)e;
    }
    PSVAE_REQUIRE(dev >= 0 && dev < MAX_DEVICES, PSVAE_ETOOLARGE, "%s: device index %d beyond %d", fn, dev, MAX_DEVICES);
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    if (g_scratch_ptr[dev] == nullptr) {
        float *p = nullptr;
        e = hipGetSymbolAddress((void **)&p, HIP_SYMBOL(scratch));
        if (e != hipSuccess || p == nullptr) {
            psvae_set_error("%s: hipGetSymbolAddress(scratch) on device %d: %s", fn, dev, hipGetErrorString(e));
            return e != hipSuccess ? (int)e : (int)hipErrorInvalidSymbol;
        }
        g_scratch_ptr[dev] = p;
    }
    *out = g_scratch_ptr[dev];
    return PSVAE_OK;
}
}  // namespace

extern "C" int psvae_gine_aggregate_bwd_noalloc(int V, int H, int C, const float *x, int ldx, const float *table,
                                                const float *g, int ldg, const int32_t *t_row_ptr,
                                                const int32_t *t_col, const uint8_t *t_cls, const float *add,
                                                int ldadd, float *g_x, int ldgx, float *g_table,
                                                int accumulate_table, void *stream) {
    const char *fn = "psvae_gine_aggregate_bwd_noalloc";
    int rc = check_common(fn, V, H, C, x, ldx);
    if (rc) return rc;
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "%s: at most 8 edge classes (C=%d)", fn, C);
    PSVAE_REQUIRE((long long)C * H <= PSVAE_NOALLOC_MAX_CH, PSVAE_ETOOLARGE,
                  "%s: C*H = %lld exceeds the fixed scratch (PSVAE_NOALLOC_MAX_CH = %d floats per workgroup)", fn,
                  (long long)C * H, PSVAE_NOALLOC_MAX_CH);
    PSVAE_REQUIRE(g_table != nullptr, PSVAE_EINVAL, "%s: g_table is null", fn);
    float *partial = nullptr;
    rc = scratch_of_current_device(fn, &partial);   // also on V == 0: a set-up call outside graph capture
    if (rc) return rc;
    return aggregate_bwd_impl(fn, V, H, C, x, ldx, table, g, ldg, t_row_ptr, t_col, t_cls, add, ldadd, g_x, ldgx,
                              g_table, accumulate_table, partial, (hipStream_t)stream);
}
