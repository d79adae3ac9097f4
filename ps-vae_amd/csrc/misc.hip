// The memory-bound helpers around the GEMMs: embedding gather, graph pooling, the
// reparameterised latent, pair-feature gather / scatter, the fused link loss, column sums.
#include <stdarg.h>

#include "common.h"

static thread_local char g_last_error[512] = "";

void psvae_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_last_error, sizeof(g_last_error), fmt, ap);
    va_end(ap);
}

extern "C" const char *psvae_last_error(void) { return g_last_error; }
extern "C" int psvae_version(void) { return 1; }

namespace {

__device__ __forceinline__ float4 add4(float4 a, float4 b) {
    return make_float4(a.x + b.x, a.y + b.y, a.z + b.z, a.w + b.w);
}

int grid_for(long long items, int threads, int cap = 4096) {
    long long b = (items + threads - 1) / threads;
    return (int)(b < 1 ? 1 : (b > cap ? cap : b));
}

// ---------------------------------------------------------------- embed_atom
// Every kernel that looks an id up clamps it into its table first (the same rule in the forward,
// the backward, the indicator and embed_trans), so a bad id can neither read nor add outside a
// table and all of them agree on which row it means.
__device__ __forceinline__ int clamp_id(int id, int n) { return min(max(id, 0), n - 1); }

__global__ void embed_atom_fwd_kernel(int V, const int32_t *__restrict__ x, const int32_t *__restrict__ xp,
                                      const int32_t *__restrict__ xq, const float *__restrict__ ea, int na, int da,
                                      const float *__restrict__ ep, int np, int dp, const float *__restrict__ eq,
                                      int nq, int dq, float *__restrict__ out) {
    const int D = da + dp + dq;
    const long long total = (long long)V * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / D), d = (int)(i - (long long)v * D);
        float val;
        if (d < da)
            val = ea[clamp_id(x[v], na) * da + d];
        else if (d < da + dp)
            val = ep[clamp_id(xp[v], np) * dp + (d - da)];
        else
            val = eq[clamp_id(xq[v], nq) * dq + (d - da - dp)];
        out[i] = val;
    }
}

__global__ void embed_atom_bwd_kernel(int V, const int32_t *__restrict__ x, const int32_t *__restrict__ xp,
                                      const int32_t *__restrict__ xq, const float *__restrict__ g, int na, int da,
                                      int np, int dp, int nq, int dq, float *__restrict__ ga,
                                      float *__restrict__ gp, float *__restrict__ gq) {
    const int D = da + dp + dq;
    const long long total = (long long)V * D;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / D), d = (int)(i - (long long)v * D);
        const float val = g[i];
        if (d < da)
            atomicAdd(&ga[clamp_id(x[v], na) * da + d], val);
        else if (d < da + dp)
            atomicAdd(&gp[clamp_id(xp[v], np) * dp + (d - da)], val);
        else
            atomicAdd(&gq[clamp_id(xq[v], nq) * dq + (d - da - dp)], val);
    }
}

__global__ void embed_atom_onehot_kernel(int V, const int32_t *__restrict__ x, const int32_t *__restrict__ xp,
                                         const int32_t *__restrict__ xq, int na, int np, int nq,
                                         float4 *__restrict__ out, int ld4) {
    const long long total = (long long)V * ld4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / ld4), j = (int)(i - (long long)v * ld4) * 4;
        const int a = clamp_id(x[v], na), b = na + clamp_id(xp[v], np), c = na + np + clamp_id(xq[v], nq);
        float4 o;
        o.x = (j == a || j == b || j == c) ? 1.f : 0.f;
        o.y = (j + 1 == a || j + 1 == b || j + 1 == c) ? 1.f : 0.f;
        o.z = (j + 2 == a || j + 2 == b || j + 2 == c) ? 1.f : 0.f;
        o.w = (j + 3 == a || j + 3 == b || j + 3 == c) ? 1.f : 0.f;
        out[i] = o;
    }
}

// ---------------------------------------------------------------- plain embedding lookup
// out[v] = table[ids[v]] (rows of D floats, D % 4 == 0); ids outside the table are clamped.
__global__ void embedding_fwd_kernel(int V, int d4, int n_rows, const int32_t *__restrict__ ids,
                                     const float4 *__restrict__ table, float4 *__restrict__ out) {
    const long long total = (long long)V * d4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / d4), c = (int)(i - (long long)v * d4);
        const int id = min(max(ids[v], 0), n_rows - 1);
        out[i] = table[(long long)id * d4 + c];
    }
}

// out[v][j] = (j == ids[v]) for j < ld (ld % 4 == 0): the indicator whose transpose times a
// gradient gives the table gradient deterministically on the matrix cores.
__global__ void onehot_kernel(int V, const int32_t *__restrict__ ids, float4 *__restrict__ out, int ld4) {
    const long long total = (long long)V * ld4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / ld4), j = (int)(i - (long long)v * ld4) * 4;
        const int a = ids[v];
        out[i] = make_float4(j == a ? 1.f : 0.f, j + 1 == a ? 1.f : 0.f, j + 2 == a ? 1.f : 0.f, j + 3 == a ? 1.f : 0.f);
    }
}

// ---------------------------------------------------------------- segment sum (embed_graph)
__global__ void segment_sum_fwd_kernel(int cols, const float4 *__restrict__ x, int ldx4,
                                       const int32_t *__restrict__ seg_begin, const int32_t *__restrict__ seg_end,
                                       float4 *__restrict__ out, int ldo4) {
    const int b = blockIdx.x;
    const int r0 = seg_begin[b], r1 = seg_end[b];
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        // eight rows requested at once (indices clamped, contributions predicated, summed in row order)
        for (int r = r0; r < r1; r += 8) {
            float4 v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = x[(long long)min(r + u, r1 - 1) * ldx4 + c];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (r + u < r1) acc = add4(acc, v[u]);
        }
        out[(long long)b * ldo4 + c] = acc;
    }
}

__global__ void segment_sum_bwd_kernel(int cols, const float4 *__restrict__ g_out, int ldgo4,
                                       const int32_t *__restrict__ seg_begin, const int32_t *__restrict__ seg_end,
                                       float4 *__restrict__ g_x, int ldgx4) {
    const int b = blockIdx.x;
    const int r0 = seg_begin[b], r1 = seg_end[b];
    for (int c = threadIdx.x; c < cols; c += blockDim.x) {
        const float4 v = g_out[(long long)b * ldgo4 + c];
        for (int r = r0; r < r1; ++r) g_x[(long long)r * ldgx4 + c] = v;
    }
}

// out[v] = q[node_graph[v]] gated by the sign bits of psvae_gemm_f32_bits over the [V, N] result
// (the input gradient through a Linear + ReLU whose incoming gradient is the same row for every node
// of a graph: the product runs once per graph, this kernel spreads and gates it).  A wave takes one
// 32 x 32 tile of `out` with the lane mapping of the product's epilogue: lane l owns columns
// 4 (l & 7) .. + 3 of rows (l >> 3) + 8 i, i = 0..3.
__global__ __launch_bounds__(256) void graph_rows_gated_kernel(int V, int N, const float *__restrict__ q, int ldq,
                                                               const int32_t *__restrict__ node_graph,
                                                               const unsigned long long *__restrict__ bits, int ng,
                                                               float *__restrict__ out, int ldo) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int rg = blockIdx.x;
    const int prow = lane >> 3, pc4 = (lane & 7) * 4;
    int graph[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int row = rg * 32 + prow + 8 * i;
        graph[i] = row < V ? node_graph[row] : 0;
    }
    for (int cg = wave; cg < ng; cg += 4) {
        const int col = cg * 32 + pc4;
        if (col >= N) continue;
        const unsigned long long *w = bits + ((long long)rg * ng + cg) * 16;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int row = rg * 32 + prow + 8 * i;
            if (row >= V) continue;
            float4 v = *reinterpret_cast<const float4 *>(q + (long long)graph[i] * ldq + col);
            v.x = (w[4 * i] >> lane) & 1ull ? v.x : 0.f;
            v.y = (w[4 * i + 1] >> lane) & 1ull ? v.y : 0.f;
            v.z = (w[4 * i + 2] >> lane) & 1ull ? v.z : 0.f;
            v.w = (w[4 * i + 3] >> lane) & 1ull ? v.w : 0.f;
            *reinterpret_cast<float4 *>(out + (long long)row * ldo + col) = v;
        }
    }
}

// ---------------------------------------------------------------- rsample + KL
// z and the KL sum in two stages: workgroups of 256 threads walk the B x L elements with a grid
// stride and leave one partial sum each (a fixed tree inside the workgroup); one workgroup adds the
// partials in a fixed tree.  A batch that fits one workgroup's walk writes kl directly.
constexpr int RS_THREADS = 256, RS_MAX_BLOCKS = 256, RS_PER_THREAD = 4;

__device__ __forceinline__ float rs_block_sum(float s, float *sh) {
    sh[threadIdx.x] = s;
    __syncthreads();
    for (int d = RS_THREADS / 2; d > 0; d >>= 1) {
        if (threadIdx.x < d) sh[threadIdx.x] += sh[threadIdx.x + d];
        __syncthreads();
    }
    return sh[0];
}

__global__ __launch_bounds__(RS_THREADS) void rsample_kl_fwd_kernel(int B, int L, const float *__restrict__ mean,
                                                                    const float *__restrict__ lv_pre,
                                                                    const float *__restrict__ eps, float *__restrict__ z,
                                                                    float *__restrict__ part, float *__restrict__ kl) {
    __shared__ float sh[RS_THREADS];
    const int total = B * L;
    float s = 0.f;
    for (int i = blockIdx.x * RS_THREADS + threadIdx.x; i < total; i += gridDim.x * RS_THREADS) {
        const float m = mean[i], lv = -fabsf(lv_pre[i]);
        s += 1.0f + lv - m * m - expf(lv);
        z[i] = m + expf(lv / 2) * eps[i];
    }
    s = rs_block_sum(s, sh);
    if (threadIdx.x == 0) {
        if (gridDim.x == 1)
            *kl = -0.5f * s / (float)B;
        else
            part[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(RS_THREADS) void rsample_kl_finish_kernel(int B, int n_part, const float *__restrict__ part,
                                                                       float *__restrict__ kl) {
    __shared__ float sh[RS_THREADS];
    const float s = rs_block_sum(threadIdx.x < n_part ? part[threadIdx.x] : 0.f, sh);
    if (threadIdx.x == 0) *kl = -0.5f * s / (float)B;
}

int rs_blocks(long long total) {
    const long long b = (total + RS_THREADS * RS_PER_THREAD - 1) / (RS_THREADS * RS_PER_THREAD);
    return (int)(b < 1 ? 1 : (b > RS_MAX_BLOCKS ? RS_MAX_BLOCKS : b));
}

__global__ void rsample_kl_bwd_kernel(int B, int L, const float *__restrict__ mean, const float *__restrict__ lv_pre,
                                      const float *__restrict__ eps, const float *__restrict__ g_z,
                                      const float *__restrict__ g_kl, float *__restrict__ g_mean,
                                      float *__restrict__ g_lv_pre) {
    const int total = B * L;
    const float gk = g_kl ? *g_kl : 0.f;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const float m = mean[i], pre = lv_pre[i];
        const float lv = -fabsf(pre);
        const float gz = g_z ? g_z[i] : 0.f;
        // d kl / d mean = mean / B ; d kl / d lv = -0.5 (1 - exp(lv)) / B
        g_mean[i] = gz + gk * m / (float)B;
        const float g_lv = gz * 0.5f * expf(lv / 2) * eps[i] + gk * (-0.5f) * (1.0f - expf(lv)) / (float)B;
        const float sgn = pre > 0.f ? 1.f : (pre < 0.f ? -1.f : 0.f);   // d(-|u|)/du = -sign(u)
        g_lv_pre[i] = -sgn * g_lv;
    }
}

// ---------------------------------------------------------------- GRU cell (piece LM)
__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

// PyTorch gate order (r, z, n): r = s(gi_r + gh_r), z = s(gi_z + gh_z),
// n = tanh(gi_n + r * gh_n), h = (1 - z) * n + z * h_prev.
__global__ void gru_cell_fwd_kernel(int B, int H, const float *__restrict__ gi, const float *__restrict__ gh,
                                    const float *__restrict__ hp, float *__restrict__ h, float *__restrict__ gates) {
    const int total = B * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, j = i - b * H;
        const long long o = (long long)b * 3 * H + j;
        const float r = sigmoidf_(gi[o] + gh[o]);
        const float z = sigmoidf_(gi[o + H] + gh[o + H]);
        const float n = tanhf(gi[o + 2 * H] + r * gh[o + 2 * H]);
        gates[o] = r, gates[o + H] = z, gates[o + 2 * H] = n;
        h[i] = (1.0f - z) * n + z * hp[i];
    }
}

__global__ void gru_cell_bwd_kernel(int B, int H, const float *__restrict__ gates, const float *__restrict__ gh,
                                    const float *__restrict__ hp, const float *__restrict__ dh,
                                    const float *__restrict__ dh_extra, float *__restrict__ dgi,
                                    float *__restrict__ dgh, float *__restrict__ dhp) {
    const int total = B * H;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int b = i / H, j = i - b * H;
        const long long o = (long long)b * 3 * H + j;
        const float r = gates[o], z = gates[o + H], n = gates[o + 2 * H];
        const float d = dh[i] + (dh_extra ? dh_extra[i] : 0.f);
        const float dn_pre = d * (1.0f - z) * (1.0f - n * n);
        const float dz_pre = d * (hp[i] - n) * z * (1.0f - z);
        const float dr_pre = dn_pre * gh[o + 2 * H] * r * (1.0f - r);
        dgi[o] = dr_pre, dgi[o + H] = dz_pre, dgi[o + 2 * H] = dn_pre;
        dgh[o] = dr_pre, dgh[o + H] = dz_pre, dgh[o + 2 * H] = dn_pre * r;
        dhp[i] = d * z;
    }
}

// ---------------------------------------------------------------- pair features
__global__ void pair_gather_fwd_kernel(int P, int h4, int l4, const float4 *__restrict__ h, int ldh4,
                                       const float4 *__restrict__ z, int ldz4, const int32_t *__restrict__ src,
                                       const int32_t *__restrict__ dst, const int32_t *__restrict__ graph,
                                       float4 *__restrict__ feats) {
    const int w4 = 2 * h4 + l4;
    const long long total = (long long)P * w4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i / w4), c = (int)(i - (long long)p * w4);
        float4 v;
        if (c < h4)
            v = h[(long long)src[p] * ldh4 + c];
        else if (c < 2 * h4)
            v = h[(long long)dst[p] * ldh4 + (c - h4)];
        else
            v = z[(long long)graph[p] * ldz4 + (c - 2 * h4)];
        feats[i] = v;
    }
}

// out[p] = act(a[src[p]] + b[dst[p]] + c[graph[p]]): the first link-predictor layer after its
// weight has been split by input block (h_src | h_dst | z), so that the three products run once
// per node / per graph instead of once per pair.
__global__ void pair_combine_fwd_kernel(int P, int f4, const float4 *__restrict__ a, int lda4,
                                        const float4 *__restrict__ b, int ldb4, const float4 *__restrict__ c,
                                        int ldc4, const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                        const int32_t *__restrict__ graph, int relu, float4 *__restrict__ out,
                                        int ldo4) {
    const long long total = (long long)P * f4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i / f4), f = (int)(i - (long long)p * f4);
        const float4 u = a[(long long)src[p] * lda4 + f], v = b[(long long)dst[p] * ldb4 + f];
        const float4 w = c[(long long)graph[p] * ldc4 + f];
        float4 o = make_float4((u.x + v.x) + w.x, (u.y + v.y) + w.y, (u.z + v.z) + w.z, (u.w + v.w) + w.w);
        if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
        out[(long long)p * ldo4 + f] = o;
    }
}

// The same for rows of at most 256 float4: a thread keeps its column and walks pairs (no index division in the
// loop), blockDim / f4 pairs per pass and four passes requested before the first is used.
__global__ __launch_bounds__(256) void pair_combine_fwd_rows_kernel(int P, int f4, const float4 *__restrict__ a, int lda4,
                                                                    const float4 *__restrict__ b, int ldb4,
                                                                    const float4 *__restrict__ c, int ldc4,
                                                                    const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                                    const int32_t *__restrict__ graph, int relu,
                                                                    float4 *__restrict__ out, int ldo4) {
    const int ppb = 256 / f4;
    const int tp = threadIdx.x / f4, f = threadIdx.x - tp * f4;
    if (tp >= ppb) return;
    const long long stride = (long long)gridDim.x * ppb;
    constexpr int U = 4;
    for (long long p0 = (long long)blockIdx.x * ppb + tp; p0 < P; p0 += U * stride) {
        int si[U], di[U], gi[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const long long p = p0 + u * stride < P ? p0 + u * stride : P - 1;
            si[u] = src[p], di[u] = dst[p], gi[u] = graph[p];
        }
        float4 x[U], y[U], w[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            x[u] = a[(long long)si[u] * lda4 + f], y[u] = b[(long long)di[u] * ldb4 + f];
            w[u] = c[(long long)gi[u] * ldc4 + f];
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (p0 + u * stride >= P) break;
            float4 o = make_float4((x[u].x + y[u].x) + w[u].x, (x[u].y + y[u].y) + w[u].y, (x[u].z + y[u].z) + w[u].z,
                                   (x[u].w + y[u].w) + w[u].w);
            if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
            out[(p0 + u * stride) * ldo4 + f] = o;
        }
    }
}

// One workgroup per graph: the a / b rows of the graph's nodes are staged in LDS (a block of cp4 float4 columns
// at a time) and every pair row is formed from there, so the kernel reads each node row once and is bound by
// the rows it writes - what counts when a graph has many more pairs than nodes (all candidate pairs of a
// generated molecule).  A graph with more than max_nodes nodes reads its rows from memory instead.
__global__ __launch_bounds__(256) void pair_combine_fwd_graph_kernel(int f4, int cp4, int max_nodes, const float4 *__restrict__ a,
                                                                     int lda4, const float4 *__restrict__ b, int ldb4,
                                                                     const float4 *__restrict__ c, int ldc4,
                                                                     const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                                                     const int32_t *__restrict__ pair_ptr,
                                                                     const int32_t *__restrict__ node_begin,
                                                                     const int32_t *__restrict__ node_end, int relu,
                                                                     float4 *__restrict__ out, int ldo4) {
    extern __shared__ float4 s_rows[];
    float4 *sA = s_rows, *sB = s_rows + (size_t)max_nodes * cp4;
    const int g = blockIdx.x;
    const int n0 = node_begin[g], n = node_end[g] - n0;
    const int p0 = pair_ptr[g], p1 = pair_ptr[g + 1];
    if (p0 >= p1) return;   // block-uniform
    const bool staged = n <= max_nodes;
    for (int f0 = 0; f0 < f4; f0 += cp4) {
        const int w = min(cp4, f4 - f0);
        if (staged) {
            for (int i = threadIdx.x; i < n * w; i += 256) {
                const int r = i / w, col = i - r * w;
                sA[r * cp4 + col] = a[(long long)(n0 + r) * lda4 + f0 + col];
                sB[r * cp4 + col] = b[(long long)(n0 + r) * ldb4 + f0 + col];
            }
        }
        __syncthreads();
        const int ppb = 256 / w, tp = threadIdx.x / w, f = threadIdx.x - tp * w;
        if (tp < ppb) {
            const float4 z = c[(long long)g * ldc4 + f0 + f];
            constexpr int U = 4;   // the node ids of four pairs are requested before the first row is formed
            for (int q = p0 + tp; q < p1; q += U * ppb) {
                int si[U], di[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = min(q + u * ppb, p1 - 1);
                    si[u] = src[p], di[u] = dst[p];
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int p = q + u * ppb;
                    if (p >= p1) break;
                    const float4 x = staged ? sA[(si[u] - n0) * cp4 + f] : a[(long long)si[u] * lda4 + f0 + f];
                    const float4 y = staged ? sB[(di[u] - n0) * cp4 + f] : b[(long long)di[u] * ldb4 + f0 + f];
                    float4 o = make_float4((x.x + y.x) + z.x, (x.y + y.y) + z.y, (x.z + y.z) + z.z, (x.w + y.w) + z.w);
                    if (relu) o = make_float4(fmaxf(o.x, 0.f), fmaxf(o.y, 0.f), fmaxf(o.z, 0.f), fmaxf(o.w, 0.f));
                    store_streamed(&out[(long long)p * ldo4 + f0 + f], o);
                }
            }
        }
        __syncthreads();
    }
}

// One workgroup per graph.  Pairs of the graph are staged CHUNK at a time in LDS (node ids
// local to the graph); thread (slot, c) sums column c of the matching feature rows for nodes
// slot, slot+slots, ...  The order of the sum is the pair order: deterministic.
constexpr int PAIR_CHUNK = 1024;
__global__ void pair_gather_bwd_kernel(int h4, int l4, int slots, const float4 *__restrict__ g_feats,
                                       const int32_t *__restrict__ src, const int32_t *__restrict__ dst,
                                       const int32_t *__restrict__ pair_ptr, const int32_t *__restrict__ node_begin,
                                       const int32_t *__restrict__ node_end, float4 *__restrict__ g_h, int ldgh4,
                                       float4 *__restrict__ g_z, int ldgz4) {
    __shared__ int s_src[PAIR_CHUNK], s_dst[PAIR_CHUNK];
    const int b = blockIdx.x;
    const int p0 = pair_ptr[b], p1 = pair_ptr[b + 1];
    const int n0 = node_begin[b], n1 = node_end[b];
    const int w4 = 2 * h4 + l4;
    const int c = threadIdx.x % h4, slot = threadIdx.x / h4;
    const bool active = slot < slots;
    float4 zacc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool zthread = (int)threadIdx.x < l4;
    bool first = true;
    int pc = p0;
    do {   // runs once even without pairs so that g_h / g_z are written
        const int cnt = min(PAIR_CHUNK, p1 - pc);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            s_src[i] = src[pc + i] - n0;
            s_dst[i] = dst[pc + i] - n0;
        }
        __syncthreads();
        if (active) {
            for (int j = slot; j < n1 - n0; j += slots) {
                float4 acc = first ? make_float4(0.f, 0.f, 0.f, 0.f) : g_h[(long long)(n0 + j) * ldgh4 + c];
                for (int i = 0; i < cnt; ++i) {
                    if (s_src[i] == j) acc = add4(acc, g_feats[(long long)(pc + i) * w4 + c]);
                    if (s_dst[i] == j) acc = add4(acc, g_feats[(long long)(pc + i) * w4 + h4 + c]);
                }
                g_h[(long long)(n0 + j) * ldgh4 + c] = acc;
            }
        }
        if (zthread)
            for (int i = 0; i < cnt; ++i) zacc = add4(zacc, g_feats[(long long)(pc + i) * w4 + 2 * h4 + threadIdx.x]);
        first = false;
        pc += cnt;
    } while (pc < p1);
    if (zthread) g_z[(long long)b * ldgz4 + threadIdx.x] = zacc;
}

// Backward of pair_combine_fwd with a = ab[:, :F], b = ab[:, F:] (one [V, 2F] matrix): the gradient
// of a pair's row, gated by the ReLU derivative (out > 0) when act is ReLU, is added to the a block of
// its source node, the b block of its destination node and the row of its graph.  One workgroup per
// graph.  The gated rows of up to `chunk` pairs are staged in LDS by all threads at once (independent
// 16-byte loads); then thread (slot, c) owns column c of nodes slot, slot+slots, ... and walks the
// staged pairs in order, so every sum runs in pair order and only touches LDS.
constexpr int PAIR_ROWS_MAX = 256, PAIR_ROWS_LDS_BYTES = 32 * 1024;
__global__ void pair_combine_bwd_kernel(int f4, int slots, int chunk, int relu, const float4 *__restrict__ g_out, int ldg4,
                                        const float4 *__restrict__ out, int ldo4, const int32_t *__restrict__ src,
                                        const int32_t *__restrict__ dst, const int32_t *__restrict__ pair_ptr,
                                        const int32_t *__restrict__ node_begin, const int32_t *__restrict__ node_end,
                                        float4 *__restrict__ g_ab, int ldab4, float4 *__restrict__ g_c, int ldc4) {
    extern __shared__ float4 s_rows[];   // [chunk][f4]
    __shared__ int s_src[PAIR_ROWS_MAX], s_dst[PAIR_ROWS_MAX];
    const int b = blockIdx.x;
    const int p0 = pair_ptr[b], p1 = pair_ptr[b + 1];
    const int n0 = node_begin[b], n1 = node_end[b];
    const int c = threadIdx.x % f4, slot = threadIdx.x / f4;
    const bool active = slot < slots;
    float4 cacc = make_float4(0.f, 0.f, 0.f, 0.f);
    const bool cthread = (int)threadIdx.x < f4;
    bool first = true;
    int pc = p0;
    do {   // runs once even without pairs so that the outputs are written
        const int cnt = min(chunk, p1 - pc);
        __syncthreads();
        for (int i = threadIdx.x; i < cnt; i += blockDim.x) {
            s_src[i] = src[pc + i] - n0;
            s_dst[i] = dst[pc + i] - n0;
        }
        for (int idx = threadIdx.x; idx < cnt * f4; idx += blockDim.x) {
            const int i = idx / f4, col = idx - i * f4;
            float4 g = g_out[(long long)(pc + i) * ldg4 + col];
            if (relu) {
                const float4 o = out[(long long)(pc + i) * ldo4 + col];
                g.x = o.x > 0.f ? g.x : 0.f, g.y = o.y > 0.f ? g.y : 0.f, g.z = o.z > 0.f ? g.z : 0.f, g.w = o.w > 0.f ? g.w : 0.f;
            }
            s_rows[idx] = g;
        }
        __syncthreads();
        if (active) {
            for (int j = slot; j < n1 - n0; j += slots) {
                float4 *ga = g_ab + (long long)(n0 + j) * ldab4 + c, *gb = ga + f4;
                float4 acc_a = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *ga;
                float4 acc_b = first ? make_float4(0.f, 0.f, 0.f, 0.f) : *gb;
                for (int i = 0; i < cnt; ++i) {
                    const bool is_src = s_src[i] == j, is_dst = s_dst[i] == j;
                    if (is_src || is_dst) {
                        const float4 m = s_rows[i * f4 + c];
                        if (is_src) acc_a = add4(acc_a, m);
                        if (is_dst) acc_b = add4(acc_b, m);
                    }
                }
                *ga = acc_a, *gb = acc_b;
            }
        }
        if (cthread)
            for (int i = 0; i < cnt; ++i) cacc = add4(cacc, s_rows[i * f4 + threadIdx.x]);
        first = false;
        pc += cnt;
    } while (pc < p1);
    if (cthread) g_c[(long long)b * ldc4 + threadIdx.x] = cacc;
}

// ---------------------------------------------------------------- two stacked Linears as one
// W_eff = W_last W_out, b_eff = W_last b_out + b_last (no activation between the layers).
// Thread (kq, jg) of a workgroup sums j = jg, jg+32, ... for column k = 32*blockIdx.x + kq and
// all classes; the 32 partial sums are added in jg order through LDS.  The bias is the virtual
// column k == K.
template <int MAXC>
__global__ __launch_bounds__(1024) void affine_merge_fwd_kernel(int C, int J, int K, const float *__restrict__ w_last,
                                                               const float *__restrict__ b_last,
                                                               const float *__restrict__ w_out,
                                                               const float *__restrict__ b_out,
                                                               float *__restrict__ w_eff, float *__restrict__ b_eff) {
    __shared__ float sh[32][MAXC][32];
    const int kq = threadIdx.x & 31, jg = threadIdx.x >> 5;
    const int k = blockIdx.x * 32 + kq;
    float acc[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[c] = 0.f;
    // (eight rows are requested before the first is used: a thread's ~20 rows are a chain of memory round trips
    // otherwise; the terms are added in the same order either way)
    constexpr int U = 8;
    if (k <= K)
        for (int j0 = jg; j0 < J; j0 += 32 * U) {
            float w[U], wl[U][MAXC];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int j = j0 + 32 * u, jc = j < J ? j : jg;
                w[u] = k < K ? w_out[(long long)jc * K + k] : (b_out ? b_out[jc] : 0.f);
#pragma unroll
                for (int c = 0; c < MAXC; ++c) wl[u][c] = c < C ? w_last[c * J + jc] : 0.f;
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (j0 + 32 * u < J) {
#pragma unroll
                    for (int c = 0; c < MAXC; ++c)
                        if (c < C) acc[c] += wl[u][c] * w[u];
                }
        }
#pragma unroll
    for (int c = 0; c < MAXC; ++c) sh[jg][c][kq] = acc[c];
    __syncthreads();
    if (jg == 0 && k <= K) {
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                float t = sh[0][c][kq];
#pragma unroll
                for (int q = 1; q < 32; ++q) t += sh[q][c][kq];
                if (k < K)
                    w_eff[c * K + k] = t;
                else
                    b_eff[c] = t + (b_last ? b_last[c] : 0.f);
            }
    }
}

// Backward: one wave per row j of W_out.
//   g_w_out[j][k] = sum_c w_last[c][j] g_w_eff[c][k]        g_b_out[j] = sum_c w_last[c][j] g_b_eff[c]
//   g_w_last[c][j] = sum_k g_w_eff[c][k] w_out[j][k] + g_b_eff[c] b_out[j]
template <int MAXC>
__global__ __launch_bounds__(256) void affine_merge_bwd_kernel(int C, int J, int K, const float *__restrict__ w_last,
                                                               const float *__restrict__ w_out,
                                                               const float *__restrict__ b_out,
                                                               const float *__restrict__ g_w_eff,
                                                               const float *__restrict__ g_b_eff,
                                                               float *__restrict__ g_w_last,
                                                               float *__restrict__ g_w_out,
                                                               float *__restrict__ g_b_out) {
    const int lane = threadIdx.x & 63;
    const int j = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (j >= J) return;
    float wl[MAXC], dot[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) wl[c] = c < C ? w_last[c * J + j] : 0.f, dot[c] = 0.f;
    for (int k = lane; k < K; k += 64) {
        const float w = w_out[(long long)j * K + k];
        float o = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float ge = g_w_eff[c * K + k];
                o += wl[c] * ge;
                dot[c] += ge * w;
            }
        g_w_out[(long long)j * K + k] = o;
    }
#pragma unroll
    for (int c = 0; c < MAXC; ++c)
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) dot[c] += __shfl_xor(dot[c], d);
    if (lane == 0) {
        float gb = 0.f;
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                g_w_last[c * J + j] = dot[c] + (b_out ? g_b_eff[c] * b_out[j] : 0.f);
                gb += wl[c] * g_b_eff[c];
            }
        if (g_b_out) g_b_out[j] = gb;
    }
}

// ---------------------------------------------------------------- link loss
// One wave per candidate pair: dot products of the pair's hidden row with the C rows of W
// (staged in LDS when they fit), then softmax / loss / argmax on lane 0.  VEC: F and ldhid
// are multiples of 4 and the rows are 16-byte aligned, so lanes read float4.
template <int MAXC, bool VEC, bool LDSW>
__global__ __launch_bounds__(256) void link_ce_fwd_kernel(int P, int F, int C, const float *__restrict__ hid,
                                                          int ldhid, const float *__restrict__ W,
                                                          const float *__restrict__ bias,
                                                          const int32_t *__restrict__ golden,
                                                          float *__restrict__ logits, float *__restrict__ dlogits,
                                                          float *__restrict__ part_loss, int *__restrict__ part_ok,
                                                          const int32_t *__restrict__ live) {
    extern __shared__ __align__(16) float sW[];
    __shared__ float sh_loss[4];
    __shared__ int sh_ok[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (LDSW) {
        for (int i = threadIdx.x; i < C * F; i += 256) sW[i] = W[i];
        __syncthreads();
    }
    float loss_acc = 0.f;
    int ok_acc = 0;
    // rows from *live on (when given) are padding: zero logits and gradient, no part in the mean
    const int P_live = live ? min(P, max(*live, 0)) : P;
    for (int p = blockIdx.x * 4 + wave; p < P; p += gridDim.x * 4) {
        if (p >= P_live) {   // wave-uniform
            if (lane < C) logits[(long long)p * C + lane] = 0.f, dlogits[(long long)p * C + lane] = 0.f;
            continue;
        }
        float dot[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) dot[c] = 0.f;
        const float *row = hid + (long long)p * ldhid;
        if (VEC) {
            const float4 *row4 = reinterpret_cast<const float4 *>(row);
            const int F4 = F >> 2;
            for (int f = lane; f < F4; f += 64) {
                const float4 hv = row4[f];
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) {
                        const float4 wv = LDSW ? reinterpret_cast<const float4 *>(sW)[c * F4 + f]
                                               : reinterpret_cast<const float4 *>(W)[c * F4 + f];
                        dot[c] += (hv.x * wv.x + hv.y * wv.y) + (hv.z * wv.z + hv.w * wv.w);
                    }
            }
        } else {
            for (int f = lane; f < F; f += 64) {
                const float hv = row[f];
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) dot[c] += hv * (LDSW ? sW[c * F + f] : W[c * F + f]);
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) dot[c] += __shfl_xor(dot[c], d);
        if (lane == 0) {
            float mx = -INFINITY;
            int arg = 0;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    dot[c] += bias ? bias[c] : 0.f;
                    if (dot[c] > mx) mx = dot[c], arg = c;
                }
            float se = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) se += expf(dot[c] - mx);
            const float lse = mx + logf(se);
            const int gold = golden[p];
            float lg = 0.f;
#pragma unroll
            for (int c = 0; c < MAXC; ++c)
                if (c < C) {
                    if (c == gold) lg = dot[c];
                    logits[(long long)p * C + c] = dot[c];
                    dlogits[(long long)p * C + c] = (expf(dot[c] - lse) - (c == gold ? 1.f : 0.f)) / (float)P_live;
                }
            loss_acc += lse - lg;
            ok_acc += (arg == gold);
        }
    }
    if (lane == 0) sh_loss[wave] = loss_acc, sh_ok[wave] = ok_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        part_loss[blockIdx.x] = ((sh_loss[0] + sh_loss[1]) + sh_loss[2]) + sh_loss[3];
        part_ok[blockIdx.x] = sh_ok[0] + sh_ok[1] + sh_ok[2] + sh_ok[3];
    }
}

// The same with half a wave per pair (rows of at most 32 * ITER float4): the whole row is requested at once, two
// rows are in flight per wave, and the lane sums and the softmax tail run for both at the same time.
template <int MAXC, int ITER>
__global__ __launch_bounds__(256) void link_ce_fwd_half_kernel(int P, int F, int C, const float *__restrict__ hid,
                                                               int ldhid, const float *__restrict__ W,
                                                               const float *__restrict__ bias,
                                                               const int32_t *__restrict__ golden,
                                                               float *__restrict__ logits, float *__restrict__ dlogits,
                                                               float *__restrict__ part_loss, int *__restrict__ part_ok,
                                                               const int32_t *__restrict__ live) {
    extern __shared__ __align__(16) float sW[];
    __shared__ float sh_loss[8];
    __shared__ int sh_ok[8];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, half = lane >> 5, l = lane & 31;
    for (int i = threadIdx.x; i < C * F; i += 256) sW[i] = W[i];
    __syncthreads();
    float loss_acc = 0.f;
    int ok_acc = 0;
    const int P_live = live ? min(P, max(*live, 0)) : P;
    const int F4 = F >> 2;
    for (int p0 = (blockIdx.x * 4 + wave) * 2; p0 < P; p0 += gridDim.x * 8) {
        const int p = p0 + half;
        const bool in = p < P, live_row = p < P_live;
        const float4 *row4 = reinterpret_cast<const float4 *>(hid + (long long)(live_row ? p : 0) * ldhid);
        float4 hv[ITER];
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int f = l + 32 * it;
            hv[it] = row4[f < F4 ? f : 0];
        }
        float dot[MAXC];
#pragma unroll
        for (int c = 0; c < MAXC; ++c) dot[c] = 0.f;
#pragma unroll
        for (int it = 0; it < ITER; ++it) {
            const int f = l + 32 * it;
            if (f < F4 && live_row) {
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) {
                        const float4 wv = reinterpret_cast<const float4 *>(sW)[c * F4 + f];
                        dot[c] += (hv[it].x * wv.x + hv[it].y * wv.y) + (hv[it].z * wv.z + hv[it].w * wv.w);
                    }
            }
        }
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
#pragma unroll
            for (int d = 16; d > 0; d >>= 1) dot[c] += __shfl_xor(dot[c], d);   // (stays inside the half)
        if (l == 0 && in) {
            if (!live_row) {   // padding behind the live count: zero logits and gradient, no part in the mean
                for (int c = 0; c < C; ++c) logits[(long long)p * C + c] = 0.f, dlogits[(long long)p * C + c] = 0.f;
            } else {
                float mx = -INFINITY;
                int arg = 0;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) {
                        dot[c] += bias ? bias[c] : 0.f;
                        if (dot[c] > mx) mx = dot[c], arg = c;
                    }
                float se = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) se += expf(dot[c] - mx);
                const float lse = mx + logf(se);
                const int gold = golden[p];
                float lg = 0.f;
#pragma unroll
                for (int c = 0; c < MAXC; ++c)
                    if (c < C) {
                        if (c == gold) lg = dot[c];
                        logits[(long long)p * C + c] = dot[c];
                        dlogits[(long long)p * C + c] = (expf(dot[c] - lse) - (c == gold ? 1.f : 0.f)) / (float)P_live;
                    }
                loss_acc += lse - lg;
                ok_acc += (arg == gold);
            }
        }
    }
    if (l == 0) sh_loss[wave * 2 + half] = loss_acc, sh_ok[wave * 2 + half] = ok_acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = sh_loss[0];
        int k = sh_ok[0];
        for (int i = 1; i < 8; ++i) t += sh_loss[i], k += sh_ok[i];
        part_loss[blockIdx.x] = t;
        part_ok[blockIdx.x] = k;
    }
}

__global__ void link_ce_finish_kernel(int P, int n_part, const float *__restrict__ part_loss,
                                      const int *__restrict__ part_ok, float *__restrict__ loss,
                                      int32_t *__restrict__ n_correct, const int32_t *__restrict__ live) {
    __shared__ float sh_s[256];
    __shared__ int sh_ok[256];
    float s = 0.f;
    int ok = 0;
    for (int i = threadIdx.x; i < n_part; i += 256) s += part_loss[i], ok += part_ok[i];
    sh_s[threadIdx.x] = s, sh_ok[threadIdx.x] = ok;
    __syncthreads();
    for (int d = 128; d > 0; d >>= 1) {   // fixed order: the same sum on every run
        if (threadIdx.x < d) sh_s[threadIdx.x] += sh_s[threadIdx.x + d], sh_ok[threadIdx.x] += sh_ok[threadIdx.x + d];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        const int P_live = live ? min(P, max(*live, 0)) : P;
        *loss = sh_s[0] / (float)P_live;   // 0/0 = NaN for no rows, as torch's mean over no rows
        if (n_correct) *n_correct = sh_ok[0];
    }
}

__global__ void link_ce_bwd_input_kernel(int P, int F, int C, const float *__restrict__ dlogits,
                                         const float *__restrict__ W, const float *__restrict__ g_loss,
                                         float *__restrict__ g_hid, int ldg) {
    const float gl = *g_loss;
    const long long total = (long long)P * F;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int p = (int)(i / F), f = (int)(i - (long long)p * F);
        float s = 0.f;
        for (int c = 0; c < C; ++c) s += dlogits[(long long)p * C + c] * W[c * F + f];
        g_hid[(long long)p * ldg + f] = gl * s;
    }
}

// float4 columns: thread = (pair, 4 columns), consecutive threads along the row (16-byte stores, 32-bit index math)
template <int MAXC>
__global__ __launch_bounds__(256) void link_ce_bwd_input_vec_kernel(int P, int F4, int C, const float *__restrict__ dlogits,
                                                                    const float4 *__restrict__ W4, const float *__restrict__ g_loss,
                                                                    float4 *__restrict__ g_hid4, int ldg4) {
    const float gl = *g_loss;
    const unsigned total = (unsigned)P * (unsigned)F4;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned p = i / (unsigned)F4, f = i - p * (unsigned)F4;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int c = 0; c < MAXC; ++c)
            if (c < C) {
                const float d = dlogits[(long long)p * C + c];
                const float4 w = W4[c * F4 + f];
                s.x += d * w.x, s.y += d * w.y, s.z += d * w.z, s.w += d * w.w;
            }
        g_hid4[(long long)p * ldg4 + f] = make_float4(gl * s.x, gl * s.y, gl * s.z, gl * s.w);
    }
}

__global__ void link_infer_kernel(int P, int C, const float *__restrict__ logits, float th, int none_bond,
                                  float *__restrict__ conf, int32_t *__restrict__ etype, uint8_t *__restrict__ keep) {
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < P; p += gridDim.x * blockDim.x) {
        const float *l = logits + (long long)p * C;
        float mx = l[0];
        int arg = 0;
        for (int c = 1; c < C; ++c)
            if (l[c] > mx) mx = l[c], arg = c;
        float se = 0.f;
        for (int c = 0; c < C; ++c) se += expf(l[c] - mx);
        const float cf = 1.0f / se;   // softmax value of the maximum
        conf[p] = cf;
        etype[p] = arg;
        keep[p] = (cf >= th && arg != none_bond) ? 1 : 0;
    }
}

// ---------------------------------------------------------------- column sums
__global__ void colsum_partial_kernel(int M, int N, int rows_per_block, const float *__restrict__ x,
                                      const float *__restrict__ mask, long long ldx, float *__restrict__ partial) {
    const int r0 = blockIdx.x * rows_per_block, r1 = min(M, r0 + rows_per_block);
    for (int n = threadIdx.x; n < N; n += blockDim.x) {
        float s = 0.f;
        if (mask)
            for (int r = r0; r < r1; ++r) s += mask[(long long)r * ldx + n] > 0.f ? x[(long long)r * ldx + n] : 0.f;
        else
            for (int r = r0; r < r1; ++r) s += x[(long long)r * ldx + n];
        partial[(long long)blockIdx.x * N + n] = s;
    }
}

// 64 columns per workgroup; the four waves each add a quarter of the partial rows (in row order,
// eight loads in flight) and wave 0 adds the four sums in wave order.
__global__ __launch_bounds__(256) void colsum_finish_kernel(int N, int n_part, const float *__restrict__ partial,
                                                            float *__restrict__ out, int accumulate) {
    __shared__ float sh[3][64];
    const int q = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int n = blockIdx.x * 64 + l;
    const int per = (n_part + 3) >> 2, b0 = q * per, b1 = min(n_part, b0 + per);
    float s = 0.f;
    if (n < N) {
#pragma unroll 8
        for (int b = b0; b < b1; ++b) s += partial[(long long)b * N + n];
    }
    if (q) sh[q - 1][l] = s;
    __syncthreads();
    if (q || n >= N) return;
    s = ((s + sh[0][l]) + sh[1][l]) + sh[2][l];
    out[n] = accumulate ? out[n] + s : s;
}

constexpr int COLSUM_MAX_BLOCKS = 512;
int colsum_rows_per_block(int M) {
    int r = psvae_cdiv(M, COLSUM_MAX_BLOCKS);
    return r < 32 ? 32 : r;
}

constexpr int LINK_CE_MAX_BLOCKS = 2048;
int link_ce_blocks(int P) {
    int b = psvae_cdiv(P, 8);   // (two pairs per wave at a time)
    return b < 1 ? 1 : (b > LINK_CE_MAX_BLOCKS ? LINK_CE_MAX_BLOCKS : b);
}

}  // namespace

extern "C" int psvae_embed_atom_fwd(int V, const int32_t *x, const int32_t *xp, const int32_t *xq,
                                    const float *e_atom, int n_atom, int d_atom, const float *e_piece, int n_piece,
                                    int d_piece, const float *e_pos, int n_pos, int d_pos, float *out, void *stream) {
    PSVAE_REQUIRE(V >= 0 && d_atom >= 0 && d_piece >= 0 && d_pos >= 0, PSVAE_EINVAL, "psvae_embed_atom_fwd: bad sizes");
    PSVAE_REQUIRE(V == 0 || (n_atom > 0 && n_piece > 0 && n_pos > 0), PSVAE_EINVAL, "psvae_embed_atom_fwd: empty table");
    const long long total = (long long)V * (d_atom + d_piece + d_pos);
    if (total == 0) return PSVAE_OK;
    PSVAE_REQUIRE(x && xp && xq && e_atom && e_piece && e_pos && out, PSVAE_EINVAL, "psvae_embed_atom_fwd: null pointer");
    hipLaunchKernelGGL(embed_atom_fwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, V, x, xp,
                       xq, e_atom, n_atom, d_atom, e_piece, n_piece, d_piece, e_pos, n_pos, d_pos, out);
    PSVAE_CHECK_LAUNCH("psvae_embed_atom_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_embed_atom_bwd(int V, const int32_t *x, const int32_t *xp, const int32_t *xq, const float *g_out,
                                    int n_atom, int d_atom, int n_piece, int d_piece, int n_pos, int d_pos,
                                    float *g_atom, float *g_piece, float *g_pos, void *stream) {
    PSVAE_REQUIRE(V >= 0 && d_atom >= 0 && d_piece >= 0 && d_pos >= 0, PSVAE_EINVAL, "psvae_embed_atom_bwd: bad sizes");
    PSVAE_REQUIRE(V == 0 || (n_atom > 0 && n_piece > 0 && n_pos > 0), PSVAE_EINVAL, "psvae_embed_atom_bwd: empty table");
    const long long total = (long long)V * (d_atom + d_piece + d_pos);
    if (total == 0) return PSVAE_OK;
    PSVAE_REQUIRE(x && xp && xq && g_out && g_atom && g_piece && g_pos, PSVAE_EINVAL, "psvae_embed_atom_bwd: null pointer");
    hipLaunchKernelGGL(embed_atom_bwd_kernel, dim3(grid_for(total, 256)), dim3(256), 0, (hipStream_t)stream, V, x, xp,
                       xq, g_out, n_atom, d_atom, n_piece, d_piece, n_pos, d_pos, g_atom, g_piece, g_pos);
    PSVAE_CHECK_LAUNCH("psvae_embed_atom_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_embed_atom_onehot(int V, const int32_t *x, const int32_t *xp, const int32_t *xq, int n_atom,
                                       int n_piece, int n_pos, float *out, int ld, void *stream) {
    PSVAE_REQUIRE(V >= 0 && n_atom >= 0 && n_piece >= 0 && n_pos >= 0 && (V == 0 || (n_atom > 0 && n_piece > 0 && n_pos > 0)),
                  PSVAE_EINVAL, "psvae_embed_atom_onehot: bad sizes");
    PSVAE_REQUIRE(ld % 4 == 0 && ld >= n_atom + n_piece + n_pos && ld > 0, PSVAE_EINVAL,
                  "psvae_embed_atom_onehot: ld=%d must be a multiple of 4 covering %d ids", ld,
                  n_atom + n_piece + n_pos);
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(x && xp && xq && out && psvae_aligned16(out), PSVAE_EINVAL,
                  "psvae_embed_atom_onehot: null or misaligned pointer");
    hipLaunchKernelGGL(embed_atom_onehot_kernel, dim3(grid_for((long long)V * (ld / 4), 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, V, x, xp, xq, n_atom, n_piece, n_pos, (float4 *)out, ld / 4);
    PSVAE_CHECK_LAUNCH("psvae_embed_atom_onehot");
    return PSVAE_OK;
}

extern "C" int psvae_embedding_fwd(int V, int D, int n_rows, const int32_t *ids, const float *table, float *out,
                                   void *stream) {
    const char *fn = "psvae_embedding_fwd";
    PSVAE_REQUIRE(V >= 0 && D > 0 && D % 4 == 0 && n_rows > 0, PSVAE_EINVAL, "%s: bad sizes V=%d D=%d rows=%d", fn, V, D,
                  n_rows);
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(ids && table && out && psvae_aligned16(table) && psvae_aligned16(out), PSVAE_EINVAL,
                  "%s: null or misaligned pointer", fn);
    hipLaunchKernelGGL(embedding_fwd_kernel, dim3(grid_for((long long)V * (D / 4), 256, 8192)), dim3(256), 0,
                       (hipStream_t)stream, V, D / 4, n_rows, ids, (const float4 *)table, (float4 *)out);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_onehot_f32(int V, const int32_t *ids, float *out, int ld, void *stream) {
    const char *fn = "psvae_onehot_f32";
    PSVAE_REQUIRE(V >= 0 && ld > 0 && ld % 4 == 0, PSVAE_EINVAL, "%s: ld=%d must be a positive multiple of 4", fn, ld);
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(ids && out && psvae_aligned16(out), PSVAE_EINVAL, "%s: null or misaligned pointer", fn);
    hipLaunchKernelGGL(onehot_kernel, dim3(grid_for((long long)V * (ld / 4), 256, 16384)), dim3(256), 0,
                       (hipStream_t)stream, V, ids, (float4 *)out, ld / 4);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

static int check_seg(const char *fn, int B, int W, const void *a, int lda, const void *b, int ldb, const void *s0,
                     const void *s1) {
    PSVAE_REQUIRE(B >= 0 && W > 0 && W % 4 == 0, PSVAE_EINVAL, "%s: W=%d must be a positive multiple of 4", fn, W);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(a && b && s0 && s1 && psvae_aligned16(a) && psvae_aligned16(b), PSVAE_EINVAL,
                  "%s: null or misaligned operand", fn);
    PSVAE_REQUIRE(lda % 4 == 0 && lda >= W && ldb % 4 == 0 && ldb >= W, PSVAE_EINVAL, "%s: bad row stride", fn);
    return PSVAE_OK;
}

extern "C" int psvae_segment_sum_fwd(int B, int W, const float *x, int ldx, const int32_t *seg_begin,
                                     const int32_t *seg_end, float *out, int ldo, void *stream) {
    int rc = check_seg("psvae_segment_sum_fwd", B, W, x, ldx, out, ldo, seg_begin, seg_end);
    if (rc || B == 0) return rc;
    // (every column in one pass where 1024 threads cover the row: 300 float4 columns on 320 threads, not 256 + 44)
    const int cols = W / 4, threads = cols >= 1024 ? 1024 : psvae_cdiv(cols, 64) * 64;
    hipLaunchKernelGGL(segment_sum_fwd_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, cols,
                       (const float4 *)x, ldx / 4, seg_begin, seg_end, (float4 *)out, ldo / 4);
    PSVAE_CHECK_LAUNCH("psvae_segment_sum_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_segment_sum_bwd(int B, int W, const float *g_out, int ldgo, const int32_t *seg_begin,
                                     const int32_t *seg_end, float *g_x, int ldgx, void *stream) {
    int rc = check_seg("psvae_segment_sum_bwd", B, W, g_out, ldgo, g_x, ldgx, seg_begin, seg_end);
    if (rc || B == 0) return rc;
    const int cols = W / 4, threads = cols >= 256 ? 256 : psvae_cdiv(cols, 64) * 64;
    hipLaunchKernelGGL(segment_sum_bwd_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, cols,
                       (const float4 *)g_out, ldgo / 4, seg_begin, seg_end, (float4 *)g_x, ldgx / 4);
    PSVAE_CHECK_LAUNCH("psvae_segment_sum_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_graph_rows_gated(int V, int N, const float *q, int ldq, const int32_t *node_graph,
                                      const uint64_t *gate_bits, float *out, int ldo, void *stream) {
    const char *fn = "psvae_graph_rows_gated";
    PSVAE_REQUIRE(V >= 0 && N > 0 && N % 4 == 0 && ldq % 4 == 0 && ldq >= N && ldo % 4 == 0 && ldo >= N, PSVAE_EINVAL,
                  "%s: bad sizes V=%d N=%d ldq=%d ldo=%d", fn, V, N, ldq, ldo);
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(q && node_graph && gate_bits && out && psvae_aligned16(q) && psvae_aligned16(out), PSVAE_EINVAL,
                  "%s: null or misaligned operand", fn);
    hipLaunchKernelGGL(graph_rows_gated_kernel, dim3(psvae_cdiv(V, 32)), dim3(256), 0, (hipStream_t)stream, V, N, q, ldq,
                       node_graph, (const unsigned long long *)gate_bits, psvae_cdiv(N, 32), out, ldo);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" size_t psvae_rsample_kl_workspace_bytes(int B, int L) {
    if (B <= 0 || L <= 0) return 0;
    const int nb = rs_blocks((long long)B * L);
    return nb <= 1 ? 0 : (size_t)nb * sizeof(float);
}

extern "C" int psvae_rsample_kl_fwd(int B, int L, const float *mean, const float *lv_pre, const float *eps, float *z,
                                    float *kl, void *workspace, size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(B > 0 && L > 0 && (long long)B * L < (1ll << 31), PSVAE_EINVAL, "psvae_rsample_kl_fwd: bad sizes");
    PSVAE_REQUIRE(mean && lv_pre && eps && z && kl, PSVAE_EINVAL, "psvae_rsample_kl_fwd: null pointer");
    const int nb = rs_blocks((long long)B * L);
    PSVAE_REQUIRE(nb == 1 || (workspace != nullptr && workspace_bytes >= (size_t)nb * sizeof(float)), PSVAE_EWORKSPACE,
                  "psvae_rsample_kl_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    hipLaunchKernelGGL(rsample_kl_fwd_kernel, dim3(nb), dim3(RS_THREADS), 0, s, B, L, mean, lv_pre, eps, z, (float *)workspace,
                       kl);
    PSVAE_CHECK_LAUNCH("psvae_rsample_kl_fwd");
    if (nb > 1) {
        hipLaunchKernelGGL(rsample_kl_finish_kernel, dim3(1), dim3(RS_THREADS), 0, s, B, nb, (const float *)workspace, kl);
        PSVAE_CHECK_LAUNCH("psvae_rsample_kl_fwd (finish)");
    }
    return PSVAE_OK;
}

extern "C" int psvae_rsample_kl_bwd(int B, int L, const float *mean, const float *lv_pre, const float *eps,
                                    const float *g_z, const float *g_kl, float *g_mean, float *g_lv_pre,
                                    void *stream) {
    PSVAE_REQUIRE(B > 0 && L > 0 && (long long)B * L < (1ll << 31), PSVAE_EINVAL, "psvae_rsample_kl_bwd: bad sizes");
    PSVAE_REQUIRE(mean && lv_pre && eps && g_mean && g_lv_pre, PSVAE_EINVAL, "psvae_rsample_kl_bwd: null pointer");
    hipLaunchKernelGGL(rsample_kl_bwd_kernel, dim3(grid_for((long long)B * L, 256)), dim3(256), 0, (hipStream_t)stream,
                       B, L, mean, lv_pre, eps, g_z, g_kl, g_mean, g_lv_pre);
    PSVAE_CHECK_LAUNCH("psvae_rsample_kl_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_gru_cell_fwd(int B, int H, const float *gi, const float *gh, const float *h_prev, float *h,
                                  float *gates, void *stream) {
    PSVAE_REQUIRE(B >= 0 && H > 0 && (long long)B * H < (1ll << 31), PSVAE_EINVAL, "psvae_gru_cell_fwd: bad sizes");
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(gi && gh && h_prev && h && gates, PSVAE_EINVAL, "psvae_gru_cell_fwd: null pointer");
    hipLaunchKernelGGL(gru_cell_fwd_kernel, dim3(grid_for((long long)B * H, 256)), dim3(256), 0, (hipStream_t)stream, B,
                       H, gi, gh, h_prev, h, gates);
    PSVAE_CHECK_LAUNCH("psvae_gru_cell_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_gru_cell_bwd(int B, int H, const float *gates, const float *gh, const float *h_prev,
                                  const float *dh, const float *dh_extra, float *dgi, float *dgh, float *dh_prev,
                                  void *stream) {
    PSVAE_REQUIRE(B >= 0 && H > 0 && (long long)B * H < (1ll << 31), PSVAE_EINVAL, "psvae_gru_cell_bwd: bad sizes");
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(gates && gh && h_prev && dh && dgi && dgh && dh_prev, PSVAE_EINVAL, "psvae_gru_cell_bwd: null pointer");
    hipLaunchKernelGGL(gru_cell_bwd_kernel, dim3(grid_for((long long)B * H, 256)), dim3(256), 0, (hipStream_t)stream, B,
                       H, gates, gh, h_prev, dh, dh_extra, dgi, dgh, dh_prev);
    PSVAE_CHECK_LAUNCH("psvae_gru_cell_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_pair_gather_fwd(int P, int H, int L, const float *h, int ldh, const float *z, int ldz,
                                     const int32_t *src, const int32_t *dst, const int32_t *graph, float *feats,
                                     void *stream) {
    PSVAE_REQUIRE(P >= 0 && H > 0 && L > 0 && H % 4 == 0 && L % 4 == 0, PSVAE_EINVAL,
                  "psvae_pair_gather_fwd: H=%d and L=%d must be positive multiples of 4", H, L);
    if (P == 0) return PSVAE_OK;
    PSVAE_REQUIRE(h && z && src && dst && graph && feats && psvae_aligned16(h) && psvae_aligned16(z) &&
                      psvae_aligned16(feats), PSVAE_EINVAL, "psvae_pair_gather_fwd: null or misaligned operand");
    PSVAE_REQUIRE(ldh % 4 == 0 && ldh >= H && ldz % 4 == 0 && ldz >= L, PSVAE_EINVAL, "psvae_pair_gather_fwd: bad stride");
    const long long total = (long long)P * (2 * H + L) / 4;
    hipLaunchKernelGGL(pair_gather_fwd_kernel, dim3(grid_for(total, 256, 8192)), dim3(256), 0, (hipStream_t)stream, P,
                       H / 4, L / 4, (const float4 *)h, ldh / 4, (const float4 *)z, ldz / 4, src, dst, graph,
                       (float4 *)feats);
    PSVAE_CHECK_LAUNCH("psvae_pair_gather_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_pair_combine_fwd(int P, int F, const float *a, int lda, const float *b, int ldb, const float *c,
                                      int ldc, const int32_t *src, const int32_t *dst, const int32_t *graph,
                                      int act, float *out, int ldo, void *stream) {
    const char *fn = "psvae_pair_combine_fwd";
    PSVAE_REQUIRE(P >= 0 && F > 0 && F % 4 == 0, PSVAE_EINVAL, "%s: F=%d must be a positive multiple of 4", fn, F);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "%s: bad act %d", fn, act);
    if (P == 0) return PSVAE_OK;
    PSVAE_REQUIRE(a && b && c && src && dst && graph && out, PSVAE_EINVAL, "%s: null operand", fn);
    PSVAE_REQUIRE(psvae_aligned16(a) && psvae_aligned16(b) && psvae_aligned16(c) && psvae_aligned16(out), PSVAE_EINVAL,
                  "%s: operands must be 16-byte aligned", fn);
    PSVAE_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ldo % 4 == 0 && lda >= F && ldb >= F && ldc >= F &&
                      ldo >= F, PSVAE_EINVAL, "%s: bad row stride", fn);
    if (F / 4 <= 256) {
        const int ppb = 256 / (F / 4);
        hipLaunchKernelGGL(pair_combine_fwd_rows_kernel, dim3(grid_for(psvae_cdiv(P, 4), ppb, 16384)), dim3(256), 0,
                           (hipStream_t)stream, P, F / 4, (const float4 *)a, lda / 4, (const float4 *)b, ldb / 4,
                           (const float4 *)c, ldc / 4, src, dst, graph, act, (float4 *)out, ldo / 4);
    } else {
        hipLaunchKernelGGL(pair_combine_fwd_kernel, dim3(grid_for((long long)P * (F / 4), 256, 16384)), dim3(256), 0,
                           (hipStream_t)stream, P, F / 4, (const float4 *)a, lda / 4, (const float4 *)b, ldb / 4,
                           (const float4 *)c, ldc / 4, src, dst, graph, act, (float4 *)out, ldo / 4);
    }
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_pair_combine_fwd_graphs(int B, int F, int max_nodes, const float *a, int lda, const float *b, int ldb,
                                             const float *c, int ldc, const int32_t *src, const int32_t *dst,
                                             const int32_t *pair_ptr, const int32_t *node_begin, const int32_t *node_end,
                                             int act, float *out, int ldo, void *stream) {
    const char *fn = "psvae_pair_combine_fwd_graphs";
    PSVAE_REQUIRE(B >= 0 && F > 0 && F % 4 == 0 && F <= 1024 && max_nodes >= 1, PSVAE_EINVAL,
                  "%s: F=%d must be a multiple of 4 up to 1024, max_nodes=%d at least 1", fn, F, max_nodes);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "%s: bad act %d", fn, act);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(a && b && c && src && dst && pair_ptr && node_begin && node_end && out, PSVAE_EINVAL, "%s: null operand", fn);
    PSVAE_REQUIRE(psvae_aligned16(a) && psvae_aligned16(b) && psvae_aligned16(c) && psvae_aligned16(out), PSVAE_EINVAL,
                  "%s: operands must be 16-byte aligned", fn);
    PSVAE_REQUIRE(lda % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && ldo % 4 == 0 && lda >= F && ldb >= F && ldc >= F &&
                      ldo >= F, PSVAE_EINVAL, "%s: bad row stride", fn);
    // column block: both row sets of max_nodes nodes in at most 64 KB of LDS (two workgroups per CU); a graph with
    // more nodes than the staging area holds is served from memory by the kernel itself
    const int f4 = F / 4;
    const int nodes = max_nodes > 2048 ? 2048 : max_nodes;
    int cp4 = (64 * 1024) / (nodes * 2 * 16);
    if (cp4 > f4) cp4 = f4;
    if (cp4 < 1) cp4 = 1;
    const size_t lds = (size_t)nodes * cp4 * 2 * 16;
    hipLaunchKernelGGL(pair_combine_fwd_graph_kernel, dim3(B), dim3(256), lds, (hipStream_t)stream, f4, cp4, nodes,
                       (const float4 *)a, lda / 4, (const float4 *)b, ldb / 4, (const float4 *)c, ldc / 4, src, dst, pair_ptr,
                       node_begin, node_end, act, (float4 *)out, ldo / 4);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_pair_combine_bwd(int B, int F, const float *g_out, int ldg, const float *out, int ldo, int act,
                                      const int32_t *src, const int32_t *dst, const int32_t *pair_ptr,
                                      const int32_t *node_begin, const int32_t *node_end, float *g_ab, int ldab,
                                      float *g_c, int ldc, void *stream) {
    const char *fn = "psvae_pair_combine_bwd";
    PSVAE_REQUIRE(B >= 0 && F > 0 && F % 4 == 0, PSVAE_EINVAL, "%s: F=%d must be a positive multiple of 4", fn, F);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || act == PSVAE_ACT_RELU, PSVAE_EINVAL, "%s: bad act %d", fn, act);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(pair_ptr && node_begin && node_end && g_ab && g_c && psvae_aligned16(g_ab) && psvae_aligned16(g_c) &&
                      (g_out == nullptr || psvae_aligned16(g_out)) && (out == nullptr || psvae_aligned16(out)),
                  PSVAE_EINVAL, "%s: null or misaligned operand", fn);
    PSVAE_REQUIRE(ldg % 4 == 0 && ldg >= F && ldo % 4 == 0 && ldo >= F && ldab % 4 == 0 && ldab >= 2 * F && ldc % 4 == 0 &&
                      ldc >= F, PSVAE_EINVAL, "%s: bad row stride", fn);
    PSVAE_REQUIRE(act == PSVAE_ACT_NONE || out != nullptr || g_out == nullptr, PSVAE_EINVAL, "%s: the ReLU gate needs `out`", fn);
    const int f4 = F / 4;
    PSVAE_REQUIRE(f4 <= 1024, PSVAE_ETOOLARGE, "%s: F too wide", fn);
    int slots = f4 >= 256 ? 1 : psvae_cdiv(256, f4);
    int threads = psvae_cdiv(f4 * slots, 64) * 64;
    if (threads > 1024) slots = 1024 / f4, threads = psvae_cdiv(f4 * slots, 64) * 64;
    int chunk = PAIR_ROWS_LDS_BYTES / (F * (int)sizeof(float));
    chunk = chunk < 1 ? 1 : (chunk > PAIR_ROWS_MAX ? PAIR_ROWS_MAX : chunk);
    hipLaunchKernelGGL(pair_combine_bwd_kernel, dim3(B), dim3(threads), (size_t)chunk * F * sizeof(float),
                       (hipStream_t)stream, f4, slots, chunk,
                       act == PSVAE_ACT_RELU ? 1 : 0, (const float4 *)g_out, ldg / 4, (const float4 *)out, ldo / 4, src, dst,
                       pair_ptr, node_begin, node_end, (float4 *)g_ab, ldab / 4, (float4 *)g_c, ldc / 4);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_pair_gather_bwd(int B, int H, int L, const float *g_feats, const int32_t *src, const int32_t *dst,
                                     const int32_t *pair_ptr, const int32_t *node_begin, const int32_t *node_end,
                                     float *g_h, int ldgh, float *g_z, int ldgz, void *stream) {
    PSVAE_REQUIRE(B >= 0 && H > 0 && L > 0 && H % 4 == 0 && L % 4 == 0, PSVAE_EINVAL,
                  "psvae_pair_gather_bwd: H=%d and L=%d must be positive multiples of 4", H, L);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(pair_ptr && node_begin && node_end && g_h && g_z && psvae_aligned16(g_h) && psvae_aligned16(g_z) &&
                      (g_feats == nullptr || psvae_aligned16(g_feats)), PSVAE_EINVAL,
                  "psvae_pair_gather_bwd: null or misaligned operand");
    PSVAE_REQUIRE(ldgh % 4 == 0 && ldgh >= H && ldgz % 4 == 0 && ldgz >= L, PSVAE_EINVAL, "psvae_pair_gather_bwd: bad stride");
    const int h4 = H / 4, l4 = L / 4;
    PSVAE_REQUIRE(h4 <= 1024 && l4 <= 64, PSVAE_ETOOLARGE, "psvae_pair_gather_bwd: H or L too wide");
    int slots = h4 >= 256 ? 1 : psvae_cdiv(256, h4);
    int threads = psvae_cdiv(h4 * slots, 64) * 64;
    if (threads > 1024) slots = 1024 / h4, threads = psvae_cdiv(h4 * slots, 64) * 64;
    hipLaunchKernelGGL(pair_gather_bwd_kernel, dim3(B), dim3(threads), 0, (hipStream_t)stream, h4, l4, slots,
                       (const float4 *)g_feats, src, dst, pair_ptr, node_begin, node_end, (float4 *)g_h, ldgh / 4,
                       (float4 *)g_z, ldgz / 4);
    PSVAE_CHECK_LAUNCH("psvae_pair_gather_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_affine_merge_fwd(int C, int J, int K, const float *w_last, const float *b_last,
                                      const float *w_out, const float *b_out, float *w_eff, float *b_eff,
                                      void *stream) {
    PSVAE_REQUIRE(C > 0 && J > 0 && K > 0, PSVAE_EINVAL, "psvae_affine_merge_fwd: bad sizes");
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "psvae_affine_merge_fwd: at most 8 classes (C=%d)", C);
    PSVAE_REQUIRE(w_last && w_out && w_eff && b_eff, PSVAE_EINVAL, "psvae_affine_merge_fwd: null pointer");
    const dim3 grid(psvae_cdiv(K + 1, 32));
    if (C <= 4)
        hipLaunchKernelGGL(affine_merge_fwd_kernel<4>, grid, dim3(1024), 0, (hipStream_t)stream, C, J, K, w_last, b_last,
                           w_out, b_out, w_eff, b_eff);
    else
        hipLaunchKernelGGL(affine_merge_fwd_kernel<8>, grid, dim3(1024), 0, (hipStream_t)stream, C, J, K, w_last, b_last,
                           w_out, b_out, w_eff, b_eff);
    PSVAE_CHECK_LAUNCH("psvae_affine_merge_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_affine_merge_bwd(int C, int J, int K, const float *w_last, const float *w_out,
                                      const float *b_out, const float *g_w_eff, const float *g_b_eff,
                                      float *g_w_last, float *g_w_out, float *g_b_out, void *stream) {
    PSVAE_REQUIRE(C > 0 && J > 0 && K > 0, PSVAE_EINVAL, "psvae_affine_merge_bwd: bad sizes");
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "psvae_affine_merge_bwd: at most 8 classes (C=%d)", C);
    PSVAE_REQUIRE(w_last && w_out && g_w_eff && g_b_eff && g_w_last && g_w_out, PSVAE_EINVAL,
                  "psvae_affine_merge_bwd: null pointer");
    PSVAE_REQUIRE((b_out == nullptr) == (g_b_out == nullptr), PSVAE_EINVAL,
                  "psvae_affine_merge_bwd: b_out and g_b_out go together");
    const dim3 grid(psvae_cdiv(J, 4));
    if (C <= 4)
        hipLaunchKernelGGL(affine_merge_bwd_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, C, J, K, w_last, w_out,
                           b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out);
    else
        hipLaunchKernelGGL(affine_merge_bwd_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, C, J, K, w_last, w_out,
                           b_out, g_w_eff, g_b_eff, g_w_last, g_w_out, g_b_out);
    PSVAE_CHECK_LAUNCH("psvae_affine_merge_bwd");
    return PSVAE_OK;
}

extern "C" size_t psvae_link_ce_workspace_bytes(int P) {
    return (size_t)link_ce_blocks(P) * (sizeof(float) + sizeof(int));
}

extern "C" int psvae_link_ce_fwd(int P, int F, int C, const float *hid, int ldhid, const float *W, const float *b,
                                 const int32_t *golden, const int32_t *live_count, float *logits, float *dlogits,
                                 float *loss, int32_t *n_correct, void *workspace, size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(P >= 0 && F > 0 && C > 0, PSVAE_EINVAL, "psvae_link_ce_fwd: bad sizes");
    PSVAE_REQUIRE(C <= 8, PSVAE_ETOOLARGE, "psvae_link_ce_fwd: at most 8 classes (C=%d)", C);
    PSVAE_REQUIRE(loss != nullptr && W != nullptr, PSVAE_EINVAL, "psvae_link_ce_fwd: null pointer");
    PSVAE_REQUIRE(workspace && workspace_bytes >= psvae_link_ce_workspace_bytes(P), PSVAE_EWORKSPACE,
                  "psvae_link_ce_fwd: workspace too small");
    PSVAE_REQUIRE(P == 0 || (hid && golden && logits && dlogits && ldhid >= F), PSVAE_EINVAL,
                  "psvae_link_ce_fwd: null pointer or ldhid < F");
    const int blocks = link_ce_blocks(P);
    float *part_loss = (float *)workspace;
    int *part_ok = (int *)(part_loss + blocks);
    hipStream_t s = (hipStream_t)stream;
    const bool vec = (F % 4 == 0) && (ldhid % 4 == 0) && ((uintptr_t)hid % 16 == 0) && ((uintptr_t)W % 16 == 0);
    const size_t w_bytes = (size_t)C * F * sizeof(float);
    const bool ldsw = w_bytes <= 48 * 1024;
    const size_t shm = ldsw ? w_bytes : 0;
#define PSVAE_LINK_CE(MAXC, VEC, LDSW)                                                                            \
    hipLaunchKernelGGL((link_ce_fwd_kernel<MAXC, VEC, LDSW>), dim3(blocks), dim3(256), shm, s, P, F, C, hid, ldhid, \
                       W, b, golden, logits, dlogits, part_loss, part_ok, live_count)
    const int iters = psvae_cdiv(F / 4, 32);
    if (vec && ldsw && iters <= 4) {   // half a wave per pair
#define PSVAE_LINK_CE_HALF(MAXC, ITER)                                                                                  \
    hipLaunchKernelGGL((link_ce_fwd_half_kernel<MAXC, ITER>), dim3(blocks), dim3(256), shm, s, P, F, C, hid, ldhid, W, b, \
                       golden, logits, dlogits, part_loss, part_ok, live_count)
        if (C <= 4) {
            if (iters <= 2) PSVAE_LINK_CE_HALF(4, 2); else if (iters == 3) PSVAE_LINK_CE_HALF(4, 3); else PSVAE_LINK_CE_HALF(4, 4);
        } else {
            if (iters <= 2) PSVAE_LINK_CE_HALF(8, 2); else if (iters == 3) PSVAE_LINK_CE_HALF(8, 3); else PSVAE_LINK_CE_HALF(8, 4);
        }
#undef PSVAE_LINK_CE_HALF
    } else if (C <= 4) {
        if (vec && ldsw) PSVAE_LINK_CE(4, true, true);
        else if (ldsw) PSVAE_LINK_CE(4, false, true);
        else PSVAE_LINK_CE(4, false, false);
    } else {
        if (vec && ldsw) PSVAE_LINK_CE(8, true, true);
        else if (ldsw) PSVAE_LINK_CE(8, false, true);
        else PSVAE_LINK_CE(8, false, false);
    }
#undef PSVAE_LINK_CE
    PSVAE_CHECK_LAUNCH("psvae_link_ce_fwd");
    hipLaunchKernelGGL(link_ce_finish_kernel, dim3(1), dim3(256), 0, s, P, blocks, part_loss, part_ok, loss, n_correct,
                       live_count);
    PSVAE_CHECK_LAUNCH("psvae_link_ce_fwd (finish)");
    return PSVAE_OK;
}

extern "C" int psvae_link_ce_bwd_input(int P, int F, int C, const float *dlogits, const float *W, const float *g_loss,
                                       float *g_hid, int ldg, void *stream) {
    PSVAE_REQUIRE(P >= 0 && F > 0 && C > 0 && ldg >= F, PSVAE_EINVAL, "psvae_link_ce_bwd_input: bad sizes");
    if (P == 0) return PSVAE_OK;
    PSVAE_REQUIRE(dlogits && W && g_loss && g_hid, PSVAE_EINVAL, "psvae_link_ce_bwd_input: null pointer");
    if (F % 4 == 0 && ldg % 4 == 0 && C <= 8 && (uintptr_t)W % 16 == 0 && (uintptr_t)g_hid % 16 == 0 &&
        (long long)P * (F / 4) < (1ll << 32)) {
        const dim3 grid(grid_for((long long)P * (F / 4), 256, 16384));
        if (C <= 4)
            hipLaunchKernelGGL(link_ce_bwd_input_vec_kernel<4>, grid, dim3(256), 0, (hipStream_t)stream, P, F / 4, C, dlogits,
                               (const float4 *)W, g_loss, (float4 *)g_hid, ldg / 4);
        else
            hipLaunchKernelGGL(link_ce_bwd_input_vec_kernel<8>, grid, dim3(256), 0, (hipStream_t)stream, P, F / 4, C, dlogits,
                               (const float4 *)W, g_loss, (float4 *)g_hid, ldg / 4);
    } else {
        hipLaunchKernelGGL(link_ce_bwd_input_kernel, dim3(grid_for((long long)P * F, 256, 8192)), dim3(256), 0,
                           (hipStream_t)stream, P, F, C, dlogits, W, g_loss, g_hid, ldg);
    }
    PSVAE_CHECK_LAUNCH("psvae_link_ce_bwd_input");
    return PSVAE_OK;
}

extern "C" int psvae_link_infer(int P, int C, const float *logits, float add_edge_th, int none_bond, float *conf,
                                int32_t *etype, uint8_t *keep, void *stream) {
    PSVAE_REQUIRE(P >= 0 && C > 0, PSVAE_EINVAL, "psvae_link_infer: bad sizes");
    if (P == 0) return PSVAE_OK;
    PSVAE_REQUIRE(logits && conf && etype && keep, PSVAE_EINVAL, "psvae_link_infer: null pointer");
    hipLaunchKernelGGL(link_infer_kernel, dim3(grid_for(P, 256)), dim3(256), 0, (hipStream_t)stream, P, C, logits,
                       add_edge_th, none_bond, conf, etype, keep);
    PSVAE_CHECK_LAUNCH("psvae_link_infer");
    return PSVAE_OK;
}

extern "C" size_t psvae_colsum_workspace_bytes(int M, int N) {
    if (M <= 0 || N <= 0) return 0;
    return (size_t)psvae_cdiv(M, colsum_rows_per_block(M)) * N * sizeof(float);
}

extern "C" int psvae_colsum_f32(int M, int N, const float *x, const float *mask, int64_t ldx, float *out,
                                int accumulate, void *workspace, size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(M >= 0 && N >= 0, PSVAE_EINVAL, "psvae_colsum_f32: bad sizes");
    if (N == 0) return PSVAE_OK;
    PSVAE_REQUIRE(out != nullptr, PSVAE_EINVAL, "psvae_colsum_f32: out is null");
    hipStream_t s = (hipStream_t)stream;
    if (M == 0) return accumulate ? PSVAE_OK : (int)hipMemsetAsync(out, 0, (size_t)N * sizeof(float), s);
    PSVAE_REQUIRE(x != nullptr && ldx >= N, PSVAE_EINVAL, "psvae_colsum_f32: x null or ldx < N");
    PSVAE_REQUIRE(workspace && workspace_bytes >= psvae_colsum_workspace_bytes(M, N), PSVAE_EWORKSPACE,
                  "psvae_colsum_f32: workspace too small");
    const int rpb = colsum_rows_per_block(M), blocks = psvae_cdiv(M, rpb);
    const int threads = N >= 256 ? 256 : psvae_cdiv(N, 64) * 64;
    hipLaunchKernelGGL(colsum_partial_kernel, dim3(blocks), dim3(threads), 0, s, M, N, rpb, x, mask, (long long)ldx,
                       (float *)workspace);
    PSVAE_CHECK_LAUNCH("psvae_colsum_f32");
    hipLaunchKernelGGL(colsum_finish_kernel, dim3(psvae_cdiv(N, 64)), dim3(256), 0, s, N, blocks,
                       (const float *)workspace, out, accumulate);
    PSVAE_CHECK_LAUNCH("psvae_colsum_f32 (finish)");
    return PSVAE_OK;
}
