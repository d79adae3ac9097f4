// Table gradients of embedding lookups as deterministic keyed row sums.
//
// The backward of table[ids] (src/modules/vae_piece_decoder.py:65-69, 78; nn.Embedding) is
//     g_table[r] = sum of g[v] over the entries v whose id is r.
// PyTorch adds the rows with float atomics (run-to-run different bits).  Here the entries are put
// in id order once per batch by a stable counting sort (psvae_key_sort: integer work only), and
// every table row is then the sum of a contiguous run of gathered rows of g, taken in a fixed
// order (psvae_keyed_row_sum): g is read once per table it feeds, nothing of size
// entries x table-rows is formed, no atomics touch floats.
//
// Up to three id arrays over the same entries (atom type, piece, position of every atom) are
// sorted into ONE permutation: table t's rows follow table t-1's, so one keyed sum over the
// concatenation gives the gradient of the stacked table.
#include "common.h"

namespace {

constexpr int KS_THREADS = 1024, KS_WAVES = KS_THREADS / 64;
constexpr int KS_MAX_ROWS = 8192;    // table rows in all tables together (they sit in LDS as counters)
constexpr int KS_MAX_BLOCKS = 64;    // workgroups of the counting sort: few, so that the per-row walk over them is short

__device__ __forceinline__ int clamp_id(int id, int n) { return min(max(id, 0), n - 1); }

struct KeySet {
    const int32_t *keys[3];
    int n[3];      // rows of each table
    int base[3];   // first row of each table in the stacked table
    int tables;    // 1..3
    int R;         // rows of the stacked table
};

// Exclusive prefix sums over tot[0..R) and over ceil(tot / chunk), written to row_off / unit_off
// ([R+1] each; start_out, if given, also receives row_off in LDS and may alias tot).  One workgroup
// of KS_THREADS threads; sa / sb: KS_THREADS ints of LDS each.
__device__ void block_offsets(const int *tot, int R, int chunk, int32_t *__restrict__ row_off,
                              int32_t *__restrict__ unit_off, int *start_out, int *sa, int *sb) {
    const int t = threadIdx.x;
    const int per = (R + KS_THREADS - 1) / KS_THREADS;
    const int r0 = min(R, t * per), r1 = min(R, r0 + per);
    int a = 0, b = 0;
    for (int r = r0; r < r1; ++r) a += tot[r], b += (tot[r] + chunk - 1) / chunk;
    sa[t] = a, sb[t] = b;
    __syncthreads();
    for (int off = 1; off < KS_THREADS; off <<= 1) {
        int va = 0, vb = 0;
        if (t >= off) va = sa[t - off], vb = sb[t - off];
        __syncthreads();
        sa[t] += va, sb[t] += vb;
        __syncthreads();
    }
    int pa = sa[t] - a, pb = sb[t] - b;
    for (int r = r0; r < r1; ++r) {
        const int c = tot[r];
        row_off[r] = pa, unit_off[r] = pb;
        if (start_out) start_out[r] = pa;
        pa += c, pb += (c + chunk - 1) / chunk;
    }
    if (t == KS_THREADS - 1) row_off[R] = sa[t], unit_off[R] = sb[t];
}

// Entries [begin, end) -> perm, in ascending entry order inside every row: cnt[r] (LDS) holds the
// next free position of row r.  Lanes of a wave that hold the same row form a group (found with
// ballots); the group's lowest lane reserves the group's positions; waves take their turn in wave
// order, so the position of an entry does not depend on timing.
__device__ void scatter_entries(const KeySet &ks, int begin, int end, int *cnt, int32_t *__restrict__ perm) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned long long below = lane == 0 ? 0ull : (~0ull >> (64 - lane));
    for (int c0 = begin; c0 < end; c0 += KS_THREADS) {
        const int i = c0 + threadIdx.x;
        const bool valid = i < end;
        int row[3], rank[3], size[3], lead[3];
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            row[t] = -1, rank[t] = 0, size[t] = 0, lead[t] = 0;
            if (t >= ks.tables) continue;   // uniform
            if (valid) row[t] = ks.base[t] + clamp_id(ks.keys[t][i], ks.n[t]);
            unsigned long long peers = 0;
            bool done = !valid;
            while (true) {
                const unsigned long long todo = __ballot(!done);
                if (todo == 0) break;
                const int r0 = __shfl(row[t], __ffsll((long long)todo) - 1);
                const bool mine = !done && row[t] == r0;
                const unsigned long long m = __ballot(mine);
                if (mine) peers = m, done = true;
            }
            if (valid) rank[t] = __popcll(peers & below), size[t] = __popcll(peers), lead[t] = __ffsll((long long)peers) - 1;
        }
        int first[3] = {0, 0, 0};
        for (int w = 0; w < KS_WAVES; ++w) {
            if (wave == w) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
                    if (t < ks.tables && valid && rank[t] == 0) first[t] = cnt[row[t]], cnt[row[t]] = first[t] + size[t];
            }
            __syncthreads();
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            if (t >= ks.tables) continue;
            const int f = __shfl(first[t], lead[t]);
            if (valid) perm[f + rank[t]] = i;
        }
    }
}

// ---- counting sort, three launches: per-workgroup histograms, offsets, scatter
__global__ __launch_bounds__(KS_THREADS) void key_hist_kernel(KeySet ks, int V, int per_block, int32_t *__restrict__ hist) {
    extern __shared__ int cnt[];
    for (int r = threadIdx.x; r < ks.R; r += KS_THREADS) cnt[r] = 0;
    __syncthreads();
    const int begin = blockIdx.x * per_block, end = min(V, begin + per_block);
    for (int i = begin + threadIdx.x; i < end; i += KS_THREADS)
        for (int t = 0; t < ks.tables; ++t) atomicAdd(&cnt[ks.base[t] + clamp_id(ks.keys[t][i], ks.n[t])], 1);
    __syncthreads();
    for (int r = threadIdx.x; r < ks.R; r += KS_THREADS) hist[(long long)blockIdx.x * ks.R + r] = cnt[r];
}

// hist[b][r] becomes the number of entries of row r in workgroups 0..b-1; row_off / unit_off from the totals
__global__ __launch_bounds__(KS_THREADS) void key_offsets_kernel(int R, int nblk, int chunk, int32_t *__restrict__ hist,
                                                                  int32_t *__restrict__ row_off,
                                                                  int32_t *__restrict__ unit_off) {
    extern __shared__ int lds[];
    int *tot = lds, *sa = lds + R, *sb = sa + KS_THREADS;
    for (int r = threadIdx.x; r < R; r += KS_THREADS) {
        int run = 0;
        for (int b = 0; b < nblk; ++b) {
            const int c = hist[(long long)b * R + r];
            hist[(long long)b * R + r] = run;
            run += c;
        }
        tot[r] = run;
    }
    __syncthreads();
    block_offsets(tot, R, chunk, row_off, unit_off, nullptr, sa, sb);
}

__global__ __launch_bounds__(KS_THREADS) void key_scatter_kernel(KeySet ks, int V, int per_block,
                                                                  const int32_t *__restrict__ hist,
                                                                  const int32_t *__restrict__ row_off,
                                                                  int32_t *__restrict__ perm) {
    extern __shared__ int cnt[];
    for (int r = threadIdx.x; r < ks.R; r += KS_THREADS) cnt[r] = row_off[r] + hist[(long long)blockIdx.x * ks.R + r];
    __syncthreads();
    const int begin = blockIdx.x * per_block, end = min(V, begin + per_block);
    scatter_entries(ks, begin, end, cnt, perm);
}

// ---- the same in one launch by one workgroup (small batches)
__global__ __launch_bounds__(KS_THREADS) void key_sort_single_kernel(KeySet ks, int V, int chunk, int32_t *__restrict__ perm,
                                                                      int32_t *__restrict__ row_off,
                                                                      int32_t *__restrict__ unit_off) {
    extern __shared__ int lds[];
    int *cnt = lds, *sa = lds + ks.R, *sb = sa + KS_THREADS;
    for (int r = threadIdx.x; r < ks.R; r += KS_THREADS) cnt[r] = 0;
    __syncthreads();
    for (int i = threadIdx.x; i < V; i += KS_THREADS)
        for (int t = 0; t < ks.tables; ++t) atomicAdd(&cnt[ks.base[t] + clamp_id(ks.keys[t][i], ks.n[t])], 1);
    __syncthreads();
    block_offsets(cnt, ks.R, chunk, row_off, unit_off, cnt, sa, sb);
    __syncthreads();
    scatter_entries(ks, 0, V, cnt, perm);
}

// ---- keyed row sum
// Unit u = up to `chunk` consecutive entries of one row.  A workgroup finds its row by bisection
// over unit_off, its threads (slot, float4 column) add the gathered rows slot, slot + slots, ...
// (four requests in flight) and the slots are added in slot order.
__global__ void keyed_sum_partial_kernel(int R, int cols, int slots, int chunk, const float4 *__restrict__ g, int ldg4,
                                         const int32_t *__restrict__ perm, const int32_t *__restrict__ row_off,
                                         const int32_t *__restrict__ unit_off, float4 *__restrict__ partial) {
    extern __shared__ float4 sh4[];   // [slots][cols]
    __shared__ int range[2];
    const int u = blockIdx.x;
    if (u >= unit_off[R]) return;     // block-uniform
    if (threadIdx.x == 0) {
        int lo = 0, hi = R;           // the last r with unit_off[r] <= u
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (unit_off[mid] <= u) lo = mid; else hi = mid;
        }
        const int e0 = row_off[lo] + (u - unit_off[lo]) * chunk;
        range[0] = e0, range[1] = min(row_off[lo + 1], e0 + chunk);
    }
    __syncthreads();
    const int e0 = range[0], e1 = range[1];
    const int c = threadIdx.x % cols, slot = threadIdx.x / cols;
    const bool active = slot < slots;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        for (int e = e0 + slot; e < e1; e += 4 * slots) {
            float4 v[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int ee = min(e + k * slots, e1 - 1);
                v[k] = g[(long long)perm[ee] * ldg4 + c];
            }
#pragma unroll
            for (int k = 0; k < 4; ++k)
                if (e + k * slots < e1) acc.x += v[k].x, acc.y += v[k].y, acc.z += v[k].z, acc.w += v[k].w;
        }
        sh4[slot * cols + c] = acc;
    }
    __syncthreads();
    if (active && slot == 0) {
        for (int s = 1; s < slots; ++s) {
            const float4 o = sh4[s * cols + c];
            acc.x += o.x, acc.y += o.y, acc.z += o.z, acc.w += o.w;
        }
        partial[(long long)u * cols + c] = acc;
    }
}

// out[r] = sum of the partials of row r's units: thread group q takes units q, q + groups, ...; the
// groups' sums are added in q order.  A row without entries gets zeros.
__global__ void keyed_sum_finish_kernel(int cols, int groups, const int32_t *__restrict__ unit_off,
                                        const float4 *__restrict__ partial, float4 *__restrict__ out, int ldo4) {
    extern __shared__ float4 sh4[];   // [groups][cols]
    const int r = blockIdx.x;
    const int u0 = unit_off[r], u1 = unit_off[r + 1];
    const int c = threadIdx.x % cols, q = threadIdx.x / cols;
    const bool active = q < groups;
    float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
    if (active) {
        for (int u = u0 + q; u < u1; u += groups) {
            const float4 v = partial[(long long)u * cols + c];
            acc.x += v.x, acc.y += v.y, acc.z += v.z, acc.w += v.w;
        }
        sh4[q * cols + c] = acc;
    }
    __syncthreads();
    if (active && q == 0) {
        for (int s = 1; s < groups; ++s) {
            const float4 o = sh4[s * cols + c];
            acc.x += o.x, acc.y += o.y, acc.z += o.z, acc.w += o.w;
        }
        out[(long long)r * ldo4 + c] = acc;
    }
}

int sort_blocks(int V) {
    const int b = psvae_cdiv(V, KS_THREADS);
    return b < 1 ? 1 : (b > KS_MAX_BLOCKS ? KS_MAX_BLOCKS : b);
}

long long max_units(long long entries, int R, int chunk) { return entries / chunk + R; }

}  // namespace

extern "C" size_t psvae_key_sort_workspace_bytes(int V, int n_rows_total) {
    if (V <= 0 || n_rows_total <= 0) return 0;
    const int nblk = sort_blocks(V);
    return nblk <= 1 ? 0 : (size_t)nblk * (size_t)n_rows_total * sizeof(int32_t);
}

extern "C" int psvae_key_sort(int V, int n_tables, const int32_t *k0, const int32_t *k1, const int32_t *k2, int n0, int n1,
                              int n2, int chunk, int32_t *perm, int32_t *row_off, int32_t *unit_off, void *workspace,
                              size_t workspace_bytes, void *stream) {
    const char *fn = "psvae_key_sort";
    PSVAE_REQUIRE(V >= 0 && n_tables >= 1 && n_tables <= 3 && chunk >= 1, PSVAE_EINVAL, "%s: bad sizes V=%d tables=%d chunk=%d",
                  fn, V, n_tables, chunk);
    KeySet ks;
    const int32_t *keys[3] = {k0, k1, k2};
    const int n[3] = {n0, n1, n2};
    ks.tables = n_tables, ks.R = 0;
    for (int t = 0; t < 3; ++t) {
        ks.keys[t] = t < n_tables ? keys[t] : nullptr;
        ks.n[t] = t < n_tables ? n[t] : 0;
        ks.base[t] = ks.R;
        if (t < n_tables) {
            PSVAE_REQUIRE(n[t] >= 1 && (V == 0 || keys[t] != nullptr), PSVAE_EINVAL, "%s: table %d has %d rows or null ids", fn, t,
                          n[t]);
            ks.R += n[t];
        }
    }
    PSVAE_REQUIRE(ks.R <= KS_MAX_ROWS, PSVAE_ETOOLARGE, "%s: %d table rows > %d", fn, ks.R, KS_MAX_ROWS);
    PSVAE_REQUIRE((long long)V * n_tables < (1ll << 31), PSVAE_ETOOLARGE, "%s: too many entries", fn);
    PSVAE_REQUIRE(row_off != nullptr && unit_off != nullptr && (V == 0 || perm != nullptr), PSVAE_EINVAL, "%s: null output", fn);
    hipStream_t s = (hipStream_t)stream;
    const int nblk = sort_blocks(V);
    if (nblk == 1) {
        const size_t lds = (size_t)(ks.R + 2 * KS_THREADS) * sizeof(int);
        hipLaunchKernelGGL(key_sort_single_kernel, dim3(1), dim3(KS_THREADS), lds, s, ks, V, chunk, perm, row_off, unit_off);
        PSVAE_CHECK_LAUNCH(fn);
        return PSVAE_OK;
    }
    PSVAE_REQUIRE(workspace != nullptr && workspace_bytes >= psvae_key_sort_workspace_bytes(V, ks.R), PSVAE_EWORKSPACE,
                  "%s: workspace too small", fn);
    int32_t *hist = (int32_t *)workspace;
    const int per_block = psvae_cdiv(psvae_cdiv(V, nblk), KS_THREADS) * KS_THREADS;
    hipLaunchKernelGGL(key_hist_kernel, dim3(nblk), dim3(KS_THREADS), (size_t)ks.R * sizeof(int), s, ks, V, per_block, hist);
    PSVAE_CHECK_LAUNCH(fn);
    hipLaunchKernelGGL(key_offsets_kernel, dim3(1), dim3(KS_THREADS), (size_t)(ks.R + 2 * KS_THREADS) * sizeof(int), s, ks.R, nblk,
                       chunk, hist, row_off, unit_off);
    PSVAE_CHECK_LAUNCH(fn);
    hipLaunchKernelGGL(key_scatter_kernel, dim3(nblk), dim3(KS_THREADS), (size_t)ks.R * sizeof(int), s, ks, V, per_block, hist,
                       row_off, perm);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" size_t psvae_keyed_row_sum_workspace_bytes(long long entries, int R, int H, int chunk) {
    if (entries < 0 || R <= 0 || H <= 0 || chunk <= 0) return 0;
    return (size_t)max_units(entries, R, chunk) * (size_t)H * sizeof(float);
}

extern "C" int psvae_keyed_row_sum(int R, int H, long long entries, int chunk, const float *g, int ldg, const int32_t *perm,
                                   const int32_t *row_off, const int32_t *unit_off, float *out, int ldo, void *workspace,
                                   size_t workspace_bytes, void *stream) {
    const char *fn = "psvae_keyed_row_sum";
    PSVAE_REQUIRE(R >= 1 && H >= 4 && H % 4 == 0 && entries >= 0 && chunk >= 1, PSVAE_EINVAL,
                  "%s: bad sizes R=%d H=%d entries=%lld chunk=%d", fn, R, H, entries, chunk);
    PSVAE_REQUIRE(H <= 4096, PSVAE_ETOOLARGE, "%s: H=%d too wide", fn, H);
    PSVAE_REQUIRE(ldg % 4 == 0 && ldg >= H && ldo % 4 == 0 && ldo >= H, PSVAE_EINVAL, "%s: bad row stride", fn);
    PSVAE_REQUIRE(out != nullptr && psvae_aligned16(out) && row_off != nullptr && unit_off != nullptr &&
                      (entries == 0 || (g != nullptr && psvae_aligned16(g) && perm != nullptr)),
                  PSVAE_EINVAL, "%s: null or misaligned operand", fn);
    const long long units = max_units(entries, R, chunk);
    PSVAE_REQUIRE(units < (1ll << 31), PSVAE_ETOOLARGE, "%s: too many units", fn);
    PSVAE_REQUIRE(workspace != nullptr && psvae_aligned16(workspace) &&
                      workspace_bytes >= psvae_keyed_row_sum_workspace_bytes(entries, R, H, chunk),
                  PSVAE_EWORKSPACE, "%s: workspace too small", fn);
    hipStream_t s = (hipStream_t)stream;
    const int cols = H / 4;
    const int slots = cols >= 256 ? 1 : 256 / cols;
    const int threads = psvae_cdiv(cols * slots, 64) * 64;
    hipLaunchKernelGGL(keyed_sum_partial_kernel, dim3((unsigned)units), dim3(threads), (size_t)slots * cols * sizeof(float4), s, R,
                       cols, slots, chunk, (const float4 *)g, ldg / 4, perm, row_off, unit_off, (float4 *)workspace);
    PSVAE_CHECK_LAUNCH(fn);
    int groups = 1024 / cols;
    groups = groups < 1 ? 1 : (groups > 8 ? 8 : groups);
    const int fthreads = psvae_cdiv(cols * groups, 64) * 64;
    hipLaunchKernelGGL(keyed_sum_finish_kernel, dim3(R), dim3(fthreads), (size_t)groups * cols * sizeof(float4), s, cols, groups,
                       unit_off, (const float4 *)workspace, (float4 *)out, ldo / 4);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}
