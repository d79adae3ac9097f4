// Batch collation on the GPU: packed molecules -> the reference's batch tensors plus the
// compact CSR / pair-list layout the other kernels read.
//
// Restates BPEMolDataset.process_step2 (src/data/bpe_dataset.py:56-84) and process_step3
// (:86-155).  The reference expands each molecule into dense n x n Python lists and walks
// them cell by cell, appending to lists; the position of every appended item is therefore
// "number of earlier cells (in molecule, row, column order) with the same property".  Here
// one workgroup owns one molecule: it builds the n x n byte matrix (bond class, same-piece
// flag) in LDS and walks it 256 cells at a time; ranks inside a chunk come from wave
// ballots, ranks across chunks and molecules from running totals and the host-provided
// prefix sums.  All outputs are integer and bit-identical to the reference given the same
// uniform stream.
#include "common.h"

namespace {

constexpr int THREADS = 256;
constexpr int WAVES = THREADS / 64;

struct CollateArgs {
    int B, N, S;
    const int32_t *mol_ids;
    const int32_t *atom_off, *atom_type, *atom_pos;
    const int32_t *bond_off, *bond_a, *bond_b, *bond_cls;
    const int32_t *piece_off, *piece_id;
    const float *props_in;
    const int32_t *node_off, *edge_off, *inedge_off, *rng_off, *pair_off;
    const double *uniforms;
    double keep_prob;
    int atom_pad, piece_pad, num_bond_type;
    // pass 1 output
    int32_t *pair_cnt;
    // reference layout
    int64_t *x, *x_pieces, *x_pos;
    uint8_t *atom_mask;
    int64_t *pieces, *edge_index, *edge_attr;
    uint8_t *edge_select;
    int64_t *golden_edge, *in_piece_edge_idx;
    float *props_out;
    long long E;   // total directed edges (second row of edge_index starts at E)
    // compact layout
    int32_t *cx, *cxp, *cxpos, *row_ptr, *col;
    uint8_t *cls;
    int32_t *in_row_ptr, *in_col;
    uint8_t *in_cls;
    int32_t *pair_src, *pair_dst, *pair_graph, *golden32;
};

// Exclusive rank of this thread's flag among the 256 threads, and the block total.
// `totals` is a 4-int LDS array owned by this call site; the caller separates successive
// uses of the same array by a __syncthreads().
__device__ __forceinline__ int block_rank(bool flag, int lane, int wave, int *totals, int &block_total) {
    const unsigned long long b = __ballot(flag);
    const int in_wave = __popcll(b & ((1ull << lane) - 1ull));
    if (lane == 0) totals[wave] = __popcll(b);
    __syncthreads();
    int before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < WAVES; ++w) {
        const int t = totals[w];
        before += w < wave ? t : 0;
        total += t;
    }
    block_total = total;
    return before + in_wave;
}

template <bool FILL>
__global__ __launch_bounds__(THREADS) void collate_kernel(CollateArgs a) {
    extern __shared__ __align__(16) unsigned char lds[];
    __shared__ int tot_e[WAVES], tot_i[WAVES], tot_r[WAVES], tot_p[WAVES];

    const int b = blockIdx.x;
    const int mol = a.mol_ids[b];
    const int a0 = a.atom_off[mol], n = a.atom_off[mol + 1] - a0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    int *pos = reinterpret_cast<int *>(lds);                 // [n] (4-byte aligned at offset 0)
    unsigned char *mat = lds + 4 * (size_t)((n + 3) & ~3);   // [n*n] bond class per cell

    for (int i = tid; i < n; i += THREADS) pos[i] = a.atom_pos[a0 + i];
    for (int i = tid; i < n * n; i += THREADS) mat[i] = 0;
    __syncthreads();
    const int b0 = a.bond_off[mol], nb = a.bond_off[mol + 1] - b0;
    for (int i = tid; i < nb; i += THREADS) {
        const int u = a.bond_a[b0 + i], v = a.bond_b[b0 + i];
        const unsigned char c = (unsigned char)a.bond_cls[b0 + i];
        mat[u * n + v] = c;
        mat[v * n + u] = c;
    }
    __syncthreads();

    const int node_base = FILL ? a.node_off[b] : 0;
    const int edge_base = FILL ? a.edge_off[b] : 0;
    const int in_base = FILL ? a.inedge_off[b] : 0;
    const int pair_base = FILL ? a.pair_off[b] : 0;
    const long long rng_base = a.rng_off[b];
    int run_e = 0, run_i = 0, run_r = 0, run_p = 0;

    for (int base = 0; base < n * n; base += THREADS) {
        const int idx = base + tid;
        const bool valid = idx < n * n;
        const int m = valid ? idx / n : 0, k = valid ? idx - (idx / n) * n : 0;
        const int attr = valid ? mat[idx] : 0;
        const bool same = valid && pos[m] == pos[k] && pos[m] > 0;   // edge_select == 0, :65-71
        const bool is_edge = valid && attr != 0;
        const bool is_in = is_edge && same;
        const bool is_rng = valid && !same && attr == 0;
        int te, ti, tr, tp;
        int re = 0, ri = 0;
        if (FILL) {
            re = run_e + block_rank(is_edge, lane, wave, tot_e, te);
            ri = run_i + block_rank(is_in, lane, wave, tot_i, ti);
        }
        const int rr = run_r + block_rank(is_rng, lane, wave, tot_r, tr);
        // :127-132 — bonded candidate pairs are always kept; the others draw one uniform each
        bool keep = valid && !same && attr != 0;
        if (is_rng) keep = a.uniforms[rng_base + rr] < a.keep_prob;
        const int rp = run_p + block_rank(keep, lane, wave, tot_p, tp);
        if (FILL) {
            if (valid && k == 0) {
                if (a.row_ptr) a.row_ptr[node_base + m] = edge_base + re;
                if (a.in_row_ptr) a.in_row_ptr[node_base + m] = in_base + ri;
            }
            if (is_edge) {
                const long long e = (long long)edge_base + re;
                if (a.edge_index) {
                    a.edge_index[e] = (long long)b * a.N + m;
                    a.edge_index[a.E + e] = (long long)b * a.N + k;
                    for (int c = 0; c < a.num_bond_type; ++c) a.edge_attr[e * a.num_bond_type + c] = (c == attr);
                }
                if (a.col) {
                    a.col[e] = node_base + k;
                    a.cls[e] = (uint8_t)attr;
                }
                if (is_in) {
                    const long long ei = (long long)in_base + ri;
                    if (a.in_piece_edge_idx) a.in_piece_edge_idx[ei] = e;
                    if (a.in_col) {
                        a.in_col[ei] = node_base + k;
                        a.in_cls[ei] = (uint8_t)attr;
                    }
                }
            }
            if (valid && a.edge_select) a.edge_select[((long long)b * a.N + m) * a.N + k] = keep ? 1 : 0;
            if (keep) {
                const long long p = (long long)pair_base + rp;
                if (a.golden_edge) a.golden_edge[p] = attr;
                if (a.pair_src) {
                    a.pair_src[p] = node_base + m;
                    a.pair_dst[p] = node_base + k;
                    a.pair_graph[p] = b;
                    a.golden32[p] = attr;
                }
            }
            run_e += te;
            run_i += ti;
        }
        run_r += tr;
        run_p += tp;
        __syncthreads();   // the totals arrays are rewritten by the next chunk
    }

    if (!FILL) {
        if (tid == 0) a.pair_cnt[b] = run_p;
        return;
    }

    if (b == a.B - 1 && tid == 0) {
        if (a.row_ptr) a.row_ptr[node_base + n] = a.edge_off[a.B];
        if (a.in_row_ptr) a.in_row_ptr[node_base + n] = a.inedge_off[a.B];
    }
    const int p0 = a.piece_off[mol], np = a.piece_off[mol + 1] - p0;
    // per-atom tensors, :89-103; x_pieces = pieces[group position], :72-74
    for (int j = tid; j < a.N; j += THREADS) {
        const bool real = j < n;
        const int ps = real ? pos[j] : 0;
        const int at = real ? a.atom_type[a0 + j] : a.atom_pad;
        const int pc = real ? (ps > 0 ? a.piece_id[p0 + ps] : 0) : a.piece_pad;
        if (a.x) {
            const long long o = (long long)b * a.N + j;
            a.x[o] = at;
            a.x_pieces[o] = pc;
            a.x_pos[o] = ps;
            a.atom_mask[o] = real ? 0 : 1;
        }
        if (real && a.cx) {
            a.cx[node_base + j] = at;
            a.cxp[node_base + j] = pc;
            a.cxpos[node_base + j] = ps;
        }
    }
    if (a.pieces)
        for (int j = tid; j < a.S; j += THREADS) a.pieces[(long long)b * a.S + j] = j < np ? a.piece_id[p0 + j] : a.piece_pad;
    if (a.props_out && tid < 3) a.props_out[b * 3 + tid] = a.props_in[mol * 3 + tid];
    // padding cells of edge_select (:108 zero-initialises the whole [B,N,N] tensor)
    if (a.edge_select) {
        const int NN = a.N * a.N;
        for (int i = tid; i < NN; i += THREADS) {
            const int m = i / a.N, k = i - m * a.N;
            if (m >= n || k >= n) a.edge_select[(long long)b * NN + i] = 0;
        }
    }
}

// Single-workgroup exclusive scan (n is a batch size: thousands).
__global__ __launch_bounds__(1024) void exclusive_scan_kernel(int n, const int32_t *__restrict__ in,
                                                              int32_t *__restrict__ out) {
    __shared__ int wave_tot[16];
    __shared__ int carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (int base = 0; base < n; base += 1024) {
        const int i = base + tid;
        const int v = i < n ? in[i] : 0;
        int s = v;   // inclusive scan inside the wave
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            const int t = __shfl_up(s, d);
            if (lane >= d) s += t;
        }
        if (lane == 63) wave_tot[wave] = s;
        __syncthreads();
        int before = carry_s, total = 0;
        for (int w = 0; w < 16; ++w) {
            const int t = wave_tot[w];
            before += w < wave ? t : 0;
            total += t;
        }
        if (i < n) out[i] = before + s - v;
        __syncthreads();
        if (tid == 0) carry_s += total;
        __syncthreads();
    }
    if (tid == 0) out[n] = carry_s;
}

size_t lds_bytes(int N) { return 4 * (size_t)((N + 3) & ~3) + (size_t)N * N + 16; }

}  // namespace

extern "C" int psvae_exclusive_scan_i32(int n, const int32_t *in, int32_t *out, void *stream) {
    PSVAE_REQUIRE(n >= 0 && n <= (1 << 24), PSVAE_EINVAL, "psvae_exclusive_scan_i32: bad n %d", n);
    PSVAE_REQUIRE(out != nullptr && (n == 0 || in != nullptr), PSVAE_EINVAL, "psvae_exclusive_scan_i32: null pointer");
    hipLaunchKernelGGL(exclusive_scan_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, n, in, out);
    PSVAE_CHECK_LAUNCH("psvae_exclusive_scan_i32");
    return PSVAE_OK;
}

extern "C" int psvae_collate_count(int B, int N, const int32_t *mol_ids, const int32_t *atom_off,
                                   const int32_t *atom_pos, const int32_t *bond_off, const int32_t *bond_a,
                                   const int32_t *bond_b, const int32_t *bond_cls, const int32_t *rng_off,
                                   const double *uniforms, double keep_prob, int32_t *pair_cnt,
                                   int32_t *pair_off, void *stream) {
    PSVAE_REQUIRE(B >= 0 && N >= 0, PSVAE_EINVAL, "psvae_collate_count: bad sizes");
    PSVAE_REQUIRE(N <= PSVAE_MAX_ATOMS, PSVAE_ETOOLARGE, "psvae_collate_count: %d atoms > PSVAE_MAX_ATOMS", N);
    PSVAE_REQUIRE(pair_off != nullptr, PSVAE_EINVAL, "psvae_collate_count: pair_off is null");
    if (B > 0) {
        PSVAE_REQUIRE(mol_ids && atom_off && atom_pos && bond_off && rng_off && pair_cnt, PSVAE_EINVAL,
                      "psvae_collate_count: null pointer");
        CollateArgs a = {};
        a.B = B, a.N = N;
        a.mol_ids = mol_ids, a.atom_off = atom_off, a.atom_pos = atom_pos;
        a.bond_off = bond_off, a.bond_a = bond_a, a.bond_b = bond_b, a.bond_cls = bond_cls;
        a.rng_off = rng_off, a.uniforms = uniforms, a.keep_prob = keep_prob, a.pair_cnt = pair_cnt;
        hipLaunchKernelGGL(collate_kernel<false>, dim3(B), dim3(THREADS), lds_bytes(N), (hipStream_t)stream, a);
        PSVAE_CHECK_LAUNCH("psvae_collate_count");
    }
    return psvae_exclusive_scan_i32(B, pair_cnt, pair_off, stream);
}

extern "C" int psvae_collate_fill(
    int B, int N, int S, const int32_t *mol_ids, const int32_t *atom_off, const int32_t *atom_type,
    const int32_t *atom_pos, const int32_t *bond_off, const int32_t *bond_a, const int32_t *bond_b,
    const int32_t *bond_cls, const int32_t *piece_off, const int32_t *piece_id, const float *props_in,
    const int32_t *node_off, const int32_t *edge_off, const int32_t *inedge_off, const int32_t *rng_off,
    const int32_t *pair_off, long long E, const double *uniforms, double keep_prob, int atom_pad, int piece_pad,
    int64_t *x, int64_t *x_pieces, int64_t *x_pos, uint8_t *atom_mask, int64_t *pieces, int64_t *edge_index,
    int64_t *edge_attr, int num_bond_type, uint8_t *edge_select, int64_t *golden_edge,
    int64_t *in_piece_edge_idx, float *props_out, int32_t *cx, int32_t *cxp, int32_t *cxpos, int32_t *row_ptr,
    int32_t *col, uint8_t *cls, int32_t *in_row_ptr, int32_t *in_col, uint8_t *in_cls, int32_t *pair_src,
    int32_t *pair_dst, int32_t *pair_graph, int32_t *golden32, void *stream) {
    PSVAE_REQUIRE(B >= 0 && N >= 0 && S >= 0 && E >= 0, PSVAE_EINVAL, "psvae_collate_fill: bad sizes");
    PSVAE_REQUIRE(N <= PSVAE_MAX_ATOMS, PSVAE_ETOOLARGE, "psvae_collate_fill: %d atoms > PSVAE_MAX_ATOMS", N);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(mol_ids && atom_off && atom_type && atom_pos && bond_off && piece_off && piece_id && node_off &&
                      edge_off && inedge_off && rng_off && pair_off, PSVAE_EINVAL, "psvae_collate_fill: null pointer");
    PSVAE_REQUIRE((x != nullptr) == (x_pieces != nullptr) && (x != nullptr) == (x_pos != nullptr) &&
                      (x != nullptr) == (atom_mask != nullptr), PSVAE_EINVAL,
                  "psvae_collate_fill: x, x_pieces, x_pos, atom_mask go together");
    PSVAE_REQUIRE((edge_index != nullptr) == (edge_attr != nullptr), PSVAE_EINVAL,
                  "psvae_collate_fill: edge_index and edge_attr go together");
    PSVAE_REQUIRE((col != nullptr) == (cls != nullptr) && (col != nullptr) == (row_ptr != nullptr), PSVAE_EINVAL,
                  "psvae_collate_fill: row_ptr, col, cls go together");
    PSVAE_REQUIRE((in_col != nullptr) == (in_cls != nullptr) && (in_col != nullptr) == (in_row_ptr != nullptr),
                  PSVAE_EINVAL, "psvae_collate_fill: in_row_ptr, in_col, in_cls go together");
    PSVAE_REQUIRE((pair_src != nullptr) == (pair_dst != nullptr) && (pair_src != nullptr) == (pair_graph != nullptr) &&
                      (pair_src != nullptr) == (golden32 != nullptr), PSVAE_EINVAL,
                  "psvae_collate_fill: pair_src, pair_dst, pair_graph, golden32 go together");
    PSVAE_REQUIRE((cx != nullptr) == (cxp != nullptr) && (cx != nullptr) == (cxpos != nullptr), PSVAE_EINVAL,
                  "psvae_collate_fill: cx, cxp, cxpos go together");
    PSVAE_REQUIRE(props_out == nullptr || props_in != nullptr, PSVAE_EINVAL, "psvae_collate_fill: props_in is null");
    CollateArgs a = {};
    a.B = B, a.N = N, a.S = S, a.mol_ids = mol_ids;
    a.atom_off = atom_off, a.atom_type = atom_type, a.atom_pos = atom_pos;
    a.bond_off = bond_off, a.bond_a = bond_a, a.bond_b = bond_b, a.bond_cls = bond_cls;
    a.piece_off = piece_off, a.piece_id = piece_id, a.props_in = props_in;
    a.node_off = node_off, a.edge_off = edge_off, a.inedge_off = inedge_off, a.rng_off = rng_off, a.pair_off = pair_off;
    a.uniforms = uniforms, a.keep_prob = keep_prob, a.atom_pad = atom_pad, a.piece_pad = piece_pad;
    a.num_bond_type = num_bond_type, a.E = E;
    a.x = x, a.x_pieces = x_pieces, a.x_pos = x_pos, a.atom_mask = atom_mask, a.pieces = pieces;
    a.edge_index = edge_index, a.edge_attr = edge_attr, a.edge_select = edge_select, a.golden_edge = golden_edge;
    a.in_piece_edge_idx = in_piece_edge_idx, a.props_out = props_out;
    a.cx = cx, a.cxp = cxp, a.cxpos = cxpos, a.row_ptr = row_ptr, a.col = col, a.cls = cls;
    a.in_row_ptr = in_row_ptr, a.in_col = in_col, a.in_cls = in_cls;
    a.pair_src = pair_src, a.pair_dst = pair_dst, a.pair_graph = pair_graph, a.golden32 = golden32;
    hipLaunchKernelGGL(collate_kernel<true>, dim3(B), dim3(THREADS), lds_bytes(N), (hipStream_t)stream, a);
    PSVAE_CHECK_LAUNCH("psvae_collate_fill");
    return PSVAE_OK;
}
