/*
 * psvae_hip.h — C ABI of libpsvae_hip.so, the gfx950 (MI355X) kernels behind PS-VAE's
 * training hot path.
 *
 * The reference (THUNLP-MT/PS-VAE) is pure Python and has no FFI of its own; each entry
 * point below names the reference lines whose arithmetic it replaces.  The Python host
 * side (ps-vae_amd/hip.py) binds these symbols with ctypes and the module classes in
 * ps-vae_amd/modules mirror the reference's module API on top of them.
 *
 * Conventions
 *   - Every function returns 0 on success, a negative PSVAE_E* code for a rejected
 *     argument (nothing was launched), or a positive hipError_t from the runtime.
 *     psvae_last_error() returns a thread-local message for the last non-zero return.
 *   - All pointers are DEVICE pointers unless a name ends in _host.  The caller owns every
 *     buffer; the library allocates nothing and is re-entrant.  It keeps no state between calls
 *     apart from two per-device facts looked up once under a mutex: the address of the fixed
 *     scratch array of psvae_gine_aggregate_bwd_noalloc and the raised dynamic-LDS limit of the
 *     tall-K product kernel.  Both are keyed by the CURRENT device: make the device that owns the
 *     buffers current before a call.
 *   - Work is enqueued on `stream` (a hipStream_t passed as void*); nothing synchronises.
 *   - Matrices are row-major float32; `ld*` are row strides in ELEMENTS.
 *   - int32 index arrays; bond classes are uint8 (0 = no bond .. 3 = triple,
 *     reference src/utils/chem_utils.py:15).
 */
#ifndef PSVAE_HIP_H
#define PSVAE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PSVAE_OK 0
#define PSVAE_EINVAL (-1)      /* bad size / null pointer / misaligned operand        */
#define PSVAE_ETOOLARGE (-2)   /* a dimension exceeds what the kernel supports         */
#define PSVAE_EWORKSPACE (-3)  /* workspace_bytes smaller than the *_workspace_bytes() */

#define PSVAE_MAX_ATOMS 250    /* atoms per molecule the collation kernels accept      */
#define PSVAE_ACT_NONE 0
#define PSVAE_ACT_RELU 1

int psvae_version(void);
const char *psvae_last_error(void);

/* ------------------------------------------------------------------------------------
 * Collation.  Replaces BPEMolDataset.process_step2 + process_step3
 * (src/data/bpe_dataset.py:56-84, 86-155) on molecules held in the packed layout
 * (ps-vae_amd/data/packed.py).  One workgroup per molecule builds the n x n bond-class /
 * same-piece matrix in LDS and walks it row-major, so every output index is the one the
 * reference's triple loop produces.
 *
 * mol_ids [B]            molecules of this batch, in batch order
 * rng_off [B+1]          prefix sum of n_rng over the batch: the j-th non-bonded candidate
 *                        pair of molecule b reads uniforms[rng_off[b] + j], which is the
 *                        order in which the reference calls random() (:129)
 * keep_prob              0.022 in the reference
 * ---------------------------------------------------------------------------------- */

/* Pass 1: pair_off[b] = number of kept candidate pairs in molecules 0..b-1 (exclusive
 * prefix sum, pair_off[B] = P).  pair_cnt [B] is scratch.  N = largest atom count in the
 * batch (it sizes the LDS matrix). */
int psvae_collate_count(int B, int N, const int32_t *mol_ids,
                        const int32_t *atom_off, const int32_t *atom_pos,
                        const int32_t *bond_off, const int32_t *bond_a, const int32_t *bond_b,
                        const int32_t *bond_cls,
                        const int32_t *rng_off, const double *uniforms, double keep_prob,
                        int32_t *pair_cnt, int32_t *pair_off, void *stream);

/* Pass 2: write the batch.  Any output pointer may be NULL to skip that tensor (all
 * tensors of one group are skipped or written together).
 *
 * Offsets (host-computed prefix sums over the batch, device arrays [B+1]):
 *   node_off, edge_off, inedge_off, rng_off, pair_off (from pass 1); E = edge_off[B]
 * Reference layout (shapes / dtypes of the dict process_step3 returns, :142-155):
 *   x, x_pieces, x_pos int64 [B,N]; atom_mask uint8(bool) [B,N]; pieces int64 [B,S];
 *   edge_index int64 [2,E] (node id = b*N + m); edge_attr int64 [E,4] one-hot;
 *   edge_select uint8(bool) [B,N,N]; golden_edge int64 [P]; in_piece_edge_idx int64 [E'];
 *   props float32 [B,3]
 * Compact layout (real atoms only, node id = node_off[b] + m):
 *   cx, cxp, cxpos int32 [V]; row_ptr int32 [V+1], col int32 [E], cls uint8 [E];
 *   in_row_ptr [V+1], in_col [E'], in_cls [E']; pair_src, pair_dst, pair_graph, golden32 [P]
 */
int psvae_collate_fill(int B, int N, int S, const int32_t *mol_ids,
                       const int32_t *atom_off, const int32_t *atom_type, const int32_t *atom_pos,
                       const int32_t *bond_off, const int32_t *bond_a, const int32_t *bond_b,
                       const int32_t *bond_cls,
                       const int32_t *piece_off, const int32_t *piece_id, const float *props_in,
                       const int32_t *node_off, const int32_t *edge_off, const int32_t *inedge_off,
                       const int32_t *rng_off, const int32_t *pair_off, long long E,
                       const double *uniforms, double keep_prob, int atom_pad, int piece_pad,
                       /* reference layout */
                       int64_t *x, int64_t *x_pieces, int64_t *x_pos, uint8_t *atom_mask,
                       int64_t *pieces, int64_t *edge_index, int64_t *edge_attr, int num_bond_type,
                       uint8_t *edge_select, int64_t *golden_edge, int64_t *in_piece_edge_idx,
                       float *props_out,
                       /* compact layout */
                       int32_t *cx, int32_t *cxp, int32_t *cxpos,
                       int32_t *row_ptr, int32_t *col, uint8_t *cls,
                       int32_t *in_row_ptr, int32_t *in_col, uint8_t *in_cls,
                       int32_t *pair_src, int32_t *pair_dst, int32_t *pair_graph, int32_t *golden32,
                       void *stream);

/* out[0] = 0, out[i+1] = in[0] + .. + in[i]  (n <= 2^24) */
int psvae_exclusive_scan_i32(int n, const int32_t *in, int32_t *out, void *stream);

/* ------------------------------------------------------------------------------------
 * embed_atom: out[v] = [ E_atom[x[v]] | E_piece[xp[v]] | E_pos[xq[v]] ]
 * (src/modules/vae_piece_decoder.py:65-69).  Backward adds g_out rows into the three
 * table gradients (float atomics; caller zero-fills them).  An id outside its table is clamped
 * into it, by the same rule in every entry point that takes ids (forward, backward, indicator,
 * psvae_embed_trans_fwd, psvae_embedding_fwd): nothing is read or added outside a table.
 * ---------------------------------------------------------------------------------- */
int psvae_embed_atom_fwd(int V, const int32_t *x, const int32_t *xp, const int32_t *xq,
                         const float *e_atom, int n_atom, int d_atom,
                         const float *e_piece, int n_piece, int d_piece,
                         const float *e_pos, int n_pos, int d_pos,
                         float *out, void *stream);
int psvae_embed_atom_bwd(int V, const int32_t *x, const int32_t *xp, const int32_t *xq,
                         const float *g_out, int n_atom, int d_atom, int n_piece, int d_piece,
                         int n_pos, int d_pos,
                         float *g_atom, float *g_piece, float *g_pos, void *stream);
/* Indicator rows for the deterministic form of the same backward: out[v][j] = 1 where
 * j == x[v], j == n_atom + xp[v] or j == n_atom + n_piece + xq[v], else 0, for j < ld
 * (ld % 4 == 0, ld >= n_atom + n_piece + n_pos).  out^T g_out, one psvae_gemm_f32 call,
 * then holds the three table gradients in its diagonal blocks. */
int psvae_embed_atom_onehot(int V, const int32_t *x, const int32_t *xp, const int32_t *xq,
                            int n_atom, int n_piece, int n_pos, float *out, int ld, void *stream);

/* A single embedding table (the piece embedding of the language model,
 * src/modules/vae_piece_decoder.py:26,78): out[v] = table[ids[v]], rows of D floats (D % 4 == 0),
 * ids outside [0, n_rows) clamped into it.  psvae_onehot_f32 writes the indicator rows
 * out[v][j] = (j == ids[v]), j < ld (ld % 4 == 0, ld >= n_rows): out^T g, one psvae_gemm_f32 call,
 * is the table gradient, summed in a fixed order. */
int psvae_embedding_fwd(int V, int D, int n_rows, const int32_t *ids, const float *table, float *out,
                        void *stream);
int psvae_onehot_f32(int V, const int32_t *ids, float *out, int ld, void *stream);

/* ------------------------------------------------------------------------------------
 * Deterministic table gradients of embedding lookups (the backward of nn.Embedding at
 * src/modules/vae_piece_decoder.py:25-27, 65-69, 78): g_table[r] = sum of g[v] over the entries v
 * whose id is r.  psvae_key_sort puts the entries in id order once per batch (stable counting
 * sort, integers only); psvae_keyed_row_sum then adds, for every table row, its run of gathered
 * rows of g in a fixed order (units of `chunk` entries, then the units of a row).  No float
 * atomics, no [entries x rows] indicator matrix; equal calls give equal bits.
 *
 * psvae_key_sort: up to three id arrays k0, k1, k2 [V] over the same entries with table sizes
 * n0, n1, n2 (ids outside a table are clamped into it, as in every entry point that takes ids;
 * pass NULL / 0 for unused tables) are sorted into ONE permutation over the stacked table
 * (table t's rows follow those of table t-1, R = n0 + n1 + n2 <= 8192 rows):
 *   perm     [n_tables * V]  entry indices 0..V-1, grouped by stacked row, ascending inside a row
 *   row_off  [R + 1]         entries of stacked row r are perm[row_off[r] .. row_off[r+1])
 *   unit_off [R + 1]         prefix sum of ceil(entries of row r / chunk): the work units of the sum
 * psvae_keyed_row_sum: out[r] = sum over i in [row_off[r], row_off[r+1]) of g[perm[i]], rows of H
 * floats (H, ldg, ldo multiples of 4; 16-byte aligned); rows without entries get zeros.
 * `entries` = length of perm, `chunk` = the value given to psvae_key_sort.
 * ---------------------------------------------------------------------------------- */
size_t psvae_key_sort_workspace_bytes(int V, int n_rows_total);
int psvae_key_sort(int V, int n_tables, const int32_t *k0, const int32_t *k1, const int32_t *k2,
                   int n0, int n1, int n2, int chunk, int32_t *perm, int32_t *row_off,
                   int32_t *unit_off, void *workspace, size_t workspace_bytes, void *stream);
size_t psvae_keyed_row_sum_workspace_bytes(long long entries, int R, int H, int chunk);
int psvae_keyed_row_sum(int R, int H, long long entries, int chunk, const float *g, int ldg,
                        const int32_t *perm, const int32_t *row_off, const int32_t *unit_off,
                        float *out, int ldo, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * GINE aggregation, the message/aggregate half of torch_geometric.nn.GINEConv as called at
 * src/modules/encoder.py:58 with eps = 0:
 *     out[i] = x[i] + sum over edges e into i of ReLU(x[col[e]] + table[cls[e]])
 * row_ptr/col/cls: CSR over destination nodes (col = source node of the edge).
 * table [C,H] = edge_trans applied to the C one-hot rows (src/modules/encoder.py:55).
 * Requires H % 4 == 0, ld* % 4 == 0 and 16-byte aligned rows.
 *
 * Backward, given g = dL/d out and the CSR over SOURCE nodes (t_col = destination):
 *     g_x[j]     = g[j] + sum over edges e out of j of g[t_col[e]] * [x[j] + table[t_cls[e]] > 0]
 *     g_table[c] = sum over edges e with class c of the same masked rows
 * `add` (optional, [V,H]) is added to g_x — the gradient that reaches x directly (the
 * encoder's concatenated output slot).  accumulate_table != 0 adds to g_table (the SAME
 * conv is applied t times, src/modules/encoder.py:57-59).  g_table is reduced
 * deterministically through `workspace` (psvae_gine_aggregate_bwd_workspace_bytes).
 * ---------------------------------------------------------------------------------- */
int psvae_gine_aggregate_fwd(int V, int H, int C, const float *x, int ldx, const float *table,
                             const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                             float *out, int ldo, void *stream);
size_t psvae_gine_aggregate_bwd_workspace_bytes(int V, int H, int C);
int psvae_gine_aggregate_bwd(int V, int H, int C, const float *x, int ldx, const float *table,
                             const float *g, int ldg,
                             const int32_t *t_row_ptr, const int32_t *t_col, const uint8_t *t_cls,
                             const float *add, int ldadd, float *g_x, int ldgx, float *g_table,
                             int accumulate_table, void *workspace, size_t workspace_bytes, void *stream);

/* psvae_gine_aggregate_bwd with a per-graph add: row seg_add[node_graph[j]] (row stride ld_seg_add;
 * both NULL for none) is added to g_x[j] - the gradient of a sum pooled over each graph's nodes
 * (src/modules/encoder.py:63-67) is one row per graph, read where it is needed instead of being
 * spread to [V, H] first. */
int psvae_gine_aggregate_bwd_graph_add(int V, int H, int C, const float *x, int ldx, const float *table,
                                       const float *g, int ldg,
                                       const int32_t *t_row_ptr, const int32_t *t_col, const uint8_t *t_cls,
                                       const float *add, int ldadd, const float *seg_add, int ld_seg_add,
                                       const int32_t *node_graph, float *g_x, int ldgx, float *g_table,
                                       int accumulate_table, void *workspace, size_t workspace_bytes, void *stream);

/* The same pair with the ReLU decisions carried as bits.  The message node j sends along an edge of
 * class k is ReLU(x[j] + table[k]): its derivative depends on (j, k) only, so the forward leaves
 * 4 C bits per float4 column of every node behind - mask_bits [V][H / 4], 16-bit words for C <= 4,
 * 32-bit words for C <= 8 (bit 4 k + e: element e of the column under class k;
 * psvae_gine_mask_bytes(V, H, C) bytes, aligned to its word size) - and the backward reads those instead of
 * x and the table: H / 2 bytes per node instead of 4 H.  seg_add / node_graph (optional): row
 * seg_add[node_graph[j]] (row stride ld_seg_add) is added to g_x[j] - the gradient of a sum pooled
 * over each graph's nodes (src/modules/encoder.py:63-67), which is one row per graph, read where it
 * is needed instead of being spread to [V, H] first.  Other arguments as above. */
size_t psvae_gine_mask_bytes(int V, int H, int C);
int psvae_gine_aggregate_fwd_bits(int V, int H, int C, const float *x, int ldx, const float *table,
                                  const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                                  float *out, int ldo, void *mask_bits, void *stream);
int psvae_gine_aggregate_bwd_bits(int V, int H, int C, const void *mask_bits, const float *g, int ldg,
                                  const int32_t *t_row_ptr, const int32_t *t_col, const uint8_t *t_cls,
                                  const float *add, int ldadd, const float *seg_add, int ld_seg_add,
                                  const int32_t *node_graph, float *g_x, int ldgx, float *g_table,
                                  int accumulate_table, void *workspace, size_t workspace_bytes, void *stream);
/* accumulate_table of psvae_gine_aggregate_bwd_bits may be PSVAE_TABLE_PARTIALS_ONLY: the [C,H] slabs of
 * table-gradient partials (psvae_gine_aggregate_bwd_slabs(V, H, C) of them, one per workgroup) stay in
 * `workspace`, g_table is not touched (it may be null) and the caller sums the slabs of several calls at
 * once with psvae_gine_table_reduce - the t layers of an encoder share one bond table, so their backward
 * passes need one reduction, not t. */
#define PSVAE_TABLE_PARTIALS_ONLY 2
int psvae_gine_aggregate_bwd_slabs(int V, int H, int C);
int psvae_gine_table_reduce(int n_slabs, int C, int H, const float *partials, float *g_table, int accumulate,
                            void *stream);

/* psvae_gine_aggregate_bwd for callers that cannot allocate.  It has no counterpart in the
 * reference; it exists for code under HIP graph capture, tight serving loops and harnesses that
 * forbid hipMalloc.  Same arguments and results as psvae_gine_aggregate_bwd, without the
 * workspace: the table-gradient partials go to a fixed device array inside the library
 * (1024 x PSVAE_NOALLOC_MAX_CH floats = 8 MiB, held by every GPU the library runs on until the
 * process ends).  C * H > PSVAE_NOALLOC_MAX_CH is rejected with PSVAE_ETOOLARGE; V is not
 * limited.  It calls no hipMalloc / hipFree.  The first call on a device looks the array's
 * address up (make that call — V = 0 is enough — outside graph capture).  All calls on one GPU
 * share the array: calls that overlap on one GPU (two streams or two threads) race. */
#define PSVAE_NOALLOC_MAX_CH 2048
int psvae_gine_aggregate_bwd_noalloc(int V, int H, int C, const float *x, int ldx, const float *table,
                                     const float *g, int ldg,
                                     const int32_t *t_row_ptr, const int32_t *t_col, const uint8_t *t_cls,
                                     const float *add, int ldadd, float *g_x, int ldgx, float *g_table,
                                     int accumulate_table, void *stream);

/* ------------------------------------------------------------------------------------
 * Hidden layer + class layer of the link predictor over all candidate pairs of generated molecules
 * (src/modules/vae_piece_decoder.py:224-238: the MLP's Linear + ReLU followed by the class Linear), the
 * [M,N] hidden rows never written:
 *   partials[m][t][c] = sum over the columns n of column tile t of relu(A[m] . W[n] + bias[n]) * cls_w[c][n]
 * for the 4 classes c; logits[m][c] = sum_t partials[m][t][c] + the class bias (left to the caller).
 * A [M,K] rows lda apart, W [N,K] rows ldw apart, bias [N] or null, cls_w [4][N] contiguous (rows of unused
 * classes zero), partials [M][tiles][4] with tiles = psvae_gemm_f32_relu_classes_tiles(N) (0: this N is not
 * covered - the column tiles must be 129..192 columns wide, e.g. N in 132..192 or 260..384).
 * ---------------------------------------------------------------------------------- */
int psvae_gemm_f32_relu_classes_tiles(int N);
int psvae_gemm_f32_relu_classes(int M, int N, int K, const float *A, int64_t lda, const float *W, int64_t ldw,
                                const float *bias, const float *cls_w, float *partials, void *stream);

/* ------------------------------------------------------------------------------------
 * f32 matrix product on the matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chains):
 *     C[m][n] (+)= act( sum_k A(m,k) * B(k,n) + bias[n] )
 * A(m,k) = A[m*sa_m + k*sa_k], B(k,n) = B[k*sb_k + n*sb_n]: any of the Linear products
 * (src/modules/common_nn.py:8-23; forward X W^T, input gradient dY W, weight gradient
 * dY^T X) is one call.  amask (optional, addressed like A) gates A: A(m,k) counts as 0 where
 * amask(m,k) <= 0 — the ReLU derivative applied to a gradient operand as it is loaded.
 * cmask (optional, [M,N] with the row stride of C) gates the result instead: C[m][n] = 0
 * where cmask[m][n] <= 0 — the ReLU derivative applied once, as a gradient is produced, so that
 * the products that consume it need no gate.
 * bias may be NULL.  rowsum (optional, [M]) receives sum_k A(m,k) (after the gate) — with
 * A = dY^T that is the bias gradient, computed in the same pass as the weight gradient.
 * accumulate != 0 adds to C (and rowsum).  split_k > 1 reduces partial products through
 * `workspace` in a fixed order; split_k == 0 lets the library choose.
 * Operands with unit stride along one axis, 16-byte aligned bases and major strides that
 * are multiples of 4 take the fast kernels (128-row panels; 64 x 64 tiles, whose 64-deep k-tiles
 * are summed in two levels, when the panels would give fewer than 256 workgroups; a kernel of its
 * own for weight gradients over more than 16 384 rows); anything else a slower generic one.
 * Which kernel runs depends on the shapes and strides only: equal calls give equal bits.
 * ---------------------------------------------------------------------------------- */
size_t psvae_gemm_f32_workspace_bytes(int M, int N, int K, int split_k);
int psvae_gemm_f32(int M, int N, int K,
                   const float *A, const float *amask, int64_t sa_m, int64_t sa_k,
                   const float *B, int64_t sb_k, int64_t sb_n,
                   float *C, int64_t ldc, const float *bias, float *rowsum, const float *cmask,
                   int act, int accumulate,
                   int split_k, void *workspace, size_t workspace_bytes, void *stream);

/* The same product with the ReLU decisions carried as BITS instead of a full-width gate (no
 * amask / cmask / rowsum / accumulate / K split): one 64-bit lane mask per 32 x 32 tile of C, row
 * group i (0..3) and component (0..3), at index ((row / 32) * ceil(N / 32) + col / 32) * 16 + 4 i +
 * component; bit l of it belongs to row 32 * (row / 32) + (l >> 3) + 8 i, column
 * 32 * (col / 32) + 4 * (l & 7) + component.  psvae_gemm_f32_bits_words(M, N) 64-bit words.
 * relu_bits (optional): written, bit = (C[m][n] > 0) after bias and activation - the forward
 * Linear + ReLU of src/modules/common_nn.py:15-20 leaves its derivative behind in M N / 8 bytes.
 * gate_bits (optional): read, C[m][n] = 0 where its bit is clear - the input gradient through that
 * ReLU, gated as it is produced, without re-reading the M x N activations.  The masks are fetched
 * by scalar loads and applied as lane selects: the gated product runs at the speed of the plain one.
 * M >= 32 and 16-byte aligned operands with multiple-of-4 strides are required (PSVAE_EINVAL
 * otherwise: use psvae_gemm_f32 with cmask). */
size_t psvae_gemm_f32_bits_words(int M, int N);
int psvae_gemm_f32_bits(int M, int N, int K, const float *A, int64_t sa_m, int64_t sa_k,
                        const float *B, int64_t sb_k, int64_t sb_n, float *C, int64_t ldc,
                        const float *bias, int act, uint64_t *relu_bits, const uint64_t *gate_bits,
                        void *stream);

/* out[v][n] = q[node_graph[v]][n] where the bit of (v, n) in gate_bits is set, else 0 (gate_bits: the
 * sign-bit array of psvae_gemm_f32_bits over an [V, N] result).  The input gradient through a
 * Linear + ReLU when the incoming gradient is the same row for every node of a graph (the last
 * encoder iterate feeds only the per-graph pooled sum, src/modules/encoder.py:60-67): the product
 * dY W runs once per graph, q = [graphs, N], and this kernel spreads and gates it.  N, ldq, ldo
 * multiples of 4; 16-byte aligned rows. */
int psvae_graph_rows_gated(int V, int N, const float *q, int ldq, const int32_t *node_graph,
                           const uint64_t *gate_bits, float *out, int ldo, void *stream);

/* out[n] (+)= sum_m x[m][n] * [mask[m][n] > 0] (mask optional, same ld as x);
 * deterministic two-stage reduction through `workspace`.  The bias gradient of a Linear. */
size_t psvae_colsum_workspace_bytes(int M, int N);
int psvae_colsum_f32(int M, int N, const float *x, const float *mask, int64_t ldx, float *out,
                     int accumulate, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * embed_graph pooling (src/modules/encoder.py:63-67): out[b] = sum of rows
 * seg_begin[b] .. seg_end[b]-1 of x.  Backward: g_x[v] = g_out[b] for rows inside segment b
 * (rows outside every segment are left untouched).  W % 4 == 0.
 * ---------------------------------------------------------------------------------- */
int psvae_segment_sum_fwd(int B, int W, const float *x, int ldx, const int32_t *seg_begin,
                          const int32_t *seg_end, float *out, int ldo, void *stream);
int psvae_segment_sum_bwd(int B, int W, const float *g_out, int ldgo, const int32_t *seg_begin,
                          const int32_t *seg_end, float *g_x, int ldgx, void *stream);

/* ------------------------------------------------------------------------------------
 * Reparameterised latent (src/modules/vae_piece_decoder.py:56-63) after the two Linears:
 *     lv = -|lv_pre|;  z = mean + exp(lv/2) * eps;
 *     kl = -0.5 * sum(1 + lv - mean^2 - exp(lv)) / B
 * eps is supplied by the caller (the reference draws it with torch.randn_like).  The KL sum is
 * taken in two stages in a fixed order through `workspace` (psvae_rsample_kl_workspace_bytes; 0 for
 * a batch small enough for one workgroup).
 * Backward takes g_z [B,L] and the scalar g_kl (device pointer).
 * ---------------------------------------------------------------------------------- */
size_t psvae_rsample_kl_workspace_bytes(int B, int L);
int psvae_rsample_kl_fwd(int B, int L, const float *mean, const float *lv_pre, const float *eps,
                         float *z, float *kl, void *workspace, size_t workspace_bytes, void *stream);
int psvae_rsample_kl_bwd(int B, int L, const float *mean, const float *lv_pre, const float *eps,
                         const float *g_z, const float *g_kl, float *g_mean, float *g_lv_pre,
                         void *stream);

/* ------------------------------------------------------------------------------------
 * One time step of the piece language model's GRU (nn.GRU at
 * src/modules/vae_piece_decoder.py:29,89; PyTorch gate order r, z, n), element-wise part:
 * gi = x_t W_ih^T + b_ih and gh = h_prev W_hh^T + b_hh ([B,3H], computed by psvae_gemm_f32),
 *     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r * gh_n),
 *     h = (1 - z) * n + z * h_prev;   gates [B,3H] keeps (r, z, n) for the backward.
 * Backward: given dh (+ optional dh_extra) = dL/dh, writes dgi, dgh [B,3H] (gradients of the
 * two pre-activation inputs) and dh_prev = dh * z (the direct path; the caller adds dgh W_hh).
 * ---------------------------------------------------------------------------------- */
int psvae_gru_cell_fwd(int B, int H, const float *gi, const float *gh, const float *h_prev,
                       float *h, float *gates, void *stream);
int psvae_gru_cell_bwd(int B, int H, const float *gates, const float *gh, const float *h_prev,
                       const float *dh, const float *dh_extra, float *dgi, float *dgh,
                       float *dh_prev, void *stream);

/* The whole recurrence of that GRU in one launch per direction (teacher-forced training,
 * src/modules/vae_piece_decoder.py:85-89): a workgroup owns 16 batch rows and walks all T steps.
 * gi [T,B,3H] = x_t W_ih^T + b_ih for every step (one psvae_gemm_f32 call), h0 [B,H],
 * w_hh [3H,H] (16-byte aligned), b_hh [3H] or NULL.  Writes hall [T+1,B,H] (hall[0] = h0) and,
 * unless NULL, what the backward needs: gates [T,B,3H] = (r, z, n) and ghn [T,B,H] = W_hn h + b_hn.
 * Backward: g_out [T,B,H] = dL/d hall[1..T], w_hh_t [H,3H] = W_hh transposed; writes
 * dgi, dgh [T,B,3H] (gradients of the two pre-activation inputs of every step; the weight and
 * bias gradients are products over them) and dh0 [B,H].  H % 4 == 0 and H <= PSVAE_GRU_SEQ_MAX_H
 * (PSVAE_ETOOLARGE otherwise: use the per-step kernels above). */
#define PSVAE_GRU_SEQ_MAX_H 256
int psvae_gru_seq_fwd(int T, int B, int H, const float *gi, const float *h0, const float *w_hh,
                      const float *b_hh, float *hall, float *gates, float *ghn, void *stream);
int psvae_gru_seq_bwd(int T, int B, int H, const float *gates, const float *ghn, const float *hall,
                      const float *g_out, const float *w_hh_t, float *dgi, float *dgh, float *dh0,
                      void *stream);

/* ------------------------------------------------------------------------------------
 * Link predictor input (src/modules/vae_piece_decoder.py:99-109) without the [B,N,N,*]
 * intermediates: feats[p] = [ h[src[p]] | h[dst[p]] | z[graph[p]] ], width 2H + L.
 * Backward sums the rows back per node / per graph without atomics: graph b owns pairs
 * pair_ptr[b] .. pair_ptr[b+1]-1 and nodes node_begin[b] .. node_end[b]-1.
 * g_h rows outside every graph's node range are left untouched.
 * ---------------------------------------------------------------------------------- */
int psvae_pair_gather_fwd(int P, int H, int L, const float *h, int ldh, const float *z, int ldz,
                          const int32_t *src, const int32_t *dst, const int32_t *graph,
                          float *feats, void *stream);
int psvae_pair_gather_bwd(int B, int H, int L, const float *g_feats,
                          const int32_t *src, const int32_t *dst, const int32_t *pair_ptr,
                          const int32_t *node_begin, const int32_t *node_end,
                          float *g_h, int ldgh, float *g_z, int ldgz, void *stream);

/* First link-predictor layer on a split weight, for inference over ALL candidate pairs
 * (src/modules/vae_piece_decoder.py:224-233): with W1 = [Wa | Wb | Wc] cut at the blocks of
 * [h_src | h_dst | z], W1 feats[p] + b1 = (h Wa^T)[src[p]] + (h Wb^T)[dst[p]] + (z Wc^T + b1)[graph[p]],
 * so the three products run once per node / per graph (psvae_gemm_f32) and this kernel only
 * gathers, adds and applies `act`: out[p] = act(a[src[p]] + b[dst[p]] + c[graph[p]]), rows of F
 * floats (F and every row stride multiples of 4). */
int psvae_pair_combine_fwd(int P, int F, const float *a, int lda, const float *b, int ldb,
                           const float *c, int ldc, const int32_t *src, const int32_t *dst,
                           const int32_t *graph, int act, float *out, int ldo, void *stream);
/* The same rows with one workgroup per graph that stages the graph's a / b rows in LDS (pairs of graph g are
 * pair_ptr[g] .. pair_ptr[g+1], its nodes node_begin[g] .. node_end[g]; src / dst hold global node ids as above and
 * the graph row of c is g): every node row is read once, which pays when a graph has many more pairs than nodes -
 * all candidate pairs of generated molecules (src/modules/vae_piece_decoder.py:224-238).  max_nodes sizes the
 * staging area; a graph with more nodes is served from memory, with the same result. */
int psvae_pair_combine_fwd_graphs(int B, int F, int max_nodes, const float *a, int lda, const float *b, int ldb,
                                  const float *c, int ldc, const int32_t *src, const int32_t *dst,
                                  const int32_t *pair_ptr, const int32_t *node_begin, const int32_t *node_end,
                                  int act, float *out, int ldo, void *stream);
/* Its backward for a = ab[:, :F], b = ab[:, F:] (one [V, 2F] matrix, as (h [Wa;Wb]^T) produces it): with
 * m[p] = g_out[p] gated by out[p] > 0 when act is ReLU (`out` = the forward's result),
 * g_ab[v][:F] = sum of m[p] over pairs with src[p] == v, g_ab[v][F:] = the same over dst[p] == v,
 * g_c[b] = sum of m[p] over the pairs of graph b — every sum in pair order, no atomics (the training
 * form of the link predictor's first layer, src/modules/vae_piece_decoder.py:99-110: the [P, 2H+L]
 * feature rows and their gradient are never formed).  Pairs of graph b are pair_ptr[b] ..
 * pair_ptr[b+1]-1 and refer to nodes node_begin[b] .. node_end[b]-1, whose g_ab rows are all written. */
int psvae_pair_combine_bwd(int B, int F, const float *g_out, int ldg, const float *out, int ldo, int act,
                           const int32_t *src, const int32_t *dst, const int32_t *pair_ptr,
                           const int32_t *node_begin, const int32_t *node_end,
                           float *g_ab, int ldab, float *g_c, int ldc, void *stream);

/* ------------------------------------------------------------------------------------
 * The scalar tail of the step and the block table of the atom features, one launch each (at batch 32 the
 * torch expressions they stand for are five to twelve launches of a step that is a chain of ~200).
 *
 * psvae_block_table: the three embedding tables of embed_atom (src/modules/vae_piece_decoder.py:65-69:
 *   atom [na,da], piece [np,dp], position [nq,dq]) as one matrix Z [na+np+nq, da+dp+dq+4], table k in its own
 *   column block, zeros elsewhere and a one in column da+dp+dq of the atom rows, so that
 *   Z . [W | b | 0]^T is node_trans (src/modules/encoder.py:54) of every table row with the bias on the atom rows.
 * psvae_mse_selected_fwd/bwd: nn.MSELoss() between the predictor's output pred [B,k] and the selected property
 *   columns props[:, sel] (src/pl_models/ps_vae_model.py:55-57): loss = mean (pred - props[:, sel])^2 with the
 *   differences kept in diff [B*k]; backward g_pred = g * 2 diff / (B k), g a device scalar.  sel: int64 [k].
 * psvae_loss_mix_fwd/bwd: weighted_loss (src/pl_models/ps_vae_model.py:71-74):
 *   out = alpha rec + (1 - alpha) pred + beta kl on device scalars, every product and sum rounded on its own;
 *   beta comes from beta_dev when that is not NULL (a recorded step reads it from device memory).  Backward
 *   writes g3 = g * (alpha, 1 - alpha, beta).
 * ---------------------------------------------------------------------------------- */
int psvae_block_table(int na, int da, int np, int dp, int nq, int dq, const float *ea, const float *ep,
                      const float *eq, float *Z, void *stream);
int psvae_mse_selected_fwd(int B, int k, const float *pred, int ld_pred, const float *props, int ld_props,
                           const int64_t *sel, float *diff, float *loss, void *stream);
int psvae_mse_selected_bwd(int n, const float *diff, const float *g, float *g_pred, void *stream);
int psvae_loss_mix_fwd(const float *rec, const float *pred, const float *kl, float alpha, float one_minus_alpha,
                       float beta, const float *beta_dev, float *out, void *stream);
int psvae_loss_mix_bwd(const float *g, float alpha, float one_minus_alpha, float beta, const float *beta_dev,
                       float *g3, void *stream);

/* ------------------------------------------------------------------------------------
 * Two Linear layers with no activation between them as one affine map — the link
 * predictor's MLP output layer followed by the class layer
 * (src/modules/vae_piece_decoder.py:42-46, 110):
 *     w_eff [C,K] = w_last [C,J] . w_out [J,K]      b_eff [C] = w_last . b_out + b_last
 * Backward returns the gradients of both layers from those of (w_eff, b_eff); the gradient
 * of b_last is g_b_eff itself.  C <= 8; biases may be NULL (g_b_out NULL with b_out).
 * ---------------------------------------------------------------------------------- */
int psvae_affine_merge_fwd(int C, int J, int K, const float *w_last, const float *b_last,
                           const float *w_out, const float *b_out, float *w_eff, float *b_eff,
                           void *stream);
int psvae_affine_merge_bwd(int C, int J, int K, const float *w_last, const float *w_out,
                           const float *b_out, const float *g_w_eff, const float *g_b_eff,
                           float *g_w_last, float *g_w_out, float *g_b_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Last Linear of edge_predictor fused with the link loss
 * (src/modules/vae_piece_decoder.py:45,54,110,115): logits = hid W^T + b  [P,C], C <= 8;
 * loss = mean_p( logsumexp(logits[p]) - logits[p][golden[p]] );
 * dlogits[p] = (softmax(logits[p]) - onehot(golden[p])) / P;
 * n_correct = #{p : argmax(logits[p]) == golden[p]} (first maximum, as torch.argmax).
 * P == 0 gives loss = NaN like torch's mean over an empty batch.
 * live_count (optional, device int32): only rows 0 .. *live_count - 1 are pairs, the rest of the P
 * rows is padding of a fixed-size buffer (zero logits, zero dlogits, no part in the mean or in
 * n_correct) - the count is read on the device when the kernel runs.
 * psvae_link_ce_bwd_input: g_hid[p][f] = g_loss * sum_c dlogits[p][c] * W[c][f].
 * ---------------------------------------------------------------------------------- */
size_t psvae_link_ce_workspace_bytes(int P);
int psvae_link_ce_fwd(int P, int F, int C, const float *hid, int ldhid, const float *W,
                      const float *b, const int32_t *golden, const int32_t *live_count,
                      float *logits, float *dlogits, float *loss, int32_t *n_correct,
                      void *workspace, size_t workspace_bytes, void *stream);
int psvae_link_ce_bwd_input(int P, int F, int C, const float *dlogits, const float *W,
                            const float *g_loss, float *g_hid, int ldg, void *stream);

/* ------------------------------------------------------------------------------------
 * Cross-entropy of the piece language model with an ignored label
 * (nn.CrossEntropyLoss(ignore_index=pad): src/modules/vae_piece_decoder.py:53,113-114):
 * logits [R,V] rows of stride ld, gold [R]; over the rows with gold != ignore_index
 *     loss[0] = mean( logsumexp(logits[r]) - logits[r][gold[r]] ),  loss[1] = 1 / #rows
 * (no such row: NaN, as torch's mean over nothing); dlogits [R,V] (rows of stride ldd >= V) =
 * softmax - onehot, zero rows where ignored, and zeros in the columns V..ldd-1 of every row (a
 * vocabulary padded to a multiple of 4 so that the products on either side take the 16-byte
 * loaders): times g_loss * loss[1] it is the gradient; counts[0] = #rows,
 * counts[1] = #{r : first argmax == gold[r]} (src/modules/vae_piece_decoder.py:120-122).
 * Sums run in a fixed order.  (Rows wider than 1024 columns are read three times instead of once.)
 * ---------------------------------------------------------------------------------- */
size_t psvae_token_ce_workspace_bytes(int R);
int psvae_token_ce_fwd(int R, int V, const float *logits, int64_t ld, const int32_t *gold, int ignore_index,
                       float *dlogits, int64_t ldd, float *loss, int32_t *counts,
                       void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * One step of the piece sampling loop of generation, after the vocabulary projection
 * (src/modules/vae_piece_decoder.py:139-158): per molecule b that is still active, logits[b][pad] and
 * (first_step) logits[b][end] count as -inf, probs = softmax(logits[b] / temperature) and
 *   next = min(#{i : cdf[i] <= uniforms[b] * cdf[last]}, last),  cdf = running sum of probs in index
 *   order, last = largest index of positive probability
 * (the inverse-CDF form of torch.multinomial(probs, 1) from a supplied uniform number); then
 * out[b] = next, length[b] += 1, atoms[b] += piece_atoms[next], and the molecule becomes inactive when
 * next == end, atoms[b] > max_atom_num or length[b] == max_pos; cur[b] = next while it stays active,
 * end otherwise.  An inactive molecule writes out[b] = pad, cur[b] = end and keeps its counters.
 * logits [B, V] rows ld_logits apart; uniforms and out are columns of [B, *] arrays (element b at
 * b * ld_uniforms / b * ld_out); active [B] bytes, atoms / length [B] int32, out / cur int64.
 * ---------------------------------------------------------------------------------- */
int psvae_piece_draw(int B, int V, const float *logits, int64_t ld_logits, float temperature, const float *uniforms,
                     int64_t ld_uniforms, int pad_idx, int end_idx, int first_step, const int32_t *piece_atoms,
                     int max_atom_num, int max_pos, uint8_t *active, int32_t *atoms, int32_t *length,
                     int64_t *out, int64_t ld_out, int64_t *cur, void *stream);

/* ------------------------------------------------------------------------------------
 * Candidate atom pairs of generated molecules (src/modules/vae_piece_decoder.py:213-222): for molecule b
 * (atoms node_off[b] .. node_off[b+1], fragment id frag[i] per atom) the pairs (m, k), m < k, of atoms in
 * different fragments, in row-major order.  _count writes the number per molecule; after
 * ptr = exclusive scan of counts, _fill writes for candidate ptr[b] + j the global atom ids src / dst, the
 * molecule id graph and the molecule-local ids m / k.  max_atoms: at least the largest molecule (it sizes the
 * LDS stage; at most 15 360) - a molecule with more atoms is treated as its first max_atoms atoms, by both calls.
 * ---------------------------------------------------------------------------------- */
int psvae_pair_enumerate_count(int B, int max_atoms, const int32_t *node_off, const int32_t *frag, int32_t *counts,
                               void *stream);
int psvae_pair_enumerate_fill(int B, int max_atoms, const int32_t *node_off, const int32_t *frag, const int32_t *ptr,
                              int32_t *src, int32_t *dst, int32_t *graph, int32_t *m, int32_t *k, void *stream);

/* ------------------------------------------------------------------------------------
 * Inference link prediction (src/modules/vae_piece_decoder.py:233-240):
 * prob = softmax(logits[p]); conf[p] = max prob, etype[p] = argmax (first maximum);
 * keep[p] = conf[p] >= add_edge_th && etype[p] != none_bond.
 * ---------------------------------------------------------------------------------- */
int psvae_link_infer(int P, int C, const float *logits, float add_edge_th, int none_bond,
                     float *conf, int32_t *etype, uint8_t *keep, void *stream);

/* ------------------------------------------------------------------------------------
 * First encoder layer on embedding ids.  node_trans (src/modules/encoder.py:54) applied to
 * embed_atom's concatenated rows (src/modules/vae_piece_decoder.py:65-69) equals
 * T[x] + T[n_atom + xp] + T[n_atom + n_piece + xq], with T [n_atom + n_piece + n_pos, H] =
 * the three tables multiplied by the matching column blocks of the weight (bias folded into the
 * atom rows by the caller).  Writes out [V,H] (row stride ld_out, H and ld_out multiples of 4);
 * ids outside their table are clamped into it.  The gradient of T is the sum of the rows of g per
 * table row: psvae_key_sort over the three id columns (once per batch) + psvae_keyed_row_sum.
 * ---------------------------------------------------------------------------------- */
int psvae_embed_trans_fwd(int V, int H, const int32_t *x, const int32_t *xp, const int32_t *xq,
                          int n_atom, int n_piece, int n_pos, const float *T, float *out,
                          int ld_out, void *stream);

/* ------------------------------------------------------------------------------------
 * Greedy bond assembly of B generated molecules (src/modules/vae_piece_decoder.py:245-270 with
 * valence_check / cycle_check of src/utils/chem_utils.py:447-483), one lane per molecule.
 * Flat per-atom arrays [V] (molecule b owns node_off[b] .. node_off[b+1]): atom_type (chemical
 * vocabulary index), block0 (fragment index of the atom inside its molecule, 0 .. n_blocks[b]-1).
 * row_ptr/col/cls: existing (in-fragment) bonds as a symmetric CSR over the flat atom ids, cls =
 * bond order 1..3.  Candidates of molecule b are entries order_ptr[b] .. order_ptr[b+1] of
 * ord_m / ord_n (molecule-local atom ids) and ord_type (bond class), already in visiting order.
 * max_valence[n_atom_type]; sulfur_idx = index of 'S' (the 2-or-6 rule, :479-481).
 * Out: accepted[k] (bond k was added), block[V] (final connected block of every atom),
 * keep_atom[V] (atom belongs to the largest block), status[B] (0 ok, 1 degree overflow,
 * 2 bad ids).
 * ---------------------------------------------------------------------------------- */
size_t psvae_assemble_links_workspace_bytes(int V);
int psvae_assemble_links(int B, int V, const int32_t *node_off, const int32_t *atom_type,
                         const int32_t *block0, const int32_t *n_blocks,
                         const int32_t *row_ptr, const int32_t *col, const uint8_t *cls,
                         const int32_t *order_ptr, const int32_t *ord_m, const int32_t *ord_n,
                         const int32_t *ord_type, const int32_t *max_valence, int n_atom_type,
                         int sulfur_idx, uint8_t *accepted, int32_t *block, uint8_t *keep_atom,
                         int32_t *status, void *workspace, size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------
 * Optimiser tail on flat buffers of n floats (all parameters / gradients / Adam moments of the
 * model laid out back to back): g *= grad_scale (1 / world size after the gradient all-reduce),
 * global-norm clipping as torch.nn.utils.clip_grad_norm_ (Trainer gradient_clip_val,
 * src/train.py:47; max_norm <= 0 disables it) and one torch.optim.Adam step
 * (src/pl_models/ps_vae_model.py:123-127; step counts from 1, weight_decay is Adam's L2 form).
 * write_grad != 0 leaves the scaled and clipped gradient in g, as clip_grad_norm_ does.
 * norm_out (optional, device) receives the global gradient norm before clipping.
 * Deterministic: fixed-order reductions, no atomics.  Buffers 16-byte aligned.
 * ---------------------------------------------------------------------------------- */
size_t psvae_clip_adam_workspace_bytes(long long n);
int psvae_clip_adam_step(long long n, float *p, float *g, float *m, float *v, float grad_scale,
                         float max_norm, double lr, double beta1, double beta2, double eps,
                         double weight_decay, int step, int write_grad, float *norm_out,
                         void *workspace, size_t workspace_bytes, void *stream);

/* The same step with the two step-dependent scalars read from DEVICE memory when the kernel runs:
 * step_scalars[0] = lr / (1 - beta1^t), step_scalars[1] = sqrt(1 - beta2^t) (the caller evaluates
 * them in double and stores them as float).  For a step recorded once in a HIP graph and replayed:
 * nothing that changes from step to step is a kernel argument. */
int psvae_clip_adam_step_dev(long long n, float *p, float *g, float *m, float *v, float grad_scale,
                             float max_norm, double beta1, double beta2, double eps, double weight_decay,
                             const float *step_scalars, int write_grad, float *norm_out,
                             void *workspace, size_t workspace_bytes, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* PSVAE_HIP_H */
