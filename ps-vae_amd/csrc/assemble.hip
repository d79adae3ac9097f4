// Greedy bond assembly of generated molecules, one lane per molecule.
//
// Restates the part of VAEPieceDecoder.inference that needs only adjacency
// (src/modules/vae_piece_decoder.py:245-270): candidate bonds are visited in the given order
// (confidence descending); a bond is added when valence_check
// (src/utils/chem_utils.py:471-483) and cycle_check (:447-468: the new bond may close no ring,
// or one of 5 or 6 atoms) allow it; connected blocks are merged as bonds are added (:257-263)
// and the atoms outside the largest block are marked for removal (:267-268).  The reference
// walks one molecule at a time in Python (src/pl_models/ps_vae_model.py:139-141); molecules
// are independent, so here each is one lane's serial walk over its own slice of the flat
// arrays.  SanitizeMol / Kekulize / SMILES output stay with RDKit.
#include "common.h"

namespace {

constexpr int MAX_DEG = 8;   // valence <= 6 bounds the degree; 8 leaves slack

struct AsmArgs {
    int B;
    const int32_t *node_off, *atom_type, *block0, *n_blocks;
    const int32_t *row_ptr, *col;
    const uint8_t *cls;
    const int32_t *order_ptr, *ord_m, *ord_n, *ord_type;
    const int32_t *max_valence;
    int n_atom_type, sulfur;
    uint8_t *accepted;
    int32_t *block;
    uint8_t *keep_atom;
    int32_t *status;   // [B]: 0 ok, 1 degree overflow, 2 bad atom / block id
    int32_t *adj, *deg, *val, *cnt, *queue, *dist, *mark;
};

__global__ void assemble_kernel(AsmArgs a) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= a.B) return;
    const int a0 = a.node_off[b], n = a.node_off[b + 1] - a0;
    const int nb = a.n_blocks[b];
    int32_t *adj = a.adj + (long long)a0 * MAX_DEG, *deg = a.deg + a0, *val = a.val + a0, *cnt = a.cnt + a0;
    int32_t *queue = a.queue + a0, *dist = a.dist + a0, *mark = a.mark + a0, *block = a.block + a0;
    int status = 0;
    for (int i = 0; i < n; ++i) cnt[i] = 0, mark[i] = 0;
    for (int i = 0; i < n; ++i) {
        int d = 0, v = 0;
        for (int e = a.row_ptr[a0 + i]; e < a.row_ptr[a0 + i + 1]; ++e) {
            if (d < MAX_DEG) adj[i * MAX_DEG + d++] = a.col[e] - a0; else status = 1;
            v += a.cls[e];   // bond class = bond order (None never appears as an existing bond)
        }
        deg[i] = d, val[i] = v;
        const int bl = a.block0[a0 + i], ty = a.atom_type[a0 + i];
        if (bl < 0 || bl >= nb || bl >= n || ty < 0 || ty >= a.n_atom_type) {
            status = 2;
            block[i] = 0;
        } else {
            block[i] = bl;
            cnt[bl] += 1;
        }
    }
    if (status == 2) {
        for (int k = a.order_ptr[b]; k < a.order_ptr[b + 1]; ++k) a.accepted[k] = 0;
        for (int i = 0; i < n; ++i) a.keep_atom[a0 + i] = 0;
        a.status[b] = status;
        return;
    }
    int stamp = 0;
    for (int k = a.order_ptr[b]; k < a.order_ptr[b + 1]; ++k) {
        const int i = a.ord_m[k], j = a.ord_n[k], t = a.ord_type[k];
        bool ok = i >= 0 && j >= 0 && i < n && j < n && i != j && t >= 1 && t <= 3;   // bond_idx_to_valence == -1 otherwise
        if (ok) {   // valence_check
            const int t1 = a.atom_type[a0 + i], t2 = a.atom_type[a0 + j];
            if ((t1 == a.sulfur && val[i] == 2) || (t2 == a.sulfur && val[j] == 2)) ok = false;
            else ok = val[i] + t <= a.max_valence[t1] && val[j] + t <= a.max_valence[t2];
        }
        if (ok) {   // cycle_check: atoms on the shortest existing path i .. j (None, 5 or 6 pass)
            ++stamp;
            int head = 0, tail = 0, found = 0;
            queue[tail++] = i, dist[i] = 1, mark[i] = stamp;
            while (head < tail && !found) {
                const int u = queue[head++];
                for (int q = 0; q < deg[u]; ++q) {
                    const int w = adj[u * MAX_DEG + q];
                    if (w == j) {
                        found = dist[u] + 1;
                        break;
                    }
                    if (mark[w] != stamp) mark[w] = stamp, dist[w] = dist[u] + 1, queue[tail++] = w;
                }
            }
            ok = found == 0 || found == 5 || found == 6;
        }
        if (ok && (deg[i] >= MAX_DEG || deg[j] >= MAX_DEG)) ok = false, status = 1;
        a.accepted[k] = ok ? 1 : 0;
        if (!ok) continue;
        adj[i * MAX_DEG + deg[i]++] = j, adj[j * MAX_DEG + deg[j]++] = i;
        val[i] += t, val[j] += t;
        const int b1 = block[i], b2 = block[j];
        if (b1 != b2) {
            for (int q = 0; q < n; ++q)
                if (block[q] == b1) block[q] = b2;
            cnt[b2] += cnt[b1];   // (the reference leaves cnt[b1] as it was; it can no longer be the maximum)
        }
    }
    int best = 0;
    for (int q = 1; q < nb; ++q)
        if (cnt[q] > cnt[best]) best = q;   // Python's max(): the first maximum
    for (int i = 0; i < n; ++i) a.keep_atom[a0 + i] = block[i] == best ? 1 : 0;
    a.status[b] = status;
}

}  // namespace

extern "C" size_t psvae_assemble_links_workspace_bytes(int V) {
    return V <= 0 ? 0 : (size_t)V * (MAX_DEG + 6) * sizeof(int32_t);
}

extern "C" int psvae_assemble_links(int B, int V, const int32_t *node_off, const int32_t *atom_type,
                                    const int32_t *block0, const int32_t *n_blocks, const int32_t *row_ptr,
                                    const int32_t *col, const uint8_t *cls, const int32_t *order_ptr,
                                    const int32_t *ord_m, const int32_t *ord_n, const int32_t *ord_type,
                                    const int32_t *max_valence, int n_atom_type, int sulfur_idx, uint8_t *accepted,
                                    int32_t *block, uint8_t *keep_atom, int32_t *status, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(B >= 0 && V >= 0 && n_atom_type > 0, PSVAE_EINVAL, "psvae_assemble_links: bad sizes");
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(node_off && n_blocks && row_ptr && order_ptr && max_valence && status, PSVAE_EINVAL,
                  "psvae_assemble_links: null pointer");
    PSVAE_REQUIRE(V == 0 || (atom_type && block0 && block && keep_atom), PSVAE_EINVAL,
                  "psvae_assemble_links: null per-atom pointer");
    PSVAE_REQUIRE(workspace_bytes >= psvae_assemble_links_workspace_bytes(V) && (V == 0 || workspace), PSVAE_EWORKSPACE,
                  "psvae_assemble_links: workspace too small");
    AsmArgs a;
    a.B = B, a.node_off = node_off, a.atom_type = atom_type, a.block0 = block0, a.n_blocks = n_blocks;
    a.row_ptr = row_ptr, a.col = col, a.cls = cls;
    a.order_ptr = order_ptr, a.ord_m = ord_m, a.ord_n = ord_n, a.ord_type = ord_type;
    a.max_valence = max_valence, a.n_atom_type = n_atom_type, a.sulfur = sulfur_idx;
    a.accepted = accepted, a.block = block, a.keep_atom = keep_atom, a.status = status;
    int32_t *w = (int32_t *)workspace;
    a.adj = w, w += (size_t)V * MAX_DEG;
    a.deg = w, w += V;
    a.val = w, w += V;
    a.cnt = w, w += V;
    a.queue = w, w += V;
    a.dist = w, w += V;
    a.mark = w;
    hipLaunchKernelGGL(assemble_kernel, dim3(psvae_cdiv(B, 64)), dim3(64), 0, (hipStream_t)stream, a);
    PSVAE_CHECK_LAUNCH("psvae_assemble_links");
    return PSVAE_OK;
}
