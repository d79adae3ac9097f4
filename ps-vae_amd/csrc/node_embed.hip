// First layer of the GINE encoders on embedding ids instead of embedded rows.
//
// The reference embeds every atom as [E_atom[x] | E_piece[xp] | E_pos[xq]]
// (src/modules/vae_piece_decoder.py:65-69) and both encoders begin with node_trans, a Linear
// over that row (src/modules/encoder.py:54).  A Linear over a concatenation of table rows is a
// sum of rows of the tables multiplied by the matching column blocks of the weight:
//     node_trans(embed(i)) = T[x_i] + T[n_atom + xp_i] + T[n_atom + n_piece + xq_i] + b,
//     T = [E_atom W_a^T ; E_piece W_p^T ; E_pos W_q^T]          (R = 363 rows for the defaults)
// so the [V, 200] x [200, 300] product and the [V, 200] embedded rows disappear: the forward is
// a three-row gather-sum (HBM-bound: V*H*4 bytes written).  The gradient of T is the sum of
// gradient rows by key; the host side takes it as indicator^T g on the matrix cores (one
// product in place of two), because LDS float atomics measured 3-8 cycles per LANE on gfx950
// (a private-table kernel took 0.45-1.2 ms for 95k rows where the product takes 0.3 ms).
#include "common.h"

namespace {


__device__ __forceinline__ int clamp_id(int v, int n) { return v < 0 ? 0 : (v >= n ? n - 1 : v); }

__global__ void embed_trans_fwd_kernel(int V, int H4, const int32_t *__restrict__ x, const int32_t *__restrict__ xp,
                                       const int32_t *__restrict__ xq, int n_atom, int n_piece, int n_pos,
                                       const float4 *__restrict__ T, float4 *__restrict__ out, int ld4) {
    const long long total = (long long)V * H4;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int v = (int)(i / H4), c = (int)(i - (long long)v * H4);
        const int ka = clamp_id(x[v], n_atom), kp = n_atom + clamp_id(xp[v], n_piece);
        const int kq = n_atom + n_piece + clamp_id(xq[v], n_pos);
        const float4 a = T[(long long)ka * H4 + c], p = T[(long long)kp * H4 + c], q = T[(long long)kq * H4 + c];
        float4 r;
        r.x = (a.x + p.x) + q.x, r.y = (a.y + p.y) + q.y, r.z = (a.z + p.z) + q.z, r.w = (a.w + p.w) + q.w;
        out[(long long)v * ld4 + c] = r;
    }
}

int check_ids(const char *fn, int V, int H, int n_atom, int n_piece, int n_pos, const void *x, const void *xp,
              const void *xq) {
    PSVAE_REQUIRE(V >= 0 && H > 0 && n_atom > 0 && n_piece > 0 && n_pos > 0, PSVAE_EINVAL, "%s: bad sizes", fn);
    PSVAE_REQUIRE(V == 0 || (x && xp && xq), PSVAE_EINVAL, "%s: null id array", fn);
    return PSVAE_OK;
}

}  // namespace

extern "C" int psvae_embed_trans_fwd(int V, int H, const int32_t *x, const int32_t *xp, const int32_t *xq, int n_atom,
                                     int n_piece, int n_pos, const float *T, float *out, int ld_out, void *stream) {
    const int rc = check_ids("psvae_embed_trans_fwd", V, H, n_atom, n_piece, n_pos, x, xp, xq);
    if (rc != PSVAE_OK) return rc;
    if (V == 0) return PSVAE_OK;
    PSVAE_REQUIRE(H % 4 == 0 && ld_out % 4 == 0 && ld_out >= H, PSVAE_EINVAL,
                  "psvae_embed_trans_fwd: H=%d and ld_out=%d must be multiples of 4, ld_out >= H", H, ld_out);
    PSVAE_REQUIRE(T && out && psvae_aligned16(T) && psvae_aligned16(out), PSVAE_EINVAL,
                  "psvae_embed_trans_fwd: null or misaligned table / output");
    const long long total = (long long)V * (H / 4);
    const long long want = (total + 255) / 256;
    const int blocks = (int)(want < 16384 ? want : 16384);
    hipLaunchKernelGGL(embed_trans_fwd_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, V, H / 4, x, xp, xq,
                       n_atom, n_piece, n_pos, (const float4 *)T, (float4 *)out, ld_out / 4);
    PSVAE_CHECK_LAUNCH("psvae_embed_trans_fwd");
    return PSVAE_OK;
}
