// The piece language model's GRU recurrence (nn.GRU at src/modules/vae_piece_decoder.py:29,
// teacher-forced at :85-89, sampled at :139-158) as ONE launch per direction.
//
// The recurrence couples the hidden units of a molecule, not the molecules of a batch: a
// workgroup owns 16 batch rows and walks all T time steps for them, so nothing crosses
// workgroups and no grid-wide synchronisation exists.  Per step it computes
//     gh = h W_hh^T + b_hh           (16 x 3H, v_mfma_f32_16x16x4_f32: exact f32)
//     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
// (PyTorch gate order r, z, n; gi = x W_ih^T + b_ih comes from one large product beforehand).
// Hidden units are cut into blocks of 16; wave w owns blocks w, w + 4, ... and, for each, the
// three gate tiles of those units, so the gate arithmetic runs on the accumulators where they
// are (D layout: lane = unit, registers = 4 batch rows) and the new hidden state goes straight
// back to LDS as the next step's A operand.  h lives in LDS k-major ([unit][row], row stride 20
// floats: the four k groups of a wave read disjoint banks); W_hh (3H x H, L2-resident) is read
// 16 bytes per lane along k.  Within a 16-deep k block lane group q takes k = 16b + 4q + e in
// MFMA number e - for A and B alike - so the sum over k is complete whatever the order.
//
// Backward mirrors it: dgh (16 x 3H, k-major in LDS) times W_hh gives the gradient that flows to
// the previous step, the gate derivatives are evaluated in the same register layout, and the
// weight gradients are left to two large products over the stacked dgi / dgh the kernel writes.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GRU_ROWS = 16;        // batch rows per workgroup
constexpr int GRU_LD = 20;          // LDS row stride of the k-major operand images
constexpr int GRU_THREADS = 256;

__device__ __forceinline__ float sigmoidf_(float x) { return 1.0f / (1.0f + expf(-x)); }

struct GruFwdArgs {
    int T, B, H;
    const float *gi;      // [T,B,3H]
    const float *h0;      // [B,H]
    const float *w_hh;    // [3H,H]
    const float *b_hh;    // [3H] or null
    float *hall;          // [T+1,B,H]
    float *gates;         // [T,B,3H] (r, z, n) or null (inference: nothing saved)
    float *ghn;           // [T,B,H]  W_hn h + b_hn, needed by the backward, or null
};

template <int JB>
__global__ __launch_bounds__(GRU_THREADS, 1) void gru_seq_fwd_kernel(GruFwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int H = a.H, B = a.B, H3 = 3 * a.H;
    const int KP = (H + 15) & ~15, nblk = KP >> 4;
    float *hk[2] = {lds, lds + KP * GRU_LD};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * GRU_ROWS;

    // h0 -> stage 0 (k-major), zero rows of the padded k tail in both stages, hall[0] = h0
    for (int idx = tid; idx < KP * GRU_ROWS; idx += GRU_THREADS) {
        const int k = idx % KP, i = idx / KP;
        const int row = row0 + i;
        float v = 0.f;
        if (k < H && row < B) {
            v = a.h0[(long long)row * H + k];
            a.hall[(long long)row * H + k] = v;
        }
        hk[0][k * GRU_LD + i] = v;
        if (k >= H) hk[1][k * GRU_LD + i] = 0.f;
    }
    __syncthreads();

    for (int t = 0; t < a.T; ++t) {
        const float *hc = hk[t & 1];
        float *hn = hk[(t + 1) & 1];
        const float *gi_t = a.gi + (long long)t * B * H3;
        // input projections of this step, requested before the products they do not depend on
        float gi_r[JB][4], gi_z[JB][4], gi_n[JB][4];
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int j = (wave + 4 * q) * 16 + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                const bool ok = (wave + 4 * q) < nblk && j < H && row < B;
                const long long o = ok ? (long long)row * H3 + j : 0;
                gi_r[q][r] = gi_t[o], gi_z[q][r] = gi_t[o + (ok ? H : 0)], gi_n[q][r] = gi_t[o + (ok ? 2 * H : 0)];
            }
        }
        f32x4 acc[JB][3];
#pragma unroll
        for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // B operand rows: gate c, unit j -> row c*H + j of W_hh (clamped for the padded units, whose
        // columns are never stored); 16 bytes along k per lane
        const float *wrow[JB][3];
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int j = min((wave + 4 * q) * 16 + jl, H - 1);
#pragma unroll
            for (int c = 0; c < 3; ++c) wrow[q][c] = a.w_hh + (long long)(c * H + j) * H;
        }
        float4 wcur[JB][3], wnxt[JB][3];
        auto load_w = [&](float4 (&w)[JB][3], int kb) {
            const int k = min(16 * kb + 4 * g, H - 4);   // the padded k tail multiplies zero rows of h
#pragma unroll
            for (int q = 0; q < JB; ++q)
                if (wave + 4 * q < nblk) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) w[q][c] = *reinterpret_cast<const float4 *>(wrow[q][c] + k);
                }
        };
        load_w(wcur, 0);
        for (int kb = 0; kb < nblk; ++kb) {
            if (kb + 1 < nblk) load_w(wnxt, kb + 1);
            float av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = hc[(16 * kb + 4 * g + e) * GRU_LD + jl];
#pragma unroll
            for (int q = 0; q < JB; ++q)
                if (wave + 4 * q < nblk) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const float4 w = wcur[q][c];
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w.x, acc[q][c], 0, 0, 0);
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w.y, acc[q][c], 0, 0, 0);
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w.z, acc[q][c], 0, 0, 0);
                        acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w.w, acc[q][c], 0, 0, 0);
                    }
                }
#pragma unroll
            for (int q = 0; q < JB; ++q)
#pragma unroll
                for (int c = 0; c < 3; ++c) wcur[q][c] = wnxt[q][c];
        }
        // gates on the accumulators: lane = unit j, registers = rows 4g .. 4g+3
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int blk = wave + 4 * q;
            const int j = blk * 16 + jl;
            if (blk >= nblk || j >= H) continue;
            const float br = a.b_hh ? a.b_hh[j] : 0.f, bz = a.b_hh ? a.b_hh[H + j] : 0.f;
            const float bn = a.b_hh ? a.b_hh[2 * H + j] : 0.f;
            const float4 hp4 = *reinterpret_cast<const float4 *>(hc + j * GRU_LD + 4 * g);
            const float hp[4] = {hp4.x, hp4.y, hp4.z, hp4.w};
            float hnew[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                const float ghr = acc[q][0][r] + br, ghz = acc[q][1][r] + bz, ghn = acc[q][2][r] + bn;
                const float rr = sigmoidf_(gi_r[q][r] + ghr);
                const float zz = sigmoidf_(gi_z[q][r] + ghz);
                const float nn = tanhf(gi_n[q][r] + rr * ghn);
                hnew[r] = (1.0f - zz) * nn + zz * hp[r];
                if (row < B) {
                    const long long o = ((long long)t * B + row) * H3 + j;
                    if (a.gates) a.gates[o] = rr, a.gates[o + H] = zz, a.gates[o + 2 * H] = nn;
                    if (a.ghn) a.ghn[((long long)t * B + row) * H + j] = ghn;
                    a.hall[((long long)(t + 1) * B + row) * H + j] = hnew[r];
                }
            }
            *reinterpret_cast<float4 *>(hn + j * GRU_LD + 4 * g) = make_float4(hnew[0], hnew[1], hnew[2], hnew[3]);
        }
        __syncthreads();
    }
}

struct GruBwdArgs {
    int T, B, H;
    const float *gates;   // [T,B,3H]
    const float *ghn;     // [T,B,H]
    const float *hall;    // [T+1,B,H]
    const float *g_out;   // [T,B,H]: dL/d hall[t+1]
    const float *w_hh_t;  // [H,3H] = W_hh transposed
    float *dgi;           // [T,B,3H]
    float *dgh;           // [T,B,3H]
    float *dh0;           // [B,H]
};

template <int JB>
__global__ __launch_bounds__(GRU_THREADS, 1) void gru_seq_bwd_kernel(GruBwdArgs a) {
    extern __shared__ __align__(16) float lds[];
    const int H = a.H, B = a.B, H3 = 3 * a.H;
    const int nblk = (H + 15) >> 4;
    const int K3P = (H3 + 15) & ~15, nkb = K3P >> 4;
    float *dk[2] = {lds, lds + K3P * GRU_LD};
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int jl = lane & 15, g = lane >> 4;
    const int row0 = blockIdx.x * GRU_ROWS;

    for (int idx = tid; idx < (K3P - H3) * GRU_ROWS; idx += GRU_THREADS) {   // padded k tail: zero in both stages
        const int k = H3 + idx / GRU_ROWS, i = idx % GRU_ROWS;
        dk[0][k * GRU_LD + i] = 0.f, dk[1][k * GRU_LD + i] = 0.f;
    }
    float dh[JB][4];
#pragma unroll
    for (int q = 0; q < JB; ++q)
#pragma unroll
        for (int r = 0; r < 4; ++r) dh[q][r] = 0.f;

    // saved values of one step in the register layout of the gate arithmetic
    struct Saved {
        float rr[JB][4], zz[JB][4], nn[JB][4], gn[JB][4], hp[JB][4], go[JB][4];
    };
    auto fetch = [&](Saved &s, int t) {
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int j = (wave + 4 * q) * 16 + jl;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                const bool ok = (wave + 4 * q) < nblk && j < H && row < B;
                const long long o3 = ok ? ((long long)t * B + row) * H3 + j : 0;
                const long long o1 = ok ? ((long long)t * B + row) * H + j : 0;
                s.rr[q][r] = a.gates[o3], s.zz[q][r] = a.gates[o3 + (ok ? H : 0)], s.nn[q][r] = a.gates[o3 + (ok ? 2 * H : 0)];
                s.gn[q][r] = a.ghn[o1], s.hp[q][r] = a.hall[o1], s.go[q][r] = a.g_out[o1];
            }
        }
    };
    Saved cur;
    if (a.T > 0) fetch(cur, a.T - 1);
    __syncthreads();

    for (int t = a.T - 1; t >= 0; --t) {
        float *dc = dk[t & 1];
        float dzh[JB][4];   // d * z: the part of dL/dh_{t-1} that does not pass through W_hh
#pragma unroll
        for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) dzh[q][r] = 0.f;
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int blk = wave + 4 * q;
            const int j = blk * 16 + jl;
            if (blk >= nblk || j >= H) continue;
            float vr[4], vz[4], vn[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = row0 + 4 * g + r;
                const float rr = cur.rr[q][r], zz = cur.zz[q][r], nn = cur.nn[q][r];
                const float d = cur.go[q][r] + dh[q][r];
                const float dn_pre = d * (1.0f - zz) * (1.0f - nn * nn);
                const float dz_pre = d * (cur.hp[q][r] - nn) * zz * (1.0f - zz);
                const float dr_pre = dn_pre * cur.gn[q][r] * rr * (1.0f - rr);
                dzh[q][r] = d * zz;
                vr[r] = dr_pre, vz[r] = dz_pre, vn[r] = dn_pre * rr;
                if (row < B) {
                    const long long o = ((long long)t * B + row) * H3 + j;
                    a.dgi[o] = dr_pre, a.dgi[o + H] = dz_pre, a.dgi[o + 2 * H] = dn_pre;
                    a.dgh[o] = dr_pre, a.dgh[o + H] = dz_pre, a.dgh[o + 2 * H] = vn[r];
                } else {
                    vr[r] = vz[r] = vn[r] = 0.f;
                }
            }
            *reinterpret_cast<float4 *>(dc + j * GRU_LD + 4 * g) = make_float4(vr[0], vr[1], vr[2], vr[3]);
            *reinterpret_cast<float4 *>(dc + (H + j) * GRU_LD + 4 * g) = make_float4(vz[0], vz[1], vz[2], vz[3]);
            *reinterpret_cast<float4 *>(dc + (2 * H + j) * GRU_LD + 4 * g) = make_float4(vn[0], vn[1], vn[2], vn[3]);
        }
        __syncthreads();
        if (t > 0) fetch(cur, t - 1);   // the next step's saved values travel while the product runs

        f32x4 acc[JB];
        const float *wrow[JB];
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
            wrow[q] = a.w_hh_t + (long long)min((wave + 4 * q) * 16 + jl, H - 1) * H3;
        }
        float4 wcur[JB], wnxt[JB];
        auto load_w = [&](float4 (&w)[JB], int kb) {
            const int k = min(16 * kb + 4 * g, H3 - 4);
#pragma unroll
            for (int q = 0; q < JB; ++q)
                if (wave + 4 * q < nblk) w[q] = *reinterpret_cast<const float4 *>(wrow[q] + k);
        };
        load_w(wcur, 0);
        for (int kb = 0; kb < nkb; ++kb) {
            if (kb + 1 < nkb) load_w(wnxt, kb + 1);
            float av[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) av[e] = dc[(16 * kb + 4 * g + e) * GRU_LD + jl];
#pragma unroll
            for (int q = 0; q < JB; ++q)
                if (wave + 4 * q < nblk) {
                    const float4 w = wcur[q];
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[0], w.x, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[1], w.y, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[2], w.z, acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[3], w.w, acc[q], 0, 0, 0);
                }
#pragma unroll
            for (int q = 0; q < JB; ++q) wcur[q] = wnxt[q];
        }
#pragma unroll
        for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) dh[q][r] = acc[q][r] + dzh[q][r];
    }
#pragma unroll
    for (int q = 0; q < JB; ++q) {
        const int j = (wave + 4 * q) * 16 + jl;
        if (wave + 4 * q >= nblk || j >= H) continue;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = row0 + 4 * g + r;
            if (row < B) a.dh0[(long long)row * H + j] = dh[q][r];
        }
    }
}

int gru_check(const char *fn, int T, int B, int H) {
    PSVAE_REQUIRE(T >= 0 && B >= 0 && H > 0, PSVAE_EINVAL, "%s: bad sizes T=%d B=%d H=%d", fn, T, B, H);
    PSVAE_REQUIRE(H % 4 == 0, PSVAE_EINVAL, "%s: H=%d must be a multiple of 4", fn, H);
    PSVAE_REQUIRE(H <= PSVAE_GRU_SEQ_MAX_H, PSVAE_ETOOLARGE, "%s: H=%d exceeds %d (use the per-step kernels)", fn, H,
                  PSVAE_GRU_SEQ_MAX_H);
    return PSVAE_OK;
}

}  // namespace

extern "C" int psvae_gru_seq_fwd(int T, int B, int H, const float *gi, const float *h0, const float *w_hh,
                                 const float *b_hh, float *hall, float *gates, float *ghn, void *stream) {
    const char *fn = "psvae_gru_seq_fwd";
    int rc = gru_check(fn, T, B, H);
    if (rc) return rc;
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(h0 && w_hh && hall && (T == 0 || gi), PSVAE_EINVAL, "%s: null pointer", fn);
    PSVAE_REQUIRE(psvae_aligned16(w_hh), PSVAE_EINVAL, "%s: w_hh must be 16-byte aligned", fn);
    GruFwdArgs a{T, B, H, gi, h0, w_hh, b_hh, hall, gates, ghn};
    const int KP = (H + 15) & ~15, nblk = KP / 16, jb = psvae_cdiv(nblk, 4);
    const size_t lds = 2 * (size_t)KP * GRU_LD * sizeof(float);
    const dim3 grid(psvae_cdiv(B, GRU_ROWS)), blk(GRU_THREADS);
    hipStream_t s = (hipStream_t)stream;
    switch (jb) {
        case 1: hipLaunchKernelGGL(gru_seq_fwd_kernel<1>, grid, blk, lds, s, a); break;
        case 2: hipLaunchKernelGGL(gru_seq_fwd_kernel<2>, grid, blk, lds, s, a); break;
        case 3: hipLaunchKernelGGL(gru_seq_fwd_kernel<3>, grid, blk, lds, s, a); break;
        default: hipLaunchKernelGGL(gru_seq_fwd_kernel<4>, grid, blk, lds, s, a); break;
    }
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_gru_seq_bwd(int T, int B, int H, const float *gates, const float *ghn, const float *hall,
                                 const float *g_out, const float *w_hh_t, float *dgi, float *dgh, float *dh0,
                                 void *stream) {
    const char *fn = "psvae_gru_seq_bwd";
    int rc = gru_check(fn, T, B, H);
    if (rc) return rc;
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(hall && w_hh_t && dh0 && (T == 0 || (gates && ghn && g_out && dgi && dgh)), PSVAE_EINVAL,
                  "%s: null pointer", fn);
    PSVAE_REQUIRE(psvae_aligned16(w_hh_t), PSVAE_EINVAL, "%s: w_hh_t must be 16-byte aligned", fn);
    GruBwdArgs a{T, B, H, gates, ghn, hall, g_out, w_hh_t, dgi, dgh, dh0};
    const int nblk = (H + 15) / 16, jb = psvae_cdiv(nblk, 4);
    const int K3P = (3 * H + 15) & ~15;
    const size_t lds = 2 * (size_t)K3P * GRU_LD * sizeof(float);
    PSVAE_REQUIRE(lds <= 160 * 1024, PSVAE_ETOOLARGE, "%s: LDS need %zu", fn, lds);
    const dim3 grid(psvae_cdiv(B, GRU_ROWS)), blk(GRU_THREADS);
    hipStream_t s = (hipStream_t)stream;
#define PSVAE_GRU_BWD(JB)                                                                                        \
    do {                                                                                                         \
        if (lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)gru_seq_bwd_kernel<JB>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                 \
        hipLaunchKernelGGL(gru_seq_bwd_kernel<JB>, grid, blk, lds, s, a);                                        \
    } while (0)
    switch (jb) {
        case 1: PSVAE_GRU_BWD(1); break;
        case 2: PSVAE_GRU_BWD(2); break;
        case 3: PSVAE_GRU_BWD(3); break;
        default: PSVAE_GRU_BWD(4); break;
    }
#undef PSVAE_GRU_BWD
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}
