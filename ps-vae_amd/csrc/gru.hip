// The piece language model's GRU recurrence (nn.GRU at src/modules/vae_piece_decoder.py:29,
// teacher-forced at :85-89, sampled at :139-158) as ONE launch per direction.
//
// The recurrence couples the hidden units of a molecule, not the molecules of a batch: a
// workgroup owns 16 batch rows and walks all T time steps for them, so nothing crosses
// workgroups and no grid-wide synchronisation exists.  Per step it computes
//     gh = h W_hh^T + b_hh           (16 x 3H, v_mfma_f32_16x16x4_f32: exact f32)
//     r = s(gi_r + gh_r), z = s(gi_z + gh_z), n = tanh(gi_n + r gh_n), h' = (1 - z) n + z h
// (PyTorch gate order r, z, n; gi = x W_ih^T + b_ih comes from one large product beforehand).
// Hidden units are cut into blocks of 16; wave w of the NW waves owns blocks w, w + NW, ... and, for
// each, the three gate tiles of those units (NW = 16 for the model's 200 units: one block per wave, a
// step's ~1900 MFMAs spread over 13 waves instead of 4 - the recurrence is a serial chain of steps, so
// the time of one step is what counts: 15 -> 5 us).  The product is taken transposed (gh^T = W_hh h^T: A = W_hh,
// B = h^T), so that in the accumulator layout a lane is a batch row and its four registers are
// four CONSECUTIVE hidden units: the gate arithmetic runs on the accumulators where they are and
// every global access of the step (gi, gates, hidden states) is 16 bytes per lane.  The new
// hidden state goes straight back to LDS as the next step's B operand.  h lives in LDS k-major
// ([unit][row], row stride 20 floats: the four k groups of a wave read disjoint banks); W_hh
// (3H x H, L2-resident) is read 16 bytes per lane along k, two k blocks ahead.  Within a 16-deep
// k block lane group q takes k = 16b + 4q + e in MFMA number e - for A and B alike - so the sum
// over k is complete whatever the order.
//
// Backward mirrors it: dgh (16 x 3H, k-major in LDS) times W_hh gives the gradient that flows to
// the previous step, the gate derivatives are evaluated in the same register layout, and the
// weight gradients are left to two large products over the stacked dgi / dgh the kernel writes.
#include "common.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));

namespace {

constexpr int GRU_ROWS = 16;        // batch rows per workgroup
constexpr int GRU_LD = 20;          // LDS row stride of the k-major operand images
constexpr int GRU_WAVES_WIDE = 16;  // waves per workgroup when the hidden width has more than four blocks of 16 units

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

template <int JB, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_seq_fwd_kernel(GruFwdArgs a) {
    constexpr int GRU_THREADS = 64 * NW;
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

    // A operand rows: gate c, unit 16 blk + jl -> row c*H + unit of W_hh (clamped for the padded
    // units, whose results are never stored); 16 bytes along k per lane
    const float *wrow[JB][3];
#pragma unroll
    for (int q = 0; q < JB; ++q) {
        const int u = min((wave + NW * q) * 16 + jl, H - 1);
#pragma unroll
        for (int c = 0; c < 3; ++c) wrow[q][c] = a.w_hh + (long long)(c * H + u) * H;
    }
    const int brow = row0 + jl;                 // the batch row of this lane in the accumulator layout
    const bool row_ok = brow < B;
    for (int t = 0; t < a.T; ++t) {
        const float *hc = hk[t & 1];
        float *hn = hk[(t + 1) & 1];
        // input projections of this step (units 16 blk + 4g .. +3 of row brow: 16 bytes per gate),
        // requested before the products they do not depend on
        float4 gi4[JB][3];
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int u0 = (wave + NW * q) * 16 + 4 * g;
            const bool ok = (wave + NW * q) < nblk && u0 < H && row_ok;
            const float *p = a.gi + (ok ? ((long long)t * B + brow) * H3 + u0 : 0);
#pragma unroll
            for (int c = 0; c < 3; ++c) gi4[q][c] = *reinterpret_cast<const float4 *>(p + (ok ? c * H : 0));
        }
        f32x4 acc[JB][3];
#pragma unroll
        for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int c = 0; c < 3; ++c) acc[q][c] = (f32x4){0.f, 0.f, 0.f, 0.f};
        auto load_w = [&](float4 (&w)[JB][3], int kb) {
            const int k = min(16 * kb + 4 * g, H - 4);   // the padded k tail multiplies zero rows of h
#pragma unroll
            for (int q = 0; q < JB; ++q)
                if (wave + NW * q < nblk) {
#pragma unroll
                    for (int c = 0; c < 3; ++c) w[q][c] = *reinterpret_cast<const float4 *>(wrow[q][c] + k);
                }
        };
        auto multiply = [&](const float4 (&w)[JB][3], int kb) {
            float hv[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) hv[e] = hc[(16 * kb + 4 * g + e) * GRU_LD + jl];
            // e outermost: consecutive MFMAs hit different accumulators (a dependent pair costs 40
            // cycles, an independent one 32)
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int q = 0; q < JB; ++q)
                    if (wave + NW * q < nblk) {
#pragma unroll
                        for (int c = 0; c < 3; ++c) {
                            const float wv = e == 0 ? w[q][c].x : (e == 1 ? w[q][c].y : (e == 2 ? w[q][c].z : w[q][c].w));
                            acc[q][c] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, hv[e], acc[q][c], 0, 0, 0);
                        }
                    }
        };
        // W_hh fragments travel two k blocks ahead of their use (one wave per SIMD: nothing else
        // hides the L2 latency)
        float4 w0[JB][3], w1[JB][3], w2[JB][3];
        load_w(w0, 0);
        if (nblk > 1) load_w(w1, 1);
        for (int kb = 0; kb < nblk; kb += 3) {
            if (kb + 2 < nblk) load_w(w2, kb + 2);
            multiply(w0, kb);
            if (kb + 1 < nblk) {
                if (kb + 3 < nblk) load_w(w0, kb + 3);
                multiply(w1, kb + 1);
            }
            if (kb + 2 < nblk) {
                if (kb + 4 < nblk) load_w(w1, kb + 4);
                multiply(w2, kb + 2);
            }
        }
        // gates on the accumulators: lane = batch row, registers = units u0 .. u0+3
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int blk = wave + NW * q;
            const int u0 = blk * 16 + 4 * g;
            if (blk >= nblk || u0 >= H) continue;
            float4 b3[3];
#pragma unroll
            for (int c = 0; c < 3; ++c)
                b3[c] = a.b_hh ? *reinterpret_cast<const float4 *>(a.b_hh + c * H + u0) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float bias[3][4] = {{b3[0].x, b3[0].y, b3[0].z, b3[0].w}, {b3[1].x, b3[1].y, b3[1].z, b3[1].w},
                                      {b3[2].x, b3[2].y, b3[2].z, b3[2].w}};
            const float gir[4] = {gi4[q][0].x, gi4[q][0].y, gi4[q][0].z, gi4[q][0].w};
            const float giz[4] = {gi4[q][1].x, gi4[q][1].y, gi4[q][1].z, gi4[q][1].w};
            const float gin[4] = {gi4[q][2].x, gi4[q][2].y, gi4[q][2].z, gi4[q][2].w};
            float rr[4], zz[4], nn[4], gn[4], hnew[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float ghr = acc[q][0][r] + bias[0][r], ghz = acc[q][1][r] + bias[1][r];
                gn[r] = acc[q][2][r] + bias[2][r];
                rr[r] = sigmoidf_(gir[r] + ghr);
                zz[r] = sigmoidf_(giz[r] + ghz);
                nn[r] = tanhf(gin[r] + rr[r] * gn[r]);
                const float hp = hc[(u0 + r) * GRU_LD + jl];
                hnew[r] = (1.0f - zz[r]) * nn[r] + zz[r] * hp;
                hn[(u0 + r) * GRU_LD + jl] = row_ok ? hnew[r] : 0.f;
            }
            if (row_ok) {
                const long long o = ((long long)t * B + brow) * H3 + u0;
                if (a.gates) {
                    *reinterpret_cast<float4 *>(a.gates + o) = make_float4(rr[0], rr[1], rr[2], rr[3]);
                    *reinterpret_cast<float4 *>(a.gates + o + H) = make_float4(zz[0], zz[1], zz[2], zz[3]);
                    *reinterpret_cast<float4 *>(a.gates + o + 2 * H) = make_float4(nn[0], nn[1], nn[2], nn[3]);
                }
                if (a.ghn)
                    *reinterpret_cast<float4 *>(a.ghn + ((long long)t * B + brow) * H + u0) =
                        make_float4(gn[0], gn[1], gn[2], gn[3]);
                *reinterpret_cast<float4 *>(a.hall + ((long long)(t + 1) * B + brow) * H + u0) =
                    make_float4(hnew[0], hnew[1], hnew[2], hnew[3]);
            }
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

template <int JB, int NW>
__global__ __launch_bounds__(64 * NW, 1) void gru_seq_bwd_kernel(GruBwdArgs a) {
    constexpr int GRU_THREADS = 64 * NW;
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
    const int brow = row0 + jl;                 // the batch row of this lane in the accumulator layout
    const bool row_ok = brow < B;
    const float *wrow[JB];                      // A operand rows: unit 16 blk + jl of W_hh^T (clamped)
#pragma unroll
    for (int q = 0; q < JB; ++q) wrow[q] = a.w_hh_t + (long long)min((wave + NW * q) * 16 + jl, H - 1) * H3;

    // saved values of one step in the register layout of the gate arithmetic: for row brow, units
    // u0 .. u0+3 of every block this wave owns (16 bytes per array)
    struct Saved {
        float4 rr[JB], zz[JB], nn[JB], gn[JB], hp[JB], go[JB];
    };
    auto fetch = [&](Saved &s, int t) {
#pragma unroll
        for (int q = 0; q < JB; ++q) {
            const int u0 = (wave + NW * q) * 16 + 4 * g;
            const bool ok = (wave + NW * q) < nblk && u0 < H && row_ok;
            const long long o3 = ok ? ((long long)t * B + brow) * H3 + u0 : 0;
            const long long o1 = ok ? ((long long)t * B + brow) * H + u0 : 0;
            s.rr[q] = *reinterpret_cast<const float4 *>(a.gates + o3);
            s.zz[q] = *reinterpret_cast<const float4 *>(a.gates + o3 + (ok ? H : 0));
            s.nn[q] = *reinterpret_cast<const float4 *>(a.gates + o3 + (ok ? 2 * H : 0));
            s.gn[q] = *reinterpret_cast<const float4 *>(a.ghn + o1);
            s.hp[q] = *reinterpret_cast<const float4 *>(a.hall + o1);
            s.go[q] = *reinterpret_cast<const float4 *>(a.g_out + o1);
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
            const int blk = wave + NW * q;
            const int u0 = blk * 16 + 4 * g;
            if (blk >= nblk || u0 >= H) continue;
            const float rr[4] = {cur.rr[q].x, cur.rr[q].y, cur.rr[q].z, cur.rr[q].w};
            const float zz[4] = {cur.zz[q].x, cur.zz[q].y, cur.zz[q].z, cur.zz[q].w};
            const float nn[4] = {cur.nn[q].x, cur.nn[q].y, cur.nn[q].z, cur.nn[q].w};
            const float gn[4] = {cur.gn[q].x, cur.gn[q].y, cur.gn[q].z, cur.gn[q].w};
            const float hp[4] = {cur.hp[q].x, cur.hp[q].y, cur.hp[q].z, cur.hp[q].w};
            const float go[4] = {cur.go[q].x, cur.go[q].y, cur.go[q].z, cur.go[q].w};
            float vr[4], vz[4], vn[4], vnr[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float d = go[r] + dh[q][r];
                const float dn_pre = d * (1.0f - zz[r]) * (1.0f - nn[r] * nn[r]);
                const float dz_pre = d * (hp[r] - nn[r]) * zz[r] * (1.0f - zz[r]);
                const float dr_pre = dn_pre * gn[r] * rr[r] * (1.0f - rr[r]);
                dzh[q][r] = d * zz[r];
                vr[r] = row_ok ? dr_pre : 0.f, vz[r] = row_ok ? dz_pre : 0.f;
                vn[r] = row_ok ? dn_pre : 0.f, vnr[r] = row_ok ? dn_pre * rr[r] : 0.f;
                dc[(u0 + r) * GRU_LD + jl] = vr[r];
                dc[(H + u0 + r) * GRU_LD + jl] = vz[r];
                dc[(2 * H + u0 + r) * GRU_LD + jl] = vnr[r];
            }
            if (row_ok) {
                const long long o = ((long long)t * B + brow) * H3 + u0;
                const float4 r4 = make_float4(vr[0], vr[1], vr[2], vr[3]), z4 = make_float4(vz[0], vz[1], vz[2], vz[3]);
                *reinterpret_cast<float4 *>(a.dgi + o) = r4;
                *reinterpret_cast<float4 *>(a.dgi + o + H) = z4;
                *reinterpret_cast<float4 *>(a.dgi + o + 2 * H) = make_float4(vn[0], vn[1], vn[2], vn[3]);
                *reinterpret_cast<float4 *>(a.dgh + o) = r4;
                *reinterpret_cast<float4 *>(a.dgh + o + H) = z4;
                *reinterpret_cast<float4 *>(a.dgh + o + 2 * H) = make_float4(vnr[0], vnr[1], vnr[2], vnr[3]);
            }
        }
        __syncthreads();
        if (t > 0) fetch(cur, t - 1);   // the next step's saved values travel while the product runs

        f32x4 acc[JB];
#pragma unroll
        for (int q = 0; q < JB; ++q) acc[q] = (f32x4){0.f, 0.f, 0.f, 0.f};
        constexpr int G = NW > 4 ? 3 : 4;   // k blocks per group; one group of W_hh^T fragments travels ahead (128 registers per lane at 16 waves)
        auto load_w = [&](float4 (&w)[G][JB], int kb0) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                const int k = min(16 * (kb0 + u) + 4 * g, H3 - 4);   // the padded k tail multiplies zero rows
#pragma unroll
                for (int q = 0; q < JB; ++q)
                    if (wave + NW * q < nblk) w[u][q] = *reinterpret_cast<const float4 *>(wrow[q] + k);
            }
        };
        auto multiply = [&](const float4 (&w)[G][JB], int kb0) {
#pragma unroll
            for (int u = 0; u < G; ++u) {
                if (kb0 + u >= nkb) break;
                float dv[4];
#pragma unroll
                for (int e = 0; e < 4; ++e) dv[e] = dc[(16 * (kb0 + u) + 4 * g + e) * GRU_LD + jl];
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int q = 0; q < JB; ++q)
                        if (wave + NW * q < nblk) {
                            const float wv = e == 0 ? w[u][q].x : (e == 1 ? w[u][q].y : (e == 2 ? w[u][q].z : w[u][q].w));
                            acc[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wv, dv[e], acc[q], 0, 0, 0);
                        }
            }
        };
        float4 wa[G][JB], wb[G][JB];
        load_w(wa, 0);
        for (int kb = 0; kb < nkb; kb += 2 * G) {
            if (kb + G < nkb) load_w(wb, kb + G);
            multiply(wa, kb);
            if (kb + G < nkb) {
                if (kb + 2 * G < nkb) load_w(wa, kb + 2 * G);
                multiply(wb, kb + G);
            }
        }
#pragma unroll
        for (int q = 0; q < JB; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) dh[q][r] = acc[q][r] + dzh[q][r];
    }
#pragma unroll
    for (int q = 0; q < JB; ++q) {
        const int u0 = (wave + NW * q) * 16 + 4 * g;
        if (wave + NW * q >= nblk || u0 >= H || !row_ok) continue;
        *reinterpret_cast<float4 *>(a.dh0 + (long long)brow * H + u0) = make_float4(dh[q][0], dh[q][1], dh[q][2], dh[q][3]);
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
    PSVAE_REQUIRE(psvae_aligned16(w_hh) && psvae_aligned16(gi) && psvae_aligned16(hall) && psvae_aligned16(gates) &&
                      psvae_aligned16(ghn) && psvae_aligned16(b_hh),
                  PSVAE_EINVAL, "%s: every array must be 16-byte aligned", fn);
    GruFwdArgs a{T, B, H, gi, h0, w_hh, b_hh, hall, gates, ghn};
    const int KP = (H + 15) & ~15, nblk = KP / 16;
    const int nw = nblk > 4 ? GRU_WAVES_WIDE : 4, jb = psvae_cdiv(nblk, nw);
    const size_t lds = 2 * (size_t)KP * GRU_LD * sizeof(float);
    const dim3 grid(psvae_cdiv(B, GRU_ROWS)), blk(64 * nw);
    hipStream_t s = (hipStream_t)stream;
    if (nw == 4) {
        hipLaunchKernelGGL((gru_seq_fwd_kernel<1, 4>), grid, blk, lds, s, a);
    } else {
        static_assert(PSVAE_GRU_SEQ_MAX_H <= 16 * GRU_WAVES_WIDE, "one block of 16 units per wave");
        (void)jb;
        hipLaunchKernelGGL((gru_seq_fwd_kernel<1, GRU_WAVES_WIDE>), grid, blk, lds, s, a);
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
    PSVAE_REQUIRE(psvae_aligned16(w_hh_t) && psvae_aligned16(gates) && psvae_aligned16(ghn) && psvae_aligned16(hall) &&
                      psvae_aligned16(g_out) && psvae_aligned16(dgi) && psvae_aligned16(dgh) && psvae_aligned16(dh0),
                  PSVAE_EINVAL, "%s: every array must be 16-byte aligned", fn);
    GruBwdArgs a{T, B, H, gates, ghn, hall, g_out, w_hh_t, dgi, dgh, dh0};
    const int nblk = (H + 15) / 16;
    const int nw = nblk > 4 ? GRU_WAVES_WIDE : 4, jb = psvae_cdiv(nblk, nw);
    const int K3P = (3 * H + 15) & ~15;
    const size_t lds = 2 * (size_t)K3P * GRU_LD * sizeof(float);
    PSVAE_REQUIRE(lds <= 160 * 1024, PSVAE_ETOOLARGE, "%s: LDS need %zu", fn, lds);
    const dim3 grid(psvae_cdiv(B, GRU_ROWS)), blk(64 * nw);
    hipStream_t s = (hipStream_t)stream;
#define PSVAE_GRU_BWD(JB, NW)                                                                                    \
    do {                                                                                                         \
        if (lds > 64 * 1024)                                                                                     \
            (void)hipFuncSetAttribute((const void *)gru_seq_bwd_kernel<JB, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                      (int)lds);                                                                 \
        hipLaunchKernelGGL((gru_seq_bwd_kernel<JB, NW>), grid, blk, lds, s, a);                                  \
    } while (0)
    if (nw == 4) {
        PSVAE_GRU_BWD(1, 4);
    } else {
        (void)jb;
        PSVAE_GRU_BWD(1, GRU_WAVES_WIDE);
    }
#undef PSVAE_GRU_BWD
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}
