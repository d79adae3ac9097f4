// The scalar tail of the training step and the block table of the atom features, each as ONE launch.
// At batch 32 a step is a chain of ~200 kernels of a few microseconds; the torch expressions these replace were
// five to twelve launches each (slice copies, scalar multiplies and adds, their autograd mirrors).
#include "common.h"

namespace {

// Z [na + np + nq, D + 4]: table k in its own column block, zeros elsewhere, a one in column D of the atom rows.
__global__ __launch_bounds__(256) void block_table_kernel(int na, int da, int np, int dp, int nq, int dq,
                                                          const float *__restrict__ ea, const float *__restrict__ ep,
                                                          const float *__restrict__ eq, float *__restrict__ Z) {
    const int D = da + dp + dq, W = D + 4, rows = na + np + nq;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < (long long)rows * W;
         i += (long long)gridDim.x * blockDim.x) {
        const int r = (int)(i / W), c = (int)(i % W);
        float v = 0.f;
        if (r < na) {
            if (c < da) v = ea[(long long)r * da + c];
            else if (c == D) v = 1.f;
        } else if (r < na + np) {
            if (c >= da && c < da + dp) v = ep[(long long)(r - na) * dp + (c - da)];
        } else {
            if (c >= da + dp && c < D) v = eq[(long long)(r - na - np) * dq + (c - da - dp)];
        }
        Z[i] = v;
    }
}

// mean over B x k of (pred[b][j] - props[b][sel[j]])^2; the differences are kept for the backward.
// One workgroup, partial sums in a fixed order.
__global__ __launch_bounds__(1024) void mse_selected_fwd_kernel(int B, int k, const float *__restrict__ pred, int ldp,
                                                               const float *__restrict__ props, int ldr,
                                                               const long long *__restrict__ sel, float *__restrict__ diff,
                                                               float *__restrict__ loss) {
    __shared__ float part[1024];
    const int n = B * k;
    float s = 0.f;
    for (int i0 = threadIdx.x; i0 < n; i0 += 1024 * 8) {   // eight entries requested before the first is used
        float a[8], r[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = min(i0 + 1024 * u, n - 1), b = i / k, j = i % k;
            a[u] = pred[(long long)b * ldp + j], r[u] = props[(long long)b * ldr + (int)sel[j]];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (i0 + 1024 * u < n) {
                const float d = a[u] - r[u];
                diff[i0 + 1024 * u] = d;
                s += d * d;
            }
    }
    part[threadIdx.x] = s;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if ((int)threadIdx.x < w) part[threadIdx.x] += part[threadIdx.x + w];
        __syncthreads();
    }
    if (threadIdx.x == 0) *loss = part[0] / (float)n;
}

__global__ __launch_bounds__(256) void mse_selected_bwd_kernel(int n, const float *__restrict__ diff, const float *__restrict__ g,
                                                               float *__restrict__ g_pred) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) g_pred[i] = diff[i] * (2.f / (float)n) * *g;
}

// total = alpha rec + (1 - alpha) pred + beta kl, each product and sum rounded on its own (as three multiplies
// and two adds in f32 are); backward: the three gradients g alpha, g (1 - alpha), g beta.
__global__ void loss_mix_fwd_kernel(const float *rec, const float *pred, const float *kl, float alpha, float one_minus_alpha,
                                    float beta, const float *beta_dev, float *out) {
#pragma clang fp contract(off)
    const float b = beta_dev ? *beta_dev : beta;
    const float t0 = alpha * *rec, t1 = one_minus_alpha * *pred, t2 = b * *kl;
    const float s01 = t0 + t1;
    *out = s01 + t2;
}

__global__ void loss_mix_bwd_kernel(const float *g, float alpha, float one_minus_alpha, float beta, const float *beta_dev,
                                    float *g3) {
    const float b = beta_dev ? *beta_dev : beta;
    g3[0] = *g * alpha, g3[1] = *g * one_minus_alpha, g3[2] = *g * b;
}

}  // namespace

extern "C" int psvae_block_table(int na, int da, int np, int dp, int nq, int dq, const float *ea, const float *ep,
                                 const float *eq, float *Z, void *stream) {
    PSVAE_REQUIRE(na >= 0 && np >= 0 && nq >= 0 && da >= 0 && dp >= 0 && dq >= 0, PSVAE_EINVAL, "psvae_block_table: negative size");
    const long long cells = (long long)(na + np + nq) * (da + dp + dq + 4);
    if (cells == 0) return PSVAE_OK;
    PSVAE_REQUIRE(Z && (ea || !na || !da) && (ep || !np || !dp) && (eq || !nq || !dq), PSVAE_EINVAL, "psvae_block_table: null pointer");
    const int blocks = (int)(cells + 255) / 256 < 1024 ? (int)((cells + 255) / 256) : 1024;
    hipLaunchKernelGGL(block_table_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, na, da, np, dp, nq, dq, ea, ep, eq, Z);
    PSVAE_CHECK_LAUNCH("psvae_block_table");
    return PSVAE_OK;
}

extern "C" int psvae_mse_selected_fwd(int B, int k, const float *pred, int ld_pred, const float *props, int ld_props,
                                      const int64_t *sel, float *diff, float *loss, void *stream) {
    PSVAE_REQUIRE(B > 0 && k > 0, PSVAE_EINVAL, "psvae_mse_selected_fwd: empty input (B=%d k=%d)", B, k);
    PSVAE_REQUIRE((long long)B * k <= (1 << 24), PSVAE_ETOOLARGE, "psvae_mse_selected_fwd: more than 2^24 entries");
    PSVAE_REQUIRE(pred && props && sel && diff && loss && ld_pred >= k, PSVAE_EINVAL, "psvae_mse_selected_fwd: null pointer or bad stride");
    hipLaunchKernelGGL(mse_selected_fwd_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, B, k, pred, ld_pred, props, ld_props,
                       (const long long *)sel, diff, loss);
    PSVAE_CHECK_LAUNCH("psvae_mse_selected_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_mse_selected_bwd(int n, const float *diff, const float *g, float *g_pred, void *stream) {
    PSVAE_REQUIRE(n > 0 && diff && g && g_pred, PSVAE_EINVAL, "psvae_mse_selected_bwd: empty input or null pointer");
    hipLaunchKernelGGL(mse_selected_bwd_kernel, dim3((n + 255) / 256), dim3(256), 0, (hipStream_t)stream, n, diff, g, g_pred);
    PSVAE_CHECK_LAUNCH("psvae_mse_selected_bwd");
    return PSVAE_OK;
}

extern "C" int psvae_loss_mix_fwd(const float *rec, const float *pred, const float *kl, float alpha, float one_minus_alpha,
                                  float beta, const float *beta_dev, float *out, void *stream) {
    PSVAE_REQUIRE(rec && pred && kl && out, PSVAE_EINVAL, "psvae_loss_mix_fwd: null pointer");
    hipLaunchKernelGGL(loss_mix_fwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, rec, pred, kl, alpha, one_minus_alpha, beta,
                       beta_dev, out);
    PSVAE_CHECK_LAUNCH("psvae_loss_mix_fwd");
    return PSVAE_OK;
}

extern "C" int psvae_loss_mix_bwd(const float *g, float alpha, float one_minus_alpha, float beta, const float *beta_dev, float *g3,
                                  void *stream) {
    PSVAE_REQUIRE(g && g3, PSVAE_EINVAL, "psvae_loss_mix_bwd: null pointer");
    hipLaunchKernelGGL(loss_mix_bwd_kernel, dim3(1), dim3(1), 0, (hipStream_t)stream, g, alpha, one_minus_alpha, beta, beta_dev, g3);
    PSVAE_CHECK_LAUNCH("psvae_loss_mix_bwd");
    return PSVAE_OK;
}
