// Optimiser tail on one flat parameter buffer: gradient scaling (1 / world size after the
// all-reduce), global-norm clipping and the Adam update in two launches.
//
// Replaces, for the whole model at once, what the reference's Lightning Trainer does per step:
// gradient_clip_val = 10.0 (src/train.py:47, torch.nn.utils.clip_grad_norm_ semantics:
// coef = min(1, max_norm / (norm + 1e-6))) followed by torch.optim.Adam
// (src/pl_models/ps_vae_model.py:123-127).  The arithmetic follows torch's single-tensor Adam:
//     m += (g - m) (1 - b1);  v = v b2 + (1 - b2) g g;
//     p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps)
// with the two bias corrections evaluated in double on the host.
//
// Kernel 1 writes one partial sum of squares per workgroup (fixed tree order); kernel 2 has every
// workgroup re-reduce those partials in the same fixed order (at most 1024 floats, L2-resident),
// so all of them agree on the norm bit for bit, then updates its slice.  No atomics anywhere: the
// step is run-to-run reproducible.
#include "common.h"

namespace {

constexpr int OPT_THREADS = 256;
constexpr int OPT_MAX_BLOCKS = 1024;
constexpr int OPT_VEC_PER_THREAD = 4;   // float4 per thread per trip

int opt_blocks(long long n) {
    const long long per_block = (long long)OPT_THREADS * 4 * OPT_VEC_PER_THREAD;
    long long b = (n + per_block - 1) / per_block;
    return (int)(b < 1 ? 1 : (b > OPT_MAX_BLOCKS ? OPT_MAX_BLOCKS : b));
}

__device__ __forceinline__ float block_sum(float s, float *sh) {
    // wave butterfly, then the waves' sums in wave order: the same order on every run
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) s += __shfl_xor(s, d);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();
    if (lane == 0) sh[wave] = s;
    __syncthreads();
    float t = sh[0];
#pragma unroll
    for (int w = 1; w < OPT_THREADS / 64; ++w) t += sh[w];
    return t;
}

// slice of workgroup b: elements [b * chunk, min(n, (b + 1) * chunk)), chunk a multiple of 4
__global__ __launch_bounds__(OPT_THREADS) void sqnorm_partial_kernel(long long n, long long chunk,
                                                                      const float *__restrict__ g, float scale,
                                                                      float *__restrict__ partial) {
    __shared__ float sh[OPT_THREADS / 64];
    const long long begin = (long long)blockIdx.x * chunk;
    const long long end = begin + chunk < n ? begin + chunk : n;
    float s = 0.f;
    const long long end4 = begin + ((end - begin) & ~3ll);
    for (long long i = begin + 4ll * threadIdx.x; i < end4; i += 4ll * OPT_THREADS) {
        const float4 v = *reinterpret_cast<const float4 *>(g + i);
        const float a = v.x * scale, b = v.y * scale, c = v.z * scale, d = v.w * scale;
        s += (a * a + b * b) + (c * c + d * d);
    }
    for (long long i = end4 + threadIdx.x; i < end; i += OPT_THREADS) {
        const float a = g[i] * scale;
        s += a * a;
    }
    const float t = block_sum(s, sh);
    if (threadIdx.x == 0) partial[blockIdx.x] = t;
}

struct AdamArgs {
    long long n, chunk;
    float *p;
    float *g;
    float *m;
    float *v;
    const float *partial;
    int n_partial;
    float grad_scale, max_norm;   // max_norm <= 0: no clipping
    float beta2, omb1, omb2, eps, step_size, bc2_sqrt, weight_decay;   // omb = 1 - beta, rounded once from double
    int write_grad;               // store the scaled + clipped gradient back (what clip_grad_norm_ leaves in .grad)
    float *norm_out;              // optional: the global norm before clipping
    const float *dev_scalars;     // optional, device [2]: step_size and bc2_sqrt read at run time instead of the two above
};

__device__ __forceinline__ void adam_one(const AdamArgs &a, float coef, float step_size, float bc2_sqrt, float &p, float &g,
                                         float &m, float &v) {
    g = g * coef;
    float gg = g;
    if (a.weight_decay != 0.f) gg += a.weight_decay * p;
    m = m + (gg - m) * a.omb1;
    v = v * a.beta2 + a.omb2 * gg * gg;
    const float denom = sqrtf(v) / bc2_sqrt + a.eps;
    p = p - step_size * (m / denom);
}

__global__ __launch_bounds__(OPT_THREADS) void clip_adam_kernel(AdamArgs a) {
    __shared__ float sh[OPT_THREADS / 64];
    float coef = a.grad_scale;
    if (a.max_norm > 0.f || a.norm_out != nullptr) {
        float s = 0.f;
        for (int i = threadIdx.x; i < a.n_partial; i += OPT_THREADS) s += a.partial[i];
        const float norm = sqrtf(block_sum(s, sh));
        if (a.norm_out != nullptr && blockIdx.x == 0 && threadIdx.x == 0) *a.norm_out = norm;
        if (a.max_norm > 0.f) {
            const float c = a.max_norm / (norm + 1e-6f);
            coef *= c < 1.f ? c : 1.f;
        }
    }
    const float ss = a.dev_scalars ? a.dev_scalars[0] : a.step_size, bq = a.dev_scalars ? a.dev_scalars[1] : a.bc2_sqrt;
    const long long begin = (long long)blockIdx.x * a.chunk;
    const long long end = begin + a.chunk < a.n ? begin + a.chunk : a.n;
    const long long end4 = begin + ((end - begin) & ~3ll);
    for (long long i = begin + 4ll * threadIdx.x; i < end4; i += 4ll * OPT_THREADS) {
        float4 p = *reinterpret_cast<float4 *>(a.p + i), g = *reinterpret_cast<const float4 *>(a.g + i);
        float4 m = *reinterpret_cast<float4 *>(a.m + i), v = *reinterpret_cast<float4 *>(a.v + i);
        adam_one(a, coef, ss, bq, p.x, g.x, m.x, v.x);
        adam_one(a, coef, ss, bq, p.y, g.y, m.y, v.y);
        adam_one(a, coef, ss, bq, p.z, g.z, m.z, v.z);
        adam_one(a, coef, ss, bq, p.w, g.w, m.w, v.w);
        *reinterpret_cast<float4 *>(a.p + i) = p;
        *reinterpret_cast<float4 *>(a.m + i) = m;
        *reinterpret_cast<float4 *>(a.v + i) = v;
        if (a.write_grad) *reinterpret_cast<float4 *>(a.g + i) = g;
    }
    for (long long i = end4 + threadIdx.x; i < end; i += OPT_THREADS) {
        float p = a.p[i], g = a.g[i], m = a.m[i], v = a.v[i];
        adam_one(a, coef, ss, bq, p, g, m, v);
        a.p[i] = p, a.m[i] = m, a.v[i] = v;
        if (a.write_grad) a.g[i] = g;
    }
}

long long opt_chunk(long long n, int blocks) {
    long long c = (n + blocks - 1) / blocks;
    return (c + 3) / 4 * 4;
}

}  // namespace

extern "C" size_t psvae_clip_adam_workspace_bytes(long long n) {
    if (n <= 0) return 0;
    return (size_t)opt_blocks(n) * sizeof(float);
}

static int clip_adam_impl(const char *fn, long long n, float *p, float *g, float *m, float *v, float grad_scale,
                          float max_norm, double lr, double beta1, double beta2, double eps, double weight_decay, int step,
                          const float *dev_scalars, int write_grad, float *norm_out, void *workspace,
                          size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(n >= 0, PSVAE_EINVAL, "%s: negative length", fn);
    if (n == 0) return PSVAE_OK;
    PSVAE_REQUIRE(p && g && m && v, PSVAE_EINVAL, "%s: null buffer", fn);
    PSVAE_REQUIRE(psvae_aligned16(p) && psvae_aligned16(g) && psvae_aligned16(m) && psvae_aligned16(v), PSVAE_EINVAL,
                  "%s: buffers must be 16-byte aligned", fn);
    PSVAE_REQUIRE(dev_scalars != nullptr || step >= 1, PSVAE_EINVAL, "%s: step counts from 1 (got %d)", fn, step);
    PSVAE_REQUIRE(beta1 >= 0. && beta1 < 1. && beta2 >= 0. && beta2 < 1. && eps >= 0., PSVAE_EINVAL,
                  "%s: bad hyper-parameters", fn);
    PSVAE_REQUIRE(workspace && workspace_bytes >= psvae_clip_adam_workspace_bytes(n), PSVAE_EWORKSPACE,
                  "%s: workspace too small", fn);
    const int blocks = opt_blocks(n);
    hipStream_t s = (hipStream_t)stream;
    AdamArgs a;
    a.n = n, a.chunk = opt_chunk(n, blocks);
    a.p = p, a.g = g, a.m = m, a.v = v, a.partial = (const float *)workspace, a.n_partial = blocks;
    a.grad_scale = grad_scale, a.max_norm = max_norm, a.beta2 = (float)beta2, a.eps = (float)eps;
    a.omb1 = (float)(1.0 - beta1), a.omb2 = (float)(1.0 - beta2);
    a.weight_decay = (float)weight_decay, a.write_grad = write_grad, a.norm_out = norm_out;
    a.dev_scalars = dev_scalars;
    a.step_size = 0.f, a.bc2_sqrt = 1.f;
    if (dev_scalars == nullptr) {
        const double bc1 = 1.0 - pow(beta1, (double)step), bc2 = 1.0 - pow(beta2, (double)step);
        a.step_size = (float)(lr / bc1);
        a.bc2_sqrt = (float)sqrt(bc2);
    }
    if (max_norm > 0.f || norm_out != nullptr) {
        hipLaunchKernelGGL(sqnorm_partial_kernel, dim3(blocks), dim3(OPT_THREADS), 0, s, n, a.chunk, (const float *)g,
                           grad_scale, (float *)workspace);
        PSVAE_CHECK_LAUNCH(fn);
    }
    hipLaunchKernelGGL(clip_adam_kernel, dim3(blocks), dim3(OPT_THREADS), 0, s, a);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_clip_adam_step(long long n, float *p, float *g, float *m, float *v, float grad_scale,
                                    float max_norm, double lr, double beta1, double beta2, double eps,
                                    double weight_decay, int step, int write_grad, float *norm_out, void *workspace,
                                    size_t workspace_bytes, void *stream) {
    return clip_adam_impl("psvae_clip_adam_step", n, p, g, m, v, grad_scale, max_norm, lr, beta1, beta2, eps, weight_decay, step,
                          nullptr, write_grad, norm_out, workspace, workspace_bytes, stream);
}

extern "C" int psvae_clip_adam_step_dev(long long n, float *p, float *g, float *m, float *v, float grad_scale,
                                        float max_norm, double beta1, double beta2, double eps, double weight_decay,
                                        const float *step_scalars, int write_grad, float *norm_out, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    const char *fn = "psvae_clip_adam_step_dev";
    PSVAE_REQUIRE(step_scalars != nullptr, PSVAE_EINVAL, "%s: step_scalars is null", fn);
    return clip_adam_impl(fn, n, p, g, m, v, grad_scale, max_norm, 0.0, beta1, beta2, eps, weight_decay, 0, step_scalars,
                          write_grad, norm_out, workspace, workspace_bytes, stream);
}
