// Cross-entropy of the piece language model over its vocabulary with an ignored label
// (nn.CrossEntropyLoss(ignore_index=pad), src/modules/vae_piece_decoder.py:53,113-114):
//     loss = mean over rows r with gold[r] != ignore of ( logsumexp(x[r]) - x[r][gold[r]] ).
// One pass over the logits: a wave takes a row (lane l holds columns l, l + 64, ...), reduces max
// and sum of exponentials across its lanes and writes the row of softmax(x) - onehot(gold) (zeros
// for an ignored row) - the gradient up to the factor g_loss / count.  Loss, token count and the
// count of rows whose first maximum is the gold label are summed per workgroup and then by one
// workgroup in a fixed order: the result does not depend on timing.
#include "common.h"

namespace {

constexpr int CE_THREADS = 256, CE_WAVES = CE_THREADS / 64, CE_MAX_BLOCKS = 2048, CE_MAX_PER_LANE = 16;

struct CePartial {
    float loss;
    int count, correct;
};

template <int PER_LANE>
__global__ __launch_bounds__(CE_THREADS) void token_ce_kernel(int R, int V, const float *__restrict__ x, long long ld,
                                                              const int *__restrict__ gold, int ignore,
                                                              float *__restrict__ dx, long long ldd,
                                                              CePartial *__restrict__ part) {
    __shared__ CePartial sh[CE_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f;
    int count = 0, correct = 0;
    for (long long r = (long long)blockIdx.x * CE_WAVES + wave; r < R; r += (long long)gridDim.x * CE_WAVES) {
        const float *row = x + r * ld;
        float v[PER_LANE];
        float mx = -__builtin_inff();
        int arg = V;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const int c = lane + 64 * i;
            v[i] = c < V ? row[c] : -__builtin_inff();
            if (v[i] > mx) mx = v[i], arg = c;   // columns ascend with i: the first maximum of this lane
        }
        // wave maximum and the smallest column that attains it
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off);
            const int oa = __shfl_xor(arg, off);
            if (om > mx || (om == mx && oa < arg)) mx = om, arg = oa;
        }
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            v[i] = (lane + 64 * i < V) ? __expf(v[i] - mx) : 0.f;
            sum += v[i];
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const int g = gold[r];
        const bool live = g != ignore;
        const float inv = 1.f / sum;
        float *drow = dx + r * ldd;
#pragma unroll
        for (int i = 0; i < PER_LANE; ++i) {
            const int c = lane + 64 * i;
            if (c < V)
                drow[c] = live ? v[i] * inv - (c == g ? 1.f : 0.f) : 0.f;
            else if (c < ldd)
                drow[c] = 0.f;   // padding columns of a vocabulary rounded up for the products that follow
        }
        if (live) {
            const float xg = (g >= 0 && g < V) ? row[g] : 0.f;   // (a label outside the vocabulary is the caller's error)
            loss += __logf(sum) + mx - xg;
            count += 1;
            correct += arg == g ? 1 : 0;
        }
    }
    if (lane == 0) sh[wave] = CePartial{loss, count, correct};
    __syncthreads();
    if (threadIdx.x == 0) {
        CePartial p = sh[0];
        for (int w = 1; w < CE_WAVES; ++w) p.loss += sh[w].loss, p.count += sh[w].count, p.correct += sh[w].correct;
        part[blockIdx.x] = p;
    }
}

// The same for vocabularies beyond 64 * CE_MAX_PER_LANE classes: the row does not fit the lanes'
// registers, so it is read three times (maximum, sum of exponentials, gradient row; the second and
// third read hit the cache).  Same reduction order across lanes, same partials.
__global__ __launch_bounds__(CE_THREADS) void token_ce_wide_kernel(int R, int V, const float *__restrict__ x, long long ld,
                                                                   const int *__restrict__ gold, int ignore,
                                                                   float *__restrict__ dx, long long ldd,
                                                                   CePartial *__restrict__ part) {
    __shared__ CePartial sh[CE_WAVES];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f;
    int count = 0, correct = 0;
    for (long long r = (long long)blockIdx.x * CE_WAVES + wave; r < R; r += (long long)gridDim.x * CE_WAVES) {
        const float *row = x + r * ld;
        float mx = -__builtin_inff();
        int arg = V;
        for (int c = lane; c < V; c += 64) {
            const float v = row[c];
            if (v > mx) mx = v, arg = c;
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const float om = __shfl_xor(mx, off);
            const int oa = __shfl_xor(arg, off);
            if (om > mx || (om == mx && oa < arg)) mx = om, arg = oa;
        }
        float sum = 0.f;
        for (int c = lane; c < V; c += 64) sum += __expf(row[c] - mx);
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
        const int g = gold[r];
        const bool live = g != ignore;
        const float inv = 1.f / sum;
        float *drow = dx + r * ldd;
        for (int c = lane; c < ldd; c += 64)
            drow[c] = (c < V && live) ? __expf(row[c] - mx) * inv - (c == g ? 1.f : 0.f) : 0.f;
        if (live) {
            const float xg = (g >= 0 && g < V) ? row[g] : 0.f;
            loss += __logf(sum) + mx - xg;
            count += 1;
            correct += arg == g ? 1 : 0;
        }
    }
    if (lane == 0) sh[wave] = CePartial{loss, count, correct};
    __syncthreads();
    if (threadIdx.x == 0) {
        CePartial p = sh[0];
        for (int w = 1; w < CE_WAVES; ++w) p.loss += sh[w].loss, p.count += sh[w].count, p.correct += sh[w].correct;
        part[blockIdx.x] = p;
    }
}

// One workgroup: thread t sums partials t, t + 256, ... in that order, then the 256 sums are added
// pairwise through LDS in a fixed tree: the result depends on the number of partials only.
__global__ __launch_bounds__(CE_THREADS) void token_ce_finish_kernel(int blocks, const CePartial *__restrict__ part,
                                                                     float *__restrict__ loss, int *__restrict__ counts) {
    __shared__ float sh_loss[CE_THREADS];
    __shared__ int sh_n[CE_THREADS], sh_ok[CE_THREADS];
    const int t = threadIdx.x;
    float s = 0.f;
    int n = 0, ok = 0;
#pragma unroll 8
    for (int b = t; b < blocks; b += CE_THREADS) {
        const CePartial p = part[b];
        s += p.loss, n += p.count, ok += p.correct;
    }
    sh_loss[t] = s, sh_n[t] = n, sh_ok[t] = ok;
    __syncthreads();
    for (int half = CE_THREADS / 2; half > 0; half >>= 1) {
        if (t < half) sh_loss[t] += sh_loss[t + half], sh_n[t] += sh_n[t + half], sh_ok[t] += sh_ok[t + half];
        __syncthreads();
    }
    if (t != 0) return;
    s = sh_loss[0], n = sh_n[0], ok = sh_ok[0];
    loss[0] = s / (float)n;          // no live row: 0 / 0 = NaN, as torch's mean over nothing
    loss[1] = 1.f / (float)n;
    counts[0] = n, counts[1] = ok;
}

int ce_blocks(int R) {
    const int b = psvae_cdiv(R, CE_WAVES);
    return b < 1 ? 1 : (b > CE_MAX_BLOCKS ? CE_MAX_BLOCKS : b);
}

}  // namespace

extern "C" size_t psvae_token_ce_workspace_bytes(int R) { return (size_t)ce_blocks(R > 0 ? R : 0) * sizeof(CePartial); }

extern "C" int psvae_token_ce_fwd(int R, int V, const float *logits, int64_t ld, const int32_t *gold, int ignore_index,
                                  float *dlogits, int64_t ldd, float *loss, int32_t *counts, void *workspace,
                                  size_t workspace_bytes, void *stream) {
    PSVAE_REQUIRE(R >= 0 && V >= 1, PSVAE_EINVAL, "psvae_token_ce_fwd: bad shape %d x %d", R, V);
    PSVAE_REQUIRE(loss != nullptr && counts != nullptr, PSVAE_EINVAL, "psvae_token_ce_fwd: null output");
    PSVAE_REQUIRE(R == 0 || (logits != nullptr && gold != nullptr && dlogits != nullptr && ld >= V && ldd >= V), PSVAE_EINVAL,
                  "psvae_token_ce_fwd: null operand or a row stride below V");
    PSVAE_REQUIRE(ldd < (1ll << 31), PSVAE_ETOOLARGE, "psvae_token_ce_fwd: dlogits row stride %lld", (long long)ldd);
    const int blocks = ce_blocks(R);
    PSVAE_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)blocks * sizeof(CePartial), PSVAE_EWORKSPACE,
                  "psvae_token_ce_fwd: workspace too small");
    hipStream_t s = (hipStream_t)stream;
    CePartial *part = (CePartial *)workspace;
    const int per_lane = psvae_cdiv(R > 0 && ldd > V ? (int)ldd : V, 64);
#define PSVAE_CE(N) \
    hipLaunchKernelGGL((token_ce_kernel<N>), dim3(blocks), dim3(CE_THREADS), 0, s, R, V, logits, (long long)ld, gold, \
                       ignore_index, dlogits, (long long)ldd, part)
    if (per_lane <= 1) PSVAE_CE(1);
    else if (per_lane <= 2) PSVAE_CE(2);
    else if (per_lane <= 5) PSVAE_CE(5);
    else if (per_lane <= 8) PSVAE_CE(8);
    else if (per_lane <= CE_MAX_PER_LANE) PSVAE_CE(16);
    else
        hipLaunchKernelGGL(token_ce_wide_kernel, dim3(blocks), dim3(CE_THREADS), 0, s, R, V, logits, (long long)ld, gold,
                           ignore_index, dlogits, (long long)ldd, part);
#undef PSVAE_CE
    PSVAE_CHECK_LAUNCH("psvae_token_ce_fwd");
    hipLaunchKernelGGL(token_ce_finish_kernel, dim3(1), dim3(CE_THREADS), 0, s, blocks, part, loss, counts);
    PSVAE_CHECK_LAUNCH("psvae_token_ce_fwd (finish)");
    return PSVAE_OK;
}
