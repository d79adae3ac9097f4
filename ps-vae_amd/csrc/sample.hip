// Generation-side helpers.
//
// (1) One step of the piece sampling loop, after the vocabulary projection: mask, softmax, draw, state
// update - one wave per molecule, one launch per step in place of ~18 element-wise launches.
// (2) The candidate atom pairs of the generated molecules (further down).
//
// Reference: src/modules/vae_piece_decoder.py:139-158 (inference): logits[pad] = -inf, logits[end] = -inf
// on the first step, probs = softmax(logits / temperature), next = torch.multinomial(probs, 1); the
// molecule stops after the end token, when its atom count exceeds max_atom_num or at max_pos pieces.
// The draw is the inverse-CDF form of a multinomial draw from a supplied uniform number:
//     next = min(#{i : cdf[i] <= u * cdf[last]}, last index of positive probability)
// with cdf the running sum of probs in index order (chunks of 64 scanned inside the wave, the carry
// added in order).  Which generator produced u cannot be pinned against torch.multinomial; given u,
// the pick differs from another float32 running sum of the same probs only where u * total falls
// within rounding of a cdf entry.
#include "common.h"

namespace {

__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v = fmaxf(v, __shfl_xor(v, d));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) v += __shfl_xor(v, d);
    return v;
}
__device__ __forceinline__ float wave_scan(float v, int lane) {   // inclusive, in lane order
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const float up = __shfl_up(v, d);
        if (lane >= d) v += up;
    }
    return v;
}

constexpr int WAVES = 4;

__global__ __launch_bounds__(64 * WAVES) void piece_draw_kernel(int B, int V, const float *__restrict__ logits, long long ldl,
                                                                float temperature, const float *__restrict__ u,
                                                                long long ldu, int pad, int end, int first_step,
                                                                const int32_t *__restrict__ piece_atoms, int max_atom_num,
                                                                int max_pos, uint8_t *__restrict__ active,
                                                                int32_t *__restrict__ atoms, int32_t *__restrict__ length,
                                                                int64_t *__restrict__ out, long long ldo,
                                                                int64_t *__restrict__ cur) {
    const int lane = threadIdx.x & 63;
    const int b = blockIdx.x * WAVES + (threadIdx.x >> 6);
    if (b >= B) return;
    const float *row = logits + (long long)b * ldl;
    const float NEG = -INFINITY;
    auto value = [&](int i) {
        if (i >= V || i == pad || (first_step && i == end)) return NEG;
        return row[i] / temperature;
    };
    float m = NEG;
    for (int i = lane; i < V; i += 64) m = fmaxf(m, value(i));
    m = wave_max(m);
    float s = 0.f;
    for (int i = lane; i < V; i += 64) {
        const float x = value(i);
        s += x == NEG ? 0.f : expf(x - m);
    }
    s = wave_sum(s);
    int pick = pad;
    if (active[b]) {
        if (!(s > 0.f) || m == NEG) {
            pick = end;                                   // no class of positive probability: stop the molecule
        } else {
            const float inv = 1.f / s;
            // pass 1: the total of the running sum and the last index of positive probability
            float carry = 0.f;
            int last = 0;
            for (int base = 0; base < V; base += 64) {
                const float x = value(base + lane);
                const float p = x == NEG ? 0.f : expf(x - m) * inv;
                const float c = wave_scan(p, lane) + carry;
                carry = __shfl(c, 63);
                const unsigned long long pos = __ballot(p > 0.f);
                if (pos) last = base + 63 - __clzll(pos);
            }
            const float thr = u[(long long)b * ldu] * carry;
            // pass 2: how many entries of the running sum do not exceed the threshold
            int count = 0;
            carry = 0.f;
            for (int base = 0; base < V; base += 64) {
                const float x = value(base + lane);
                const float p = x == NEG ? 0.f : expf(x - m) * inv;
                const float c = wave_scan(p, lane) + carry;
                carry = __shfl(c, 63);
                count += __popcll(__ballot(base + lane < V && c <= thr));
            }
            pick = count < last ? count : last;
        }
    }
    if (lane == 0) {
        const int was = active[b];
        out[(long long)b * ldo] = pick;
        int len = length[b] + was, at = atoms[b];
        if (was) at += piece_atoms[pick];
        const int stop = pick == end || at > max_atom_num || len == max_pos;
        const int now = was && !stop;
        length[b] = len, atoms[b] = at, active[b] = (uint8_t)now;
        cur[b] = now ? pick : end;
    }
}

// ------------------------------------------------------------------------------------------------
// Candidate atom pairs of generated molecules (src/modules/vae_piece_decoder.py:213-222): per molecule the
// pairs (m, k), m < k, of atoms in different fragments, in row-major order.  One workgroup per molecule;
// the fragment ids of its atoms sit in LDS; thread m owns row m.  Two passes: count per molecule, then
// (after a scan over the molecules on the caller's side) fill at ptr[b] + offset of row m.
constexpr int ENUM_THREADS = 256;

__device__ __forceinline__ int block_exclusive_scan(int v, int *scratch /* [ENUM_THREADS / 64 + 1] */, int *total) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int inc = v;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const int up = __shfl_up(inc, d);
        if (lane >= d) inc += up;
    }
    if (lane == 63) scratch[wave] = inc;
    __syncthreads();
    int before = 0, all = 0;
    for (int w = 0; w < ENUM_THREADS / 64; ++w) {
        if (w < wave) before += scratch[w];
        all += scratch[w];
    }
    __syncthreads();
    *total = all;
    return before + inc - v;
}

template <bool FILL>
__global__ __launch_bounds__(ENUM_THREADS) void pair_enumerate_kernel(int B, int max_atoms, const int32_t *__restrict__ node_off,
                                                                      const int32_t *__restrict__ frag,
                                                                      int32_t *__restrict__ counts,
                                                                      const int32_t *__restrict__ ptr, int32_t *__restrict__ src,
                                                                      int32_t *__restrict__ dst, int32_t *__restrict__ graph,
                                                                      int32_t *__restrict__ m_out, int32_t *__restrict__ k_out) {
    extern __shared__ int32_t s_frag[];          // [n]
    __shared__ int scratch[ENUM_THREADS / 64 + 1];
    const int b = blockIdx.x;
    const int base = node_off[b];
    const int n = min(node_off[b + 1] - base, max_atoms);   // (the LDS stage holds max_atoms ids: a larger molecule is cut there)
    for (int i = threadIdx.x; i < n; i += ENUM_THREADS) s_frag[i] = frag[base + i];
    __syncthreads();
    int running = 0;                              // candidates of the rows before this round of ENUM_THREADS rows
    for (int m0 = 0; m0 < n; m0 += ENUM_THREADS) {
        const int m = m0 + (int)threadIdx.x;
        int c = 0;
        if (m < n) {
            const int f = s_frag[m];
            for (int k = m + 1; k < n; ++k) c += s_frag[k] != f;
        }
        int round_total;
        const int off = block_exclusive_scan(c, scratch, &round_total);
        if (FILL && m < n && c > 0) {
            long long o = (long long)ptr[b] + running + off;
            const int f = s_frag[m];
            for (int k = m + 1; k < n; ++k) {
                if (s_frag[k] == f) continue;
                src[o] = base + m, dst[o] = base + k, graph[o] = b, m_out[o] = m, k_out[o] = k;
                ++o;
            }
        }
        running += round_total;
    }
    if (!FILL && threadIdx.x == 0) counts[b] = running;
}

}  // namespace

extern "C" int psvae_pair_enumerate_count(int B, int max_atoms, const int32_t *node_off, const int32_t *frag, int32_t *counts,
                                          void *stream) {
    const char *fn = "psvae_pair_enumerate_count";
    PSVAE_REQUIRE(B >= 0 && max_atoms >= 0, PSVAE_EINVAL, "%s: bad sizes", fn);
    PSVAE_REQUIRE((size_t)max_atoms * sizeof(int32_t) <= 60 * 1024, PSVAE_ETOOLARGE, "%s: a molecule of %d atoms exceeds the LDS stage", fn,
                  max_atoms);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(node_off && counts && (frag || max_atoms == 0), PSVAE_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(pair_enumerate_kernel<false>, dim3(B), dim3(ENUM_THREADS), (size_t)(max_atoms > 0 ? max_atoms : 1) * sizeof(int32_t),
                       (hipStream_t)stream, B, max_atoms, node_off, frag, counts, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_pair_enumerate_fill(int B, int max_atoms, const int32_t *node_off, const int32_t *frag, const int32_t *ptr,
                                         int32_t *src, int32_t *dst, int32_t *graph, int32_t *m, int32_t *k, void *stream) {
    const char *fn = "psvae_pair_enumerate_fill";
    PSVAE_REQUIRE(B >= 0 && max_atoms >= 0, PSVAE_EINVAL, "%s: bad sizes", fn);
    PSVAE_REQUIRE((size_t)max_atoms * sizeof(int32_t) <= 60 * 1024, PSVAE_ETOOLARGE, "%s: a molecule of %d atoms exceeds the LDS stage", fn,
                  max_atoms);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(node_off && ptr && (frag || max_atoms == 0), PSVAE_EINVAL, "%s: null pointer", fn);
    PSVAE_REQUIRE(src && dst && graph && m && k, PSVAE_EINVAL, "%s: null output", fn);
    hipLaunchKernelGGL(pair_enumerate_kernel<true>, dim3(B), dim3(ENUM_THREADS), (size_t)(max_atoms > 0 ? max_atoms : 1) * sizeof(int32_t),
                       (hipStream_t)stream, B, max_atoms, node_off, frag, nullptr, ptr, src, dst, graph, m, k);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}

extern "C" int psvae_piece_draw(int B, int V, const float *logits, int64_t ld_logits, float temperature, const float *uniforms,
                                int64_t ld_uniforms, int pad_idx, int end_idx, int first_step, const int32_t *piece_atoms,
                                int max_atom_num, int max_pos, uint8_t *active, int32_t *atoms, int32_t *length,
                                int64_t *out, int64_t ld_out, int64_t *cur, void *stream) {
    const char *fn = "psvae_piece_draw";
    PSVAE_REQUIRE(B >= 0 && V > 0 && ld_logits >= V, PSVAE_EINVAL, "%s: bad sizes B=%d V=%d ld=%lld", fn, B, V, (long long)ld_logits);
    PSVAE_REQUIRE(temperature > 0.f, PSVAE_EINVAL, "%s: temperature must be positive", fn);
    PSVAE_REQUIRE(pad_idx >= 0 && pad_idx < V && end_idx >= 0 && end_idx < V, PSVAE_EINVAL, "%s: pad / end index outside the vocabulary", fn);
    if (B == 0) return PSVAE_OK;
    PSVAE_REQUIRE(logits && uniforms && piece_atoms && active && atoms && length && out && cur, PSVAE_EINVAL, "%s: null pointer", fn);
    hipLaunchKernelGGL(piece_draw_kernel, dim3(psvae_cdiv(B, WAVES)), dim3(64 * WAVES), 0, (hipStream_t)stream, B, V, logits,
                       (long long)ld_logits, temperature, uniforms, (long long)ld_uniforms, pad_idx, end_idx, first_step,
                       piece_atoms, max_atom_num, max_pos, active, atoms, length, out, (long long)ld_out, cur);
    PSVAE_CHECK_LAUNCH(fn);
    return PSVAE_OK;
}
