// Stand-alone probe: which ingredient of a panel-GEMM k loop costs matrix-core time?  768 workgroups of 4 waves
// (3 per CU, as the panel kernel runs), each wave 5 accumulator tiles of v_mfma_f32_32x32x2_f32, 16 k-steps per
// k-tile.  MODE bits: 1 = operands read from LDS (6 ds_read_b32 per step, one step ahead), 2 = two barriers per
// k-tile, 4 = 36 ds_write_b32 per thread and k-tile (staging), 8 = 9 global 16-byte loads per thread and k-tile,
// 16 = the same 9 x 16 bytes per thread straight to LDS (global_load_lds_dwordx4, two LDS stages, no staging
// registers, each wave waits for its own loads before the barrier), 32 = staging with ds_write_b128,
// 64 = operands read by ds_read_b64 from row-major tiles of 34-word rows (a lane's two registers feed two
// consecutive MFMAs: lanes 0-31 hold k = 4j, 4j+1, lanes 32-63 k = 4j+2, 4j+3; 6 reads per 10 MFMAs),
// 128 = staging into those tiles with 18 ds_write_b64 per thread and k-tile, 256 (with 8) = the global loads one
// per k-step behind that step's MFMAs instead of nine in a row at the top of the k-tile.
// Build: hipcc --offload-arch=gfx950 -O3 tools/mfma_loop_probe.hip -o exp_libs/mfma_loop_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
constexpr int A_STRIDE = 129, B_STRIDE = 164, BK = 32, TN = 5;

template <int MODE>
__global__ __launch_bounds__(256, 3) void probe(float *out, const float4 *src, int tiles, long long src_quads) {
    __shared__ float As[BK * A_STRIDE];
    __shared__ float Bs[BK * B_STRIDE];
    __shared__ __align__(16) float Ds[(MODE & 16) ? 2 * 9 * 256 * 4 : 4];   // two stages of 36 KiB
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < BK * A_STRIDE; i += 256) As[i] = 0.001f * (i % 97);
    for (int i = tid; i < BK * B_STRIDE; i += 256) Bs[i] = 0.002f * (i % 89);
    __syncthreads();
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    const float *ar = As + lhi * A_STRIDE + wave * 32 + l31, *br = Bs + lhi * B_STRIDE + l31;
    float4 stage[9];
    for (int i = 0; i < 9; ++i) stage[i] = make_float4(tid, i, 1.f, 2.f);
    long long gq = ((long long)blockIdx.x * 256 + tid) % src_quads;
    constexpr int RS = BK + 2;                       // row-major tiles: 34-word rows, conflict-free b64 reads
    const float2 *ar2 = reinterpret_cast<const float2 *>(As + (wave * 32 + l31) * RS + 2 * lhi);
    const float2 *br2 = reinterpret_cast<const float2 *>(Bs + l31 * RS + 2 * lhi);
    float a_reg = 0.5f + lane, b_reg[TN];
    for (int j = 0; j < TN; ++j) b_reg[j] = 0.25f * (j + 1);
    for (int t = 0; t < tiles; ++t) {
        if (MODE & 4) {   // staging stores of the tile loaded during the previous one
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                float *d = As + (4 * ((tid + 256 * i) % 8)) * A_STRIDE + (tid + 256 * i) / 8;
                d[0] = stage[i].x, d[A_STRIDE] = stage[i].y, d[2 * A_STRIDE] = stage[i].z, d[3 * A_STRIDE] = stage[i].w;
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int item = tid + 256 * i;
                float *d = Bs + (4 * (item % 8)) * B_STRIDE + item / 8;
                d[0] = stage[4 + i].x, d[B_STRIDE] = stage[4 + i].y, d[2 * B_STRIDE] = stage[4 + i].z, d[3 * B_STRIDE] = stage[4 + i].w;
            }
        }
        if (MODE & 32) {
#pragma unroll
            for (int i = 0; i < 9; ++i) *reinterpret_cast<float4 *>(As + ((tid + 256 * i) * 4) % (BK * A_STRIDE - 4) / 4 * 4) = stage[i];
        }
        if (MODE & 128) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int item = tid + 256 * i;
                float2 *d = reinterpret_cast<float2 *>(As + (item / 8) * RS + 4 * (item % 8));
                d[0] = make_float2(stage[i].x, stage[i].y), d[1] = make_float2(stage[i].z, stage[i].w);
            }
#pragma unroll
            for (int i = 0; i < 5; ++i) {
                const int item = tid + 256 * i;
                float2 *d = reinterpret_cast<float2 *>(Bs + (item / 8) * RS + 4 * (item % 8));
                d[0] = make_float2(stage[4 + i].x, stage[4 + i].y), d[1] = make_float2(stage[4 + i].z, stage[4 + i].w);
            }
        }
        if (MODE & 16) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        if (MODE & 2) __syncthreads();
        if (MODE & 16) {
            float *dst = Ds + (t & 1) * 9 * 256 * 4 + wave * 9 * 64 * 4;
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + gq),
                                                 (__attribute__((address_space(3))) void *)(dst + i * 64 * 4), 16, 0, 0);
                gq += 256 * 768;
                if (gq >= src_quads) gq -= src_quads;
            }
        }
#define PROBE_LOAD(i)                                                                                  \
    do {                                                                                               \
        if ((i) < 4) {                                                                                 \
            stage[i] = src[gq];                                                                        \
            gq += 256 * 768;                                                                           \
            if (gq >= src_quads) gq -= src_quads;                                                      \
        } else {                                                                                       \
            stage[i] = src[((t % 10) * 5 + (i) - 4) * 256 + tid];                                      \
        }                                                                                              \
    } while (0)
        if ((MODE & 8) && !(MODE & 256)) {
#pragma unroll
            for (int i = 0; i < 9; ++i) {
                if (i < 4) {                      // the A panel streams from HBM: 16 KiB per workgroup and k-tile
                    stage[i] = src[gq];
                    gq += 256 * 768;
                    if (gq >= src_quads) gq -= src_quads;
                } else {                          // the B tile is the same 400 KB for every workgroup: L2 hits
                    stage[i] = src[((t % 10) * 5 + i - 4) * 256 + tid];
                }
            }
        }
        if (MODE & 64) {
            float2 a2 = ar2[0], b2[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b2[j] = br2[j * 32 * RS / 2];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 4) {
                float2 a2n = a2, b2n[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b2n[j] = b2[j];
                if (kk + 4 < BK) {
                    a2n = ar2[(kk + 4) / 2];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b2n[j] = br2[j * 32 * RS / 2 + (kk + 4) / 2];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.x, b2[j].x, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2.y, b2[j].y, acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a2 = a2n;
#pragma unroll
                for (int j = 0; j < TN; ++j) b2[j] = b2n[j];
            }
            if (MODE & 2) __syncthreads();
            continue;
        }
        float a_cur = (MODE & 1) ? ar[0] : a_reg, b_cur[TN];
#pragma unroll
        for (int j = 0; j < TN; ++j) b_cur[j] = (MODE & 1) ? br[j * 32] : b_reg[j];
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            float a_nxt = a_cur, b_nxt[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b_nxt[j] = b_cur[j];
            if ((MODE & 1) && kk + 2 < BK) {
                a_nxt = ar[(kk + 2) * A_STRIDE];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_nxt[j] = br[(kk + 2) * B_STRIDE + j * 32];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
            if ((MODE & 256) && kk / 2 < 9) {    // this k-tile's global loads spread over the k-steps, each behind five queued MFMAs
                PROBE_LOAD(kk / 2);
                __builtin_amdgcn_sched_barrier(0);
            }
            a_cur = a_nxt;
#pragma unroll
            for (int j = 0; j < TN; ++j) b_cur[j] = b_nxt[j];
        }
        if (MODE & 2) __syncthreads();
    }
    float s = 0.f;
    for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    for (int i = 0; i < 9; ++i) s += stage[i].x;
    out[blockIdx.x * 256 + tid] = s;
}


// Producer / consumer split: a fifth wave does all global loads and staging stores of the workgroup (36 loads and
// 144 ds_write_b32 per k-tile), the four others only read operands from LDS and multiply; two LDS stages, one
// barrier per k-tile.  WGS workgroups per CU are resident (register budget permitting).
template <int WGS>
__global__ __launch_bounds__(320, WGS) void probe_split(float *out, const float4 *src, int tiles, long long src_quads) {
    __shared__ float As[2][BK * A_STRIDE];
    __shared__ float Bs[2][BK * B_STRIDE];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, lhi = lane >> 5;
    for (int i = tid; i < 2 * BK * A_STRIDE; i += 320) (&As[0][0])[i] = 0.001f * (i % 97);
    for (int i = tid; i < 2 * BK * B_STRIDE; i += 320) (&Bs[0][0])[i] = 0.002f * (i % 89);
    __syncthreads();
    f32x16 acc[TN];
    for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    float s = 0.f;
    if (wave == 4) {
        long long gq = ((long long)blockIdx.x * 64 + lane) % src_quads;
        float4 stage[36];
        for (int t = 0; t < tiles; ++t) {
#pragma unroll
            for (int i = 0; i < 36; ++i) {
                if (i < 16) {
                    stage[i] = src[gq];
                    gq += 64 * 768;
                    if (gq >= src_quads) gq -= src_quads;
                } else {
                    stage[i] = src[((t % 10) * 20 + i - 16) * 64 + lane];
                }
            }
            float *a = As[(t + 1) & 1], *b = Bs[(t + 1) & 1];
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const int item = lane + 64 * i;
                float *d = a + (4 * (item % 8)) * A_STRIDE + item / 8;
                d[0] = stage[i].x, d[A_STRIDE] = stage[i].y, d[2 * A_STRIDE] = stage[i].z, d[3 * A_STRIDE] = stage[i].w;
            }
#pragma unroll
            for (int i = 0; i < 20; ++i) {
                const int item = lane + 64 * i;
                float *d = b + (4 * (item % 8)) * B_STRIDE + item / 8;
                d[0] = stage[16 + i].x, d[B_STRIDE] = stage[16 + i].y, d[2 * B_STRIDE] = stage[16 + i].z, d[3 * B_STRIDE] = stage[16 + i].w;
            }
            __syncthreads();
        }
        s = stage[0].x;
    } else {
        for (int t = 0; t < tiles; ++t) {
            const float *ar = As[t & 1] + lhi * A_STRIDE + wave * 32 + l31, *br = Bs[t & 1] + lhi * B_STRIDE + l31;
            float a_cur = ar[0], b_cur[TN];
#pragma unroll
            for (int j = 0; j < TN; ++j) b_cur[j] = br[j * 32];
#pragma unroll
            for (int kk = 0; kk < BK; kk += 2) {
                float a_nxt = a_cur, b_nxt[TN];
#pragma unroll
                for (int j = 0; j < TN; ++j) b_nxt[j] = b_cur[j];
                if (kk + 2 < BK) {
                    a_nxt = ar[(kk + 2) * A_STRIDE];
#pragma unroll
                    for (int j = 0; j < TN; ++j) b_nxt[j] = br[(kk + 2) * B_STRIDE + j * 32];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < TN; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur, b_cur[j], acc[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                a_cur = a_nxt;
#pragma unroll
                for (int j = 0; j < TN; ++j) b_cur[j] = b_nxt[j];
            }
            __syncthreads();
        }
        for (int j = 0; j < TN; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    }
    out[blockIdx.x * 320 + tid] = s;
}

template <int WGS>
void run_split(float *out, const float4 *src, long long quads, int blocks) {
    const int tiles = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe_split<WGS>, dim3(blocks), dim3(320), 0, 0, out, src, tiles, quads);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * tiles * 16 * TN * 4096.0;
    printf("producer/consumer, %d workgroups per CU asked, blocks %4d: %.3f ms  %.1f TFLOP/s (%.2f of 157.3)\n", WGS, blocks, best,
           flop / best / 1e9, flop / best / 1e9 / 157.3);
}

template <int MODE>
void run(float *out, const float4 *src, long long quads, int blocks) {
    const int tiles = 200;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, out, src, tiles, quads);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (ms < best) best = ms;
    }
    const double flop = (double)blocks * 4 * tiles * 16 * TN * 4096.0;
    printf("mode %2d blocks %4d: %.3f ms  %.1f TFLOP/s (%.2f of 157.3)\n", MODE, blocks, best, flop / best / 1e9, flop / best / 1e9 / 157.3);
}

int main() {
    float *out; hipMalloc(&out, 4096 * 320 * 4);
    const long long quads = 64ll << 20;   // 1 GiB of source rows
    float4 *src; hipMalloc(&src, quads * 16); hipMemset(src, 0, quads * 16);
    for (int blocks : {768, 1536}) {
        run<0>(out, src, quads, blocks);
        run<1>(out, src, quads, blocks);
        run<2>(out, src, quads, blocks);
        run<3>(out, src, quads, blocks);
        run<7>(out, src, quads, blocks);
        run<11>(out, src, quads, blocks);
        run<15>(out, src, quads, blocks);
        run<19>(out, src, quads, blocks);
        run<35>(out, src, quads, blocks);
        run<43>(out, src, quads, blocks);
        run<64>(out, src, quads, blocks);
        run<66>(out, src, quads, blocks);
        run<64 + 2 + 128>(out, src, quads, blocks);
        run<64 + 2 + 128 + 8>(out, src, quads, blocks);
        run<64 + 2 + 4 + 8>(out, src, quads, blocks);
        run<256 + 11>(out, src, quads, blocks);
        run<256 + 15>(out, src, quads, blocks);
        run_split<2>(out, src, quads, blocks);
        run_split<3>(out, src, quads, blocks);
    }
    return 0;
}
