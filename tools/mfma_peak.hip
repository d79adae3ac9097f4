// Stand-alone probe: sustained v_mfma_f32_32x32x2_f32 rate of the chip with operands in
// registers (no memory traffic), 1 or 2 waves per SIMD.  Build: hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a0, float b0, int rnd) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
    // MODE 0: constant operands; MODE 1: pseudo-random operands that change every MFMA
    float av[8], bv[NACC];
    unsigned h = (threadIdx.x + 1) * 2654435761u + blockIdx.x * 40503u;
    for (int q = 0; q < 8; ++q) { h = h * 1664525u + 1013904223u; av[q] = rnd ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : a0; }
    for (int q = 0; q < NACC; ++q) { h = h * 1664525u + 1013904223u; bv[q] = rnd ? ((int)(h >> 8) - (1 << 23)) * (1.0f / (1 << 23)) : b0; }
    for (int i = 0; i < iters; i += 8) {
#pragma unroll
        for (int q = 0; q < 8; ++q)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[q], bv[(j + q) % NACC], acc[j], 0, 0, 0);
    }
    float s = 0.f;
    for (int j = 0; j < NACC; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
int main() {
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rnd = 0; rnd < 2; ++rnd)
    for (int blocks : {512, 1024}) {
        const int iters = 8000;
        for (int rep = 0; rep < 4; ++rep) {
            hipEventRecord(e0);
            hipLaunchKernelGGL(k<10>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0f, 2.0f, rnd);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            double flop = (double)blocks * 4 * iters * 10 * 4096.0;
            printf("random %d blocks %d: %.3f ms  %.1f TFLOP/s\n", rnd, blocks, ms, flop / ms / 1e9);
        }
    }
    return 0;
}
