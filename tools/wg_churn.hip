// Stand-alone probe: what the launch structure of the conv products costs with NOTHING but
// their MFMAs.  G workgroups of 4 waves, each wave issuing `iters` x 5 independent
// v_mfma_f32_32x32x2_f32 (750 per wave = a 128 x 160 tile over K = 300), at 2, 3 or 4
// workgroups per CU; and the same total work as a persistent grid (one workgroup per slot
// looping over its share).  Build: hipcc --offload-arch=gfx950 -O3 tools/wg_churn.hip -o exp_libs/wg_churn
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int OCC>
__global__ __launch_bounds__(256, OCC) void churn(float *out, int iters, int tiles_per_wg) {
    float av = threadIdx.x * 1e-3f, bv[5];
    for (int j = 0; j < 5; ++j) bv[j] = (threadIdx.x + j) * 3e-4f;
    float s = 0.f;
    for (int t = 0; t < tiles_per_wg; ++t) {
        f32x16 acc[5];
        for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) acc[j][r] = 0.f;
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int j = 0; j < 5; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av, bv[j], acc[j], 0, 0, 0);
        }
        for (int j = 0; j < 5; ++j) for (int r = 0; r < 16; ++r) s += acc[j][r];
    }
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int OCC>
void run(float *out, int wgs, int iters, int tiles, const char *what) {
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e9;
    for (int rep = 0; rep < 6; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL(churn<OCC>, dim3(wgs), dim3(256), 0, 0, out, iters, tiles);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep > 1 && ms < best) best = ms;
    }
    double flop = (double)wgs * tiles * 4 * iters * 5 * 4096.0;
    printf("%-28s occ %d wgs %5d x %d tiles, %4d mfma/acc: %8.1f us  %6.1f TFLOP/s\n", what, OCC, wgs, tiles, iters,
           best * 1e3, flop / best / 1e9);
}

int main() {
    float *out; hipMalloc(&out, 8192 * 256 * 4);
    run<2>(out, 1488, 150, 1, "one tile per workgroup");
    run<3>(out, 1488, 150, 1, "one tile per workgroup");
    run<4>(out, 1488, 150, 1, "one tile per workgroup");
    run<2>(out, 512, 150, 3, "persistent, 3 tiles each");
    run<3>(out, 768, 150, 2, "persistent, 2 tiles each");
    run<1>(out, 256, 150, 6, "persistent, 6 tiles each");
    run<2>(out, 512, 4500, 1, "long tiles (30x)");
    run<3>(out, 768, 3000, 1, "long tiles (20x)");
    run<1>(out, 256, 9000, 1, "long tiles (60x)");
    return 0;
}
