// mfma_probe.hip - achievable v_mfma_f32_16x16x4_f32 rate (no memory traffic): sets the ceiling k_score_mfma is priced against.
// build: hipcc --offload-arch=gfx950 -O3 -o mfma_probe mfma_probe.hip ; run on the GPU box
#include <hip/hip_runtime.h>
#include <cstdio>
using f32x4 = __attribute__((ext_vector_type(4))) float;
template <int NACC>
__global__ __launch_bounds__(1024) void k(float* out, int iters, float a0, float b0) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
    }
    float s = 0.f;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    if (s == 12345.678f) out[0] = s;
}
template <int NACC>
void run(int waves_per_cu_blocks, int threads, int iters) {
    float* out; hipMalloc(&out, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<waves_per_cu_blocks, threads>>>(out, iters, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<waves_per_cu_blocks, threads>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double mfmas = (double)waves_per_cu_blocks * (threads / 64) * iters * 4.0 * NACC;
    printf("NACC=%d blocks=%d threads=%d iters=%d: %.1f us, %.1f TFLOP/s\n", NACC, waves_per_cu_blocks, threads, iters, ms * 1e3,
           mfmas * 2048.0 / (ms * 1e-3) / 1e12);
}
int main() {
    run<4>(256, 1024, 512);   // 16 waves per CU, 4 independent accumulators (the kernel's shape)
    run<4>(196, 1024, 512);   // the kernel's grid
    run<4>(256, 256, 2048);   // 4 waves per CU
    run<1>(256, 1024, 2048);  // dependent chain
    run<8>(256, 1024, 256);
    return 0;
}
