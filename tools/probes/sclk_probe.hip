// sclk_probe.hip - the shader clock a small kernel actually runs at (gfx950): s_memtime (shader clock) against s_memrealtime
// (100 MHz) around a fixed chain of dependent operations, for a tiny grid after idling, in a stream of tiny kernels, and
// right after a chip-filling kernel.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/sclk tools/probes/sclk_probe.hip && /tmp/sclk
#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_chain(long long* out, int iters, double seed) {
    const long long c0 = clock64(), w0 = wall_clock64();
    double x = seed;
    for (int i = 0; i < iters; ++i) x = x * 1.0000001 + 1e-9;  // dependent f64 chain
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = c1 - c0; out[1] = w1 - w0; out[2] = (long long)x; }
}
__global__ void k_fill(double* buf, size_t n) {
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    double a = 0;
    for (int r = 0; r < 64; ++r) for (size_t j = i; j < n; j += (size_t)gridDim.x * blockDim.x) a += buf[j];
    if (a == 12345.678) buf[0] = a;
}
int main() {
    long long *d, h[3];
    double* buf;
    const size_t n = 64 << 20;
    CK(hipMalloc(&d, 64)); CK(hipMalloc(&buf, n * 8)); CK(hipMemset(buf, 0, n * 8));
    auto run = [&](const char* what, int grid) {
        hipLaunchKernelGGL(k_chain, dim3(grid), dim3(256), 0, 0, d, 20000, 1.0);
        hipMemcpy(h, d, 24, hipMemcpyDeviceToHost);
        printf("%-44s chain of 20000: %7.1f us, shader clock %.0f MHz (cycles/iter %.1f)\n", what, h[1] / 100.0, h[0] / (h[1] / 100.0), (double)h[0] / 20000);
    };
    usleep(300000);
    run("after 0.3 s idle, 1 workgroup", 1);
    run("again", 1);
    for (int i = 0; i < 2000; ++i) hipLaunchKernelGGL(k_chain, dim3(2), dim3(256), 0, 0, d, 2000, 1.0);
    run("after 2000 tiny kernels, 2 workgroups", 2);
    hipLaunchKernelGGL(k_fill, dim3(4096), dim3(256), 0, 0, buf, n);
    run("right after a chip-filling kernel", 1);
    run("chip-wide chain (2048 workgroups)", 2048);
    return 0;
}
