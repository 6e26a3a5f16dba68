// Vector-cache lookup rate for per-lane loads of the shapes the list scans issue (L1-/L2-resident footprints).
// build: hipcc --offload-arch=gfx950 -O3 -o ta_probe ta_probe.hip ; run on the GPU box: ./ta_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int SHAPE, int W>  // W: bytes per lane (16, 8, 4)
__global__ __launch_bounds__(64) void k_probe(const char* __restrict__ base, unsigned mask, int iters, float* out) {
    const int lane = threadIdx.x;
    unsigned off;
    switch (SHAPE) {
        case 0: off = lane * 128u; break;                                  // every lane its own line
        case 1: off = lane * 16u; break;                                   // contiguous 16-byte pieces
        case 2: off = lane * 32u; break;                                   // stride 32 (first halves of consecutive 32-byte records)
        case 3: off = (lane >> 1) * 128u + (lane & 1) * 16u; break;        // lane pairs share a line
        case 4: off = (lane >> 3) * 1024u + (lane & 7) * 32u; break;       // eight groups, each 8 consecutive records (stride 32)
        case 5: off = (lane >> 3) * 1024u + (lane & 7) * 16u; break;       // eight groups, each 8 consecutive pieces
        default: off = 0; break;                                            // 6: all lanes the same address
    }
    off += blockIdx.x * 4096u;  // waves of a CU start at different places of the footprint
    float acc = 0.f;
    unsigned step = 8192u + 128u;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        unsigned o = off;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const char* p = base + ((o + u * 1024u * 5u) & mask);
            if (W == 16) { float4 v = *reinterpret_cast<const float4*>(p); acc += v.x + v.w; }
            else if (W == 8) { float2 v = *reinterpret_cast<const float2*>(p); acc += v.x + v.y; }
            else { acc += *reinterpret_cast<const float*>(p); }
        }
        off += step;
    }
    if (acc == 123.456f) out[0] = acc;
}

template <int SHAPE, int W>
static void run(const char* name, const char* buf, unsigned mask, int waves_per_cu, float* out) {
    const int iters = 2000, grid = 256 * waves_per_cu;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k_probe<SHAPE, W>), dim3(grid), dim3(64), 0, 0, buf, mask, 200, out);
    hipEventRecord(a);
    hipLaunchKernelGGL((k_probe<SHAPE, W>), dim3(grid), dim3(64), 0, 0, buf, mask, iters, out);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lane_loads = (double)grid * iters * 8 * 64;
    const double cyc = ms * 1e-3 * 2.4e9;
    printf("%-44s W=%2d waves/CU=%2d footprint=%8u B: %.3f ms, %.2f lane-loads/cycle/CU, %.1f B/cycle/CU\n", name, W, waves_per_cu, mask + 1,
           ms, lane_loads / cyc / 256.0, lane_loads * W / cyc / 256.0);
}

int main() {
    char* buf; float* out;
    const size_t bytes = 64u << 20;
    hipMalloc(&buf, bytes); hipMemset(buf, 0, bytes); hipMalloc(&out, 64);
    for (unsigned fp : {16u << 10, 1u << 20, 32u << 20}) {
        const unsigned mask = fp - 1;
        for (int w : {4, 10}) {
            run<0, 16>("own line per lane", buf, mask, w, out);
            run<0, 8>("own line per lane", buf, mask, w, out);
            run<0, 4>("own line per lane", buf, mask, w, out);
            run<1, 16>("contiguous pieces", buf, mask, w, out);
            run<2, 16>("stride 32", buf, mask, w, out);
            run<3, 16>("lane pairs share a line", buf, mask, w, out);
            run<4, 16>("8 groups x 8 records (stride 32)", buf, mask, w, out);
            run<5, 16>("8 groups x 8 contiguous pieces", buf, mask, w, out);
            run<6, 16>("same address", buf, mask, w, out);
        }
    }
    return 0;
}
