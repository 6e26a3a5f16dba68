// Latency probe (GPU box only): dependent global loads at several footprints, with 1 wave and with many.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <numeric>
#include <random>
#include <algorithm>

__global__ void chase(const uint4* __restrict__ buf, int steps, uint32_t start_stride, uint32_t n, uint64_t* out_cycles, uint32_t* sink) {
    uint32_t idx = (uint32_t)(((uint64_t)(blockIdx.x * blockDim.x + threadIdx.x) * start_stride) % n);
    uint64_t t0 = clock64();
    uint32_t acc = 0;
    for (int i = 0; i < steps; ++i) {
        uint4 v = buf[idx];
        idx = v.x;
        acc += v.y;
    }
    uint64_t t1 = clock64();
    if (threadIdx.x == 0) out_cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * blockDim.x + threadIdx.x] = acc + idx;
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("clockRate %d kHz, wallclock rate %d\n", p.clockRate, p.clockInstructionRate);
    for (size_t bytes : {size_t(32) << 10, size_t(1) << 20, size_t(3) << 20, size_t(64) << 20, size_t(1) << 30}) {
        size_t n = bytes / 16;
        std::vector<uint32_t> perm(n);
        std::iota(perm.begin(), perm.end(), 0);
        std::mt19937 rng(1);
        std::shuffle(perm.begin(), perm.end(), rng);
        std::vector<uint4> h(n);
        for (size_t i = 0; i < n; ++i) h[perm[i]] = uint4{perm[(i + 1) % n], (uint32_t)i, 0, 0};
        uint4* d; hipMalloc(&d, bytes); hipMemcpy(d, h.data(), bytes, hipMemcpyHostToDevice);
        for (int blocks : {1, 1563}) {
            uint64_t* cyc; uint32_t* sink;
            hipMalloc(&cyc, blocks * 8); hipMalloc(&sink, blocks * 64 * 4);
            const int steps = 200;
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            chase<<<blocks, 64>>>(d, steps, (uint32_t)(n / (blocks * 64) ? n / (blocks * 64) : 7), (uint32_t)n, cyc, sink);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            chase<<<blocks, 64>>>(d, steps, (uint32_t)(n / (blocks * 64) ? n / (blocks * 64) : 7), (uint32_t)n, cyc, sink);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            std::vector<uint64_t> hc(blocks); hipMemcpy(hc.data(), cyc, blocks * 8, hipMemcpyDeviceToHost);
            double avg = 0; for (auto c : hc) avg += c; avg /= blocks;
            printf("footprint %8zu KB  waves %5d : %.0f clock64-ticks/step, kernel %.1f us -> %.0f ns/step\n", bytes >> 10, blocks,
                   avg / steps, ms * 1e3, ms * 1e6 / steps);
            hipFree(cyc); hipFree(sink);
        }
        hipFree(d);
    }
    return 0;
}
