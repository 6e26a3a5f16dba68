// launch_vs_barrier.hip - what a launch boundary costs against the ways of staying inside one launch (gfx950).
//
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lvb tools/probes/launch_vs_barrier.hip && /tmp/lvb
//
// Each variant runs a chain of S dependent stages over G workgroups of 256 threads; a stage is one dependent global trip (read
// what the previous stage wrote, add, write).  (a) S launches; (b) one launch, a grid barrier (ticket + spin, agent-scope
// release / acquire) between stages; (c) one launch, narrow producer (1 workgroup) -> wide consumers spinning on a flag.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_stage(const double* __restrict__ in, double* __restrict__ out, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    out[i] = in[(i + 257) % n] + 1.0;
}

__device__ inline void grid_barrier(unsigned* ctr, unsigned target) {
    __syncthreads();
    if (threadIdx.x == 0) {
        __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
    }
    __syncthreads();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
}

__global__ __launch_bounds__(256) void k_chain(double* a, double* b, int n, int stages, unsigned* ctr) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double *in = a, *out = b;
    for (int s = 0; s < stages; ++s) {
        out[i] = in[(i + 257) % n] + 1.0;
        __threadfence();
        grid_barrier(ctr, (unsigned)(s + 1) * gridDim.x);
        double* t = in; in = out; out = t;
    }
}

// producer / consumers: workgroup 0 of each stage pair writes 16 values + flag, the others wait for the flag and read them
__global__ __launch_bounds__(256) void k_flag(double* a, double* b, int n, int stages, unsigned* flag) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    double acc = 0.0;
    for (int s = 0; s < stages; ++s) {
        if (blockIdx.x == 0) {
            if (threadIdx.x < 16) a[s * 16 + threadIdx.x] = acc + threadIdx.x;
            __threadfence();
            __syncthreads();
            if (threadIdx.x == 0) __hip_atomic_store(flag + s, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (threadIdx.x == 0)
            while (__hip_atomic_load(flag + s, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) == 0u) __builtin_amdgcn_s_sleep(1);
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        acc += __builtin_nontemporal_load(a + s * 16 + (threadIdx.x & 15));
        // the next stage's producer needs every consumer's value?  no: a chain through workgroup 0 only (the pattern of xe -> weights)
    }
    b[i] = acc;
}

int main() {
    const int S = 64;
    hipStream_t st;
    CK(hipStreamCreate(&st));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    for (int G : {2, 8, 32, 64, 128, 400}) {
        const int n = G * 256;
        double *a, *b;
        unsigned* ctr;
        CK(hipMalloc(&a, n * sizeof(double) + 65536));
        CK(hipMalloc(&b, n * sizeof(double)));
        CK(hipMalloc(&ctr, 4096));
        CK(hipMemset(a, 0, n * sizeof(double)));
        float ms_launch = 0, ms_bar = 0, ms_flag = 0;
        for (int rep = 0; rep < 3; ++rep) {
            CK(hipEventRecord(e0, st));
            for (int s = 0; s < S; ++s) hipLaunchKernelGGL(k_stage, dim3(G), dim3(256), 0, st, s & 1 ? b : a, s & 1 ? a : b, n);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_launch, e0, e1));
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_chain, dim3(G), dim3(256), 0, st, a, b, n, S, ctr);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_bar, e0, e1));
            CK(hipMemsetAsync(ctr, 0, 4096, st));
            CK(hipEventRecord(e0, st));
            hipLaunchKernelGGL(k_flag, dim3(G), dim3(256), 0, st, a, b, n, S, ctr);
            CK(hipEventRecord(e1, st));
            CK(hipEventSynchronize(e1));
            CK(hipEventElapsedTime(&ms_flag, e0, e1));
        }
        printf("G %4d  per stage: launch %.2f us  grid barrier %.2f us  flag %.2f us\n", G, ms_launch * 1000 / S, ms_bar * 1000 / S,
               ms_flag * 1000 / S);
        CK(hipFree(a)); CK(hipFree(b)); CK(hipFree(ctr));
    }
    return 0;
}
