// xcd_probe.hip - can the waves of ONE XCD hand data to each other inside a kernel without the agent-scope release
// (buffer_wbl2) that made the fused front + tail experiment of round 1 slow?  Producers (the first NB workgroups of each XCD by
// arrival) write a table with plain stores, wait for their stores (s_waitcnt vmcnt(0): workgroup-scope release) and bump a
// per-XCD counter with an L2 atomic; every workgroup of that XCD spins on the counter, invalidates its L1 (agent-scope
// acquire) and reads the table.  Reports wrong values and the kernel time with and without the hand-over.
// build: hipcc --offload-arch=gfx950 -O3 -o xcd_probe xcd_probe.hip ; run on an MI355X
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
constexpr int NB = 25, TAB = 4096, NX = 8, SLOTS = 64;
__device__ __forceinline__ unsigned xcc_id() { return __builtin_amdgcn_s_getreg((4 - 1) << 11 | 0 << 6 | 20) & 15u; }
__global__ __launch_bounds__(256) void k_probe(int frame, int* __restrict__ ctr, double* __restrict__ tab, int* __restrict__ errs,
                                               int* __restrict__ xcd_hist, int sync) {
    __shared__ int s_rank;
    const unsigned x = xcc_id();
    int* arrive = ctr + ((frame % SLOTS) * NX + x) * 2;
    int* done = arrive + 1;
    if (blockIdx.x == 0 && threadIdx.x < 2 * NX) ctr[(((frame + SLOTS / 2) % SLOTS) * NX) * 2 + threadIdx.x] = 0;  // a future frame's counters
    if (threadIdx.x == 0) {
        s_rank = atomicAdd(arrive, 1);
        if (frame == 0) atomicAdd(&xcd_hist[x], 1);
    }
    __syncthreads();
    const int rank = s_rank;
    double* mytab = tab + (size_t)x * NB * TAB;
    if (sync && rank < NB) {
        for (int i = threadIdx.x; i < TAB; i += 256) mytab[(size_t)rank * TAB + i] = (double)(frame * 1000003 + rank * 4099 + i);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");  // s_waitcnt vmcnt(0): the stores are in this XCD's L2
        __syncthreads();
        if (threadIdx.x == 0) atomicAdd(done, 1);
    }
    if (sync) {
        if (threadIdx.x == 0) {
            int spins = 0;  // bounded: a broken hand-over must end the kernel, not hang the GPU
            while (__hip_atomic_load(done, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < NB && ++spins < 2000000) __builtin_amdgcn_s_sleep(2);
            if (spins >= 2000000) atomicAdd(errs + 1, 1);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");  // buffer_inv: no stale L1 lines of the table
        int bad = 0;
        for (int k = 0; k < 4; ++k) {
            const int r = (blockIdx.x + 7 * k) % NB, i = (threadIdx.x * 16 + 97 * k + blockIdx.x) % TAB;
            const double v = mytab[(size_t)r * TAB + i];
            bad += v != (double)(frame * 1000003 + r * 4099 + i);
        }
        if (bad) atomicAdd(errs, bad);
    }
}
int main() {
    int *ctr, *errs, *hist; double* tab;
    hipMalloc(&ctr, SLOTS * NX * 2 * sizeof(int)); hipMemset(ctr, 0, SLOTS * NX * 2 * sizeof(int));
    hipMalloc(&errs, 2 * sizeof(int)); hipMemset(errs, 0, 2 * sizeof(int));
    hipMalloc(&hist, 16 * sizeof(int)); hipMemset(hist, 0, 16 * sizeof(int));
    hipMalloc(&tab, (size_t)NX * NB * TAB * sizeof(double));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int grid : {391, 1563}) {
        for (int sync : {0, 1}) {
            hipMemset(ctr, 0, SLOTS * NX * 2 * sizeof(int));
            for (int f = 0; f < 20; ++f) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, f, ctr, tab, errs, hist, sync);
            hipDeviceSynchronize();
            hipEventRecord(e0);
            for (int f = 20; f < 520; ++f) hipLaunchKernelGGL(k_probe, dim3(grid), dim3(256), 0, 0, f, ctr, tab, errs, hist, sync);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            int h_err[2]; hipMemcpy(h_err, errs, 2 * sizeof(int), hipMemcpyDeviceToHost);
            printf("grid %4d sync %d: %.2f us per launch, wrong values so far %d, spin time-outs %d\n", grid, sync, ms * 1e3 / 500, h_err[0], h_err[1]);
            fflush(stdout);
        }
    }
    int h[16]; hipMemcpy(h, hist, sizeof(h), hipMemcpyDeviceToHost);
    printf("workgroups per XCC_ID in the first launch:"); for (int i = 0; i < 16; ++i) printf(" %d", h[i]); printf("\n");
    return 0;
}
