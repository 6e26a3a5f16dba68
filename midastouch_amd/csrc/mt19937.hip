// mt19937.hip - torch's CPU random stream on the device.
//
// The reference draws on torch's default CPU generator (at::mt19937, ATen/core/MT19937RNGEngine.h): torch.manual_seed(s)
// seeds it with the low 32 bits of s; the resampler's WeightedRandomSampler -> torch.multinomial(weights.double(), N, True)
// (modules/particle_filter.py:245) consumes, per sample, one random64() = two 32-bit outputs (hi word first) masked to 53
// bits and scaled by 2^-53 - exactly torch.rand(N, dtype=float64) (verified there); the two torch.normal
// calls of add_noise_to_odom (:326-335) consume one 32-bit output per float32 value (+ 16 when the size is not a multiple of
// 16: ATen's normal_fill recomputes the last 16).  "Bit-exact resample indices under a fixed seed" therefore needs this
// stream; round 2 took it from the host every frame (0.8 MB of uniforms over PCIe, 1.0 - 1.4k frames/s).  Here the
// generator's state lives in device memory and a single workgroup advances it: the recurrence
//     x[k+624] = x[k+397] ^ twist(x[k], x[k+1])
// is sequential from block to block (624 words); see k_mt_blocks for how a block is one barrier.
#include "midas_internal.hpp"

namespace midas {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_MATRIX_A = 0x9908b0dfu, MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu;

__device__ __forceinline__ uint32_t mt_temper(uint32_t y) {
    y ^= y >> 11;
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= y >> 18;
    return y;
}

// state: [0, 624) the current block (already twisted), [624] = words of it consumed so far (624: a new block is due)
__global__ __launch_bounds__(1) void k_mt_seed(uint32_t seed, uint32_t* __restrict__ state) {
    uint32_t x = seed;
    state[0] = x;
    for (int j = 1; j < MT_N; ++j) {
        x = 1812433253u * (x ^ (x >> 30)) + (uint32_t)j;
        state[j] = x;
    }
    state[MT_N] = MT_N;  // at::mt19937 twists before its first output
    state[MT_N + 1] = 0;
}

// ---- the block recurrence: ONE barrier per 624-word block --------------------------------------------------------------------
// Word k of the next block is n[k] = n[k - 227] ^ twist(o[k], o[k + 1]) for k >= 227 (o = the block before, n[k - 227] a word of
// the NEW block) and n[k] = o[k + 397] ^ twist(o[k], o[k + 1]) below: the chain runs along "columns" c, c + 227, c + 454, and the
// twists read old words only.  So thread c (c < 227) owns a column and walks its chain in registers - three twists, seven LDS
// reads, no redundancy, no barrier inside the block (round 3 had one thread per word and a barrier after each third of the block:
// three barriers and three LDS round trips per block, ~0.65 us a block, 210 us for a frame's 2 N = 200 000 words).  The last word
// is the one exception: n[623] = n[396] ^ twist(o[623], n[0]) needs the new n[0], which its thread (c = 169) rebuilds from three
// more old words.  Four waves, one per SIMD.
// The words leave the kernel RAW (untempered, 4 bytes each, coalesced per column group); tempering, pairing into 53-bit values
// and the float64 conversion are off the sequential chain - k_mt_emit does them with the whole chip.
// Pure 32-bit integer arithmetic: the words equal at::mt19937's bit for bit (tests: against torch.rand / torch.manual_seed).
constexpr int MT_COLS = MT_N - MT_M;  // 227 columns
constexpr int MT_THREADS = 256;
static_assert(MT_COLS <= MT_THREADS && MT_THREADS % 64 == 0, "one thread per column");
static_assert(2 * MT_COLS + 169 == MT_N - 1, "column 169 ends in the block's last word");

__device__ __forceinline__ uint32_t mt_twist(uint32_t a, uint32_t b) {
    const uint32_t y = (a & MT_UPPER) | (b & MT_LOWER);
    return (y >> 1) ^ ((y & 1u) ? MT_MATRIX_A : 0u);
}

// state: [0, 624) the current block (already twisted), [624] = words of it consumed so far (624: a new block is due).
// Relative stream index r: word r % 624 of block r / 624, block 0 = the stored one.  The call consumes [pos, pos + skip + nwords)
// and hands out the last nwords of them: raw[(b - b0) * 624 + k] for the blocks b >= b0 = (pos + skip) / 624, meta[0] = the
// offset of the first wanted word in raw.
__global__ __launch_bounds__(MT_THREADS) void k_mt_blocks(uint32_t* __restrict__ state, long long skip, long long nwords,
                                                          uint32_t* __restrict__ raw, int32_t* __restrict__ meta) {
    __shared__ uint32_t s_mt[2][MT_N];
    const int c = threadIdx.x;
    for (int k = c; k < MT_N; k += MT_THREADS) s_mt[0][k] = state[k];
    const long long pos = (long long)state[MT_N];
    const long long w0 = pos + skip, e = w0 + nwords;
    if (e == 0) return;                       // nothing consumed, nothing to write (uniform)
    const long long b0 = w0 / MT_N, bl = (e - 1) / MT_N;
    const bool wanted = nwords > 0;
    if (c == 0 && meta) meta[0] = (int32_t)(w0 - b0 * MT_N);
    __syncthreads();
    if (wanted && b0 == 0)
        for (int k = c; k < MT_N; k += MT_THREADS) raw[k] = s_mt[0][k];
    // iteration b builds block b + 1 from block b; the first n_silent of them are stepped over (skip), the rest leave their words
    // in raw.  Two loops, so that the inner one tests nothing per block but its counter.
    const long long n_silent = !wanted ? bl : (b0 > 1 ? (b0 - 1 < bl ? b0 - 1 : bl) : 0);
    auto block = [&](const uint32_t* __restrict__ o, uint32_t* __restrict__ n, uint32_t* __restrict__ r) {
        if (c < MT_COLS) {
            // ten LDS reads, all issued before the first wait: no branch between them (the column's third word and the new
            // n[0] of the last word's twist are read by every thread - clamped resp. broadcast addresses - and selected after)
            const bool third = c + 2 * MT_COLS < MT_N;           // c <= 169
            const bool last = c + 2 * MT_COLS == MT_N - 1;       // c == 169
            const int i2 = third ? c + 2 * MT_COLS : 0, i3 = (third && !last) ? c + 2 * MT_COLS + 1 : 0;
            const uint32_t a0 = o[c], a1 = o[c + 1], far = o[c + MT_M];
            const uint32_t b0w = o[c + MT_COLS], b1w = o[c + MT_COLS + 1];
            const uint32_t c0w = o[i2], c1r = o[i3];
            uint32_t z0 = o[0], z1 = o[1], zm = o[MT_M];
            asm volatile("" : "+v"(z0), "+v"(z1), "+v"(zm));  // (pinned: the compiler otherwise sinks them into a branch of lane 169)
            uint32_t nz = zm ^ mt_twist(z0, z1);              // the new n[0]
            asm volatile("" : "+v"(nz));
            const uint32_t c1w = last ? nz : c1r;
            const uint32_t n0 = far ^ mt_twist(a0, a1);
            const uint32_t n1 = n0 ^ mt_twist(b0w, b1w);
            const uint32_t n2 = n1 ^ mt_twist(c0w, c1w);
            n[c] = n0;
            n[c + MT_COLS] = n1;
            if (third) n[c + 2 * MT_COLS] = n2;
            if (r) {
                r[c] = n0;
                r[c + MT_COLS] = n1;
                if (third) r[c + 2 * MT_COLS] = n2;
            }
        }
        // the barrier orders the LDS words only: __syncthreads() would also drain this wave's global stores (s_waitcnt vmcnt(0) in
        // front of s_barrier) - a round trip to memory per block.  The raw words are written once and read by the next kernel.
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    int par = 0;
    for (int k = (int)(n_silent < 0x7fffffff ? n_silent : 0x7fffffff), done = 0; done < k; ++done, par ^= 1) block(s_mt[par], s_mt[par ^ 1], nullptr);
    for (long long bb = 0x7fffffff; bb < n_silent; ++bb, par ^= 1) block(s_mt[par], s_mt[par ^ 1], nullptr);  // (skips beyond 2^31 blocks)
    {
        uint32_t* r = raw + (n_silent + 1 - b0) * MT_N;
        for (int k = (int)(bl - n_silent), done = 0; done < k; ++done, par ^= 1, r += MT_N) block(s_mt[par], s_mt[par ^ 1], r);
    }
    const uint32_t* fin = s_mt[bl & 1];
    for (int k = c; k < MT_N; k += MT_THREADS) state[k] = fin[k];
    if (c == 0) state[MT_N] = (uint32_t)(e - bl * MT_N);
}

// 2 N words -> N doubles: (hi << 32 | lo) & (2^53 - 1), times 2^-53 (at::uniform_real_distribution<double>); hi = the first word
__global__ __launch_bounds__(256) void k_mt_emit(const uint32_t* __restrict__ raw, const int32_t* __restrict__ meta, long long N,
                                                 double* __restrict__ out) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t* w = raw + meta[0] + 2 * i;
    const unsigned long long r = (((unsigned long long)mt_temper(w[0]) << 32) | mt_temper(w[1])) & ((1ull << 53) - 1ull);
    out[i] = (double)r * 1.1102230246251565e-16;
}

int launch_mt_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state) {
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(1), 0, ctx->stream, (uint32_t)(seed & 0xffffffffull), state);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_mt_rand64(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t N, double* out) {
    if (N == 0 && skip_words == 0) return MIDAS_OK;
    uint32_t* raw = nullptr;
    int32_t* meta = nullptr;
    if (N > 0) {
        // blocks b0 .. bl: at most the 2 N words, the rest of the block they start in and of the one they end in
        void* p;
        int rc = midas_scratch(ctx, ((size_t)2 * N + 3 * MT_N) * sizeof(uint32_t), &p);
        if (rc) return rc;
        raw = (uint32_t*)p;
        if ((rc = midas_scratch(ctx, 64, &p))) return rc;
        meta = (int32_t*)p;
    }
    hipLaunchKernelGGL(k_mt_blocks, dim3(1), dim3(MT_THREADS), 0, ctx->stream, state, (long long)skip_words, (long long)(2 * N), raw, meta);
    if (N > 0) hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, raw, meta, (long long)N, out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(mt19937, k_mt_seed)

}  // namespace midas
