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
// is sequential from block to block (624 words) but parallel inside one in three phases (k < 227 reads only old words,
// 227 <= k < 454 reads the new words of phase one, the rest those of phase two), so a block costs three barriers; a block's
// 312 doubles are written out beside the first phase of the next block.
// Pure 32-bit integer arithmetic: the words equal at::mt19937's bit for bit (tests: against torch.rand / torch.manual_seed).
#include "midas_internal.hpp"

namespace midas {

constexpr int MT_N = 624, MT_M = 397;
constexpr uint32_t MT_MATRIX_A = 0x9908b0dfu, MT_UPPER = 0x80000000u, MT_LOWER = 0x7fffffffu;

__device__ __forceinline__ uint32_t mt_mix(uint32_t a, uint32_t b, uint32_t far) {
    const uint32_t y = (a & MT_UPPER) | (b & MT_LOWER);
    return far ^ (y >> 1) ^ ((y & 1u) ? MT_MATRIX_A : 0u);
}
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

// One workgroup of MT_THREADS threads.  The two blocks live in ONE LDS array indexed by a toggle (two arrays behind swapped
// pointers made the compiler address them with flat instructions).  A block costs three barriers: the doubles of the block
// before it are written out beside phase one of the next (they read the old buffer only).
constexpr int MT_THREADS = 320, MT_P = MT_N - MT_M;  // 227 words per phase
struct MtEmit { int pos, end; long long gw; bool more; };  // words [pos, end) of the block; gw = stream index of word `pos`; more: the stream goes on

__global__ __launch_bounds__(MT_THREADS) void k_mt_rand64(uint32_t* __restrict__ state, long long skip, long long N, double* __restrict__ out) {
    __shared__ uint32_t s_mt[2][MT_N];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x;
    for (int k = t; k < MT_N; k += MT_THREADS) s_mt[0][k] = state[k];
    int pos = (int)state[MT_N];
    int b = 0;
    __syncthreads();
    auto phase_a = [&]() { if (t < MT_P) s_mt[b ^ 1][t] = mt_mix(s_mt[b][t], s_mt[b][t + 1], s_mt[b][t + MT_M]); };
    auto phase_b = [&]() { if (t < MT_P) { const int k = t + MT_P; s_mt[b ^ 1][k] = mt_mix(s_mt[b][k], s_mt[b][k + 1], s_mt[b ^ 1][t]); } };
    auto phase_c = [&]() {
        const int k = t + 2 * MT_P;
        if (k < MT_N - 1) s_mt[b ^ 1][k] = mt_mix(s_mt[b][k], s_mt[b][k + 1], s_mt[b ^ 1][k - MT_P]);
        else if (k == MT_N - 1) s_mt[b ^ 1][k] = mt_mix(s_mt[b][k], s_mt[b ^ 1][0], s_mt[b ^ 1][MT_M - 1]);
    };
    // 2 N words -> N doubles: (hi << 32 | lo) & (2^53 - 1), times 2^-53 (at::uniform_real_distribution<double>); hi = even stream index
    auto emit = [&](const MtEmit& e) {
        const uint32_t* cur = s_mt[b];
        const int odd = (int)(e.gw & 1);
        if (odd && t == MT_THREADS - 1) {  // the block starts with the low word of a value whose high word ended the block before
            const unsigned long long r = (((unsigned long long)s_carry << 32) | mt_temper(cur[e.pos])) & ((1ull << 53) - 1ull);
            out[e.gw >> 1] = (double)r * 1.1102230246251565e-16;
        }
        const int o = e.pos + odd + 2 * t;
        if (o < e.end) {
            const uint32_t hi = mt_temper(cur[o]);
            if (o + 1 < e.end) {
                const unsigned long long r = (((unsigned long long)hi << 32) | mt_temper(cur[o + 1])) & ((1ull << 53) - 1ull);
                out[(e.gw + (o - e.pos)) >> 1] = (double)r * 1.1102230246251565e-16;
            } else {
                s_carry = hi;  // its partner is the first word of the next block
            }
        }
    };
    // skip: whole blocks are twisted over, the rest is an offset
    while (skip > 0) {
        const int avail = MT_N - pos;
        if (skip >= avail) {
            skip -= avail;
            phase_a(); __syncthreads(); phase_b(); __syncthreads(); phase_c(); __syncthreads();
            b ^= 1;
            pos = 0;
        } else {
            pos += (int)skip;
            skip = 0;
        }
    }
    const long long W = 2 * N;
    long long gw = 0;
    bool pend = false;
    MtEmit pe{0, 0, 0, false};
    while (true) {
        if (pos == MT_N && gw < W) {
            phase_a();
            if (pend) { emit(pe); pend = false; }
            __syncthreads(); phase_b(); __syncthreads(); phase_c(); __syncthreads();
            b ^= 1;
            pos = 0;
        }
        if (gw >= W) break;
        const long long left = W - gw;
        const int m = (int)(left < (long long)(MT_N - pos) ? left : (long long)(MT_N - pos));
        pe.pos = pos; pe.end = pos + m; pe.gw = gw; pe.more = left > m;
        pend = true;
        gw += m;
        pos += m;
    }
    if (pend) emit(pe);
    __syncthreads();
    for (int k = t; k < MT_N; k += MT_THREADS) state[k] = s_mt[b][k];
    if (t == 0) state[MT_N] = (uint32_t)pos;
}

int launch_mt_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state) {
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(1), 0, ctx->stream, (uint32_t)(seed & 0xffffffffull), state);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_mt_rand64(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t N, double* out) {
    if (N == 0 && skip_words == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_mt_rand64, dim3(1), dim3(MT_THREADS), 0, ctx->stream, state, (long long)skip_words, (long long)N, out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(mt19937, k_mt_seed)

}  // namespace midas
