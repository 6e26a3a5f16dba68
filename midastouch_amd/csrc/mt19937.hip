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
// 227 <= k < 454 reads the new words of phase one, the rest those of phase two), so a block costs three barriers.
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

// next block of 624 words: old -> nw (two LDS buffers: no word is overwritten while somebody may still read it)
__device__ __forceinline__ void mt_twist(const uint32_t* old, uint32_t* nw) {
    const int t = threadIdx.x;
    if (t < MT_N - MT_M) nw[t] = mt_mix(old[t], old[t + 1], old[t + MT_M]);
    __syncthreads();
    if (t < MT_N - MT_M) { const int k = t + (MT_N - MT_M); nw[k] = mt_mix(old[k], old[k + 1], nw[k - (MT_N - MT_M)]); }
    __syncthreads();
    {
        const int k = t + 2 * (MT_N - MT_M);
        if (k < MT_N - 1) nw[k] = mt_mix(old[k], old[k + 1], nw[k - (MT_N - MT_M)]);
        else if (k == MT_N - 1) nw[k] = mt_mix(old[k], nw[0], nw[MT_M - 1]);
    }
    __syncthreads();
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

__global__ __launch_bounds__(256) void k_mt_rand64(uint32_t* __restrict__ state, long long skip, long long N, double* __restrict__ out) {
    __shared__ uint32_t s_a[MT_N], s_b[MT_N];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x;
    for (int k = t; k < MT_N; k += 256) s_a[k] = state[k];
    int pos = (int)state[MT_N];
    __syncthreads();
    uint32_t *cur = s_a, *nxt = s_b;
    // skip: whole blocks are twisted over, the rest is an offset
    while (skip > 0) {
        const int avail = MT_N - pos;
        if (skip >= avail) {
            skip -= avail;
            mt_twist(cur, nxt);
            uint32_t* sw = cur; cur = nxt; nxt = sw;
            pos = 0;
        } else {
            pos += (int)skip;
            skip = 0;
        }
    }
    // 2 N words -> N doubles: (hi << 32 | lo) & (2^53 - 1), times 2^-53 (at::uniform_real_distribution<double>)
    const long long W = 2 * N;
    long long done = 0;
    while (done < W) {
        if (pos == MT_N) {
            mt_twist(cur, nxt);
            uint32_t* sw = cur; cur = nxt; nxt = sw;
            pos = 0;
        }
        const long long left = W - done;
        const int m = (int)(left < (long long)(MT_N - pos) ? left : (long long)(MT_N - pos));
        for (int i = t; i < m; i += 256) {
            const long long gw = done + i;
            const uint32_t w = mt_temper(cur[pos + i]);
            if ((gw & 1) == 0) {
                if (i + 1 < m) {
                    const uint32_t lo = mt_temper(cur[pos + i + 1]);
                    const unsigned long long r = (((unsigned long long)w << 32) | lo) & ((1ull << 53) - 1ull);
                    out[gw >> 1] = (double)r * 1.1102230246251565e-16;
                } else {
                    s_carry = w;  // its partner is the first word of the next block
                }
            } else if (i == 0) {
                const unsigned long long r = (((unsigned long long)s_carry << 32) | w) & ((1ull << 53) - 1ull);
                out[gw >> 1] = (double)r * 1.1102230246251565e-16;
            }
        }
        done += m;
        pos += m;
        __syncthreads();
    }
    for (int k = t; k < MT_N; k += 256) state[k] = cur[k];
    if (t == 0) state[MT_N] = (uint32_t)pos;
}

int launch_mt_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state) {
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(1), 0, ctx->stream, (uint32_t)(seed & 0xffffffffull), state);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_mt_rand64(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t N, double* out) {
    if (N == 0 && skip_words == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_mt_rand64, dim3(1), dim3(256), 0, ctx->stream, state, (long long)skip_words, (long long)N, out);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

}  // namespace midas
