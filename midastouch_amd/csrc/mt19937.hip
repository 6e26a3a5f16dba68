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
// is sequential from block to block (624 words); inside a block word k >= 227 needs the new word k - 227, so a thread that
// owns the words j, j + 227, j + 454 carries that dependence in its own registers and a block costs ONE barrier (see
// k_mt_rand64); a block's 312 doubles are written out beside the compute of the next block.
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

// One workgroup of MT_THREADS threads, ONE barrier per 624-word block.  Thread j < 227 owns the words j, j + 227 and j + 454 of
// the next block: word k needs the previous block's words k and k + 1 (LDS, stable) and - for k >= 227 - the NEW word
// k - 227, which is the thread's own previous result: a private chain of three mixes, no hand-over between threads.  The
// one exception is word 623 = mix(prev[623], new[0], new[396]) (new[0] belongs to thread 0): it stays pending through the
// block's barrier and is finalised at the start of the next block's compute by the two threads that read it (k = 622's
// right neighbour, and the writer of the LDS copy); the output pass, which runs beside that compute, recomputes it the same
// way instead of reading the slot that is being written.  Two buffers alternate; slot 623 of the buffer being overwritten is
// never written by the compute (it is the pending one), so it still holds the word 623 the finalisation needs.
constexpr int MT_THREADS = 256, MT_P = MT_N - MT_M;  // 227 words per dependent step
struct MtEmit { int pos, end; long long gw; };      // words [pos, end) of the block; gw = stream index of word `pos`

__global__ __launch_bounds__(MT_THREADS) void k_mt_rand64(uint32_t* __restrict__ state, long long skip, long long N, double* __restrict__ out) {
    __shared__ uint32_t s_mt[2][MT_N];
    __shared__ uint32_t s_carry;
    const int t = threadIdx.x;
    for (int k = t; k < MT_N; k += MT_THREADS) s_mt[0][k] = state[k];
    int pos = (int)state[MT_N];
    int b = 0;              // s_mt[b]: the current block
    bool pending = false;   // s_mt[b][623] not finalised yet (its ingredients: s_mt[b ^ 1][623], s_mt[b][0], s_mt[b][396])
    __syncthreads();
    // word 623 of the current block, from LDS or - while pending - from its ingredients
    auto word623 = [&]() { return pending ? mt_mix(s_mt[b ^ 1][MT_N - 1], s_mt[b][0], s_mt[b][MT_M - 1]) : s_mt[b][MT_N - 1]; };
    auto cur_word = [&](int k) { return k == MT_N - 1 ? word623() : s_mt[b][k]; };
    // the next block into s_mt[b ^ 1] (its word 623 stays pending); ends WITHOUT a barrier
    auto compute = [&]() {
        const uint32_t* old = s_mt[b];
        uint32_t* nw = s_mt[b ^ 1];
        if (t < MT_P) {
            const int k1 = t + MT_P, k2 = t + 2 * MT_P;
            const uint32_t o0 = old[t], o1 = old[t + 1], of = t + MT_M == MT_N - 1 ? word623() : old[t + MT_M];  // (t = 226 reads word 623)
            const uint32_t p0 = old[k1], p1 = old[k1 + 1];
            const bool third = k2 < MT_N - 1;  // k2 == 623 (t == 169) is the pending word
            const uint32_t q0 = third ? old[k2] : 0u;
            const uint32_t q1 = third ? (k2 + 1 == MT_N - 1 ? word623() : old[k2 + 1]) : 0u;
            const uint32_t n0 = mt_mix(o0, o1, of);
            const uint32_t n1 = mt_mix(p0, p1, n0);
            nw[t] = n0;
            nw[k1] = n1;
            if (third) nw[k2] = mt_mix(q0, q1, n1);
        }
        if (pending && t == MT_THREADS - 1) s_mt[b][MT_N - 1] = word623();  // the LDS copy of the current block's last word
    };
    auto advance = [&]() {  // after compute + barrier: the new block is current; its word 623 is pending
        b ^= 1;
        pending = true;
        pos = 0;
    };
    // 2 N words -> N doubles: (hi << 32 | lo) & (2^53 - 1), times 2^-53 (at::uniform_real_distribution<double>); hi = even stream index.
    // Reads the CURRENT block only (beside the compute of the next one, which writes the other buffer).
    auto emit = [&](const MtEmit& e) {
        const int odd = (int)(e.gw & 1);
        if (odd && t == MT_THREADS - 1) {  // the block starts with the low word of a value whose high word ended the block before
            const unsigned long long r = (((unsigned long long)s_carry << 32) | mt_temper(cur_word(e.pos))) & ((1ull << 53) - 1ull);
            out[e.gw >> 1] = (double)r * 1.1102230246251565e-16;
        }
        for (int o = e.pos + odd + 2 * t; o < e.end; o += 2 * MT_THREADS) {
            const uint32_t hi = mt_temper(cur_word(o));
            if (o + 1 < e.end) {
                const unsigned long long r = (((unsigned long long)hi << 32) | mt_temper(cur_word(o + 1))) & ((1ull << 53) - 1ull);
                out[(e.gw + (o - e.pos)) >> 1] = (double)r * 1.1102230246251565e-16;
            } else {
                s_carry = hi;  // its partner is the first word of the next block
            }
        }
    };
    // skip: whole blocks are stepped over, the rest is an offset
    while (skip > 0) {
        const int avail = MT_N - pos;
        if (skip >= avail) {
            skip -= avail;
            compute();
            __syncthreads();
            advance();
        } else {
            pos += (int)skip;
            skip = 0;
        }
    }
    const long long W = 2 * N;
    long long gw = 0;
    bool pend_emit = false;
    MtEmit pe{0, 0, 0};
    while (true) {
        if (pos == MT_N && gw < W) {
            compute();
            if (pend_emit) { emit(pe); pend_emit = false; }
            __syncthreads();
            advance();
        }
        if (gw >= W) break;
        const long long left = W - gw;
        const int m = (int)(left < (long long)(MT_N - pos) ? left : (long long)(MT_N - pos));
        pe.pos = pos; pe.end = pos + m; pe.gw = gw;
        pend_emit = true;
        gw += m;
        pos += m;
    }
    if (pend_emit) emit(pe);
    __syncthreads();
    if (pending && t == 0) s_mt[b][MT_N - 1] = word623();  // the state goes back complete
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

}  // namespace midas
