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

// 2 N words -> N doubles: (hi << 32 | lo) & (2^53 - 1), times 2^-53 (at::uniform_real_distribution<double>); hi = the first word.
// hist (nullable): the first MT_HIST of the raw words are kept for the next call's jump (k_mt_jump).
constexpr int MT_DEG = 19937;
constexpr int MT_HIST = MT_DEG + MT_N - 1;  // x[k + i], k < 624, i < 19937
static_assert(MT_HIST == MIDAS_MT19937_HIST_WORDS, "include/midas_hip.h");
__global__ __launch_bounds__(256) void k_mt_emit(const uint32_t* __restrict__ raw, const int32_t* __restrict__ meta, long long N,
                                                 double* __restrict__ out, uint32_t* __restrict__ hist, long long off = 0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const uint32_t* w = raw + (meta ? meta[0] : 0) + off + 2 * i;  // off: the segment's first word (midas_mt19937_draws)
    const uint32_t w0 = w[0], w1 = w[1];
    const unsigned long long r = (((unsigned long long)mt_temper(w0) << 32) | mt_temper(w1)) & ((1ull << 53) - 1ull);
    out[i] = (double)r * 1.1102230246251565e-16;
    if (hist) {
        if (2 * i < MT_HIST) hist[2 * i] = w0;
        if (2 * i + 1 < MT_HIST) hist[2 * i + 1] = w1;
    }
}

// ---- chunked generation: G workgroups walk G pieces of a call's words side by side ------------------------------------------
// mt19937 is linear over GF(2): with phi its characteristic polynomial (degree 19937) and P_J = t^J mod phi,
//     x[k + J] = XOR over the exponents i of P_J of x[k + i]          (Haramoto et al. 2008; midastouch_amd/mt_jump.py)
// for every k - so the 624 words that START a piece of this call's output follow from MT_HIST consecutive words of the previous
// call's output (`hist`, kept by k_mt_emit) and one polynomial per piece (host-side set-up, 624-word bitsets).  A workgroup
// computes MT_JW consecutive words of one piece's start from the window hist[k0, k0 + 19936 + MT_JW) in LDS (78 KB, two
// workgroups a compute unit): ~10^4 terms per polynomial, MT_JW window words each; the partial sums meet in an XOR butterfly.
constexpr int MT_JW = 16;
constexpr int MT_JWIN = MT_DEG + MT_JW - 1;  // window words per workgroup: taps i <= 19936, words i .. i + 15
constexpr int MT_JQ = MT_N / 2;              // 312 groups of 64 exponents
static_assert(MT_N % MT_JW == 0 && MT_JQ % 4 == 0, "whole workgroups per piece, whole shares per wave");
// Lane L of a wave takes the exponents i = 64 q + L (bit L & 31 of polynomial word 2 q + (L >> 5)): the lanes that hold a term read
// window words i + m, m = 0 .. 15, one m at a time - bank (L + m) mod 64, all different: no LDS conflicts (a thread per term with
// its sixteen consecutive words collided four to five deep, and the kernel took 40 us where this takes 7 for eight pieces).
// The four waves share the q range; a lane's 78 polynomial words are in registers before the window has arrived.
__global__ __launch_bounds__(256) void k_mt_jump(const uint32_t* __restrict__ hist, const uint32_t* __restrict__ polys,
                                                 uint32_t* __restrict__ starts) {
    extern __shared__ uint32_t s_win[];  // MT_JWIN words (the first 4 x MT_JW reused for the cross-wave step)
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6, k0 = blockIdx.x * MT_JW, c = blockIdx.y;
    const uint32_t* P = polys + (size_t)c * MT_N + (lane >> 5);
    uint32_t pw[MT_JQ / 4];
#pragma unroll
    for (int r = 0; r < MT_JQ / 4; ++r) pw[r] = P[2 * (wave + 4 * r)];
    {   // the window: 16-byte pieces, every load of a thread in flight before its first LDS store (one round trip, not twenty)
        constexpr int NV = (MT_JWIN + 3) / 4, PER = (NV + 255) / 256;
        const uint4* src = reinterpret_cast<const uint4*>(hist + k0);  // k0 is a multiple of 16 words
        uint4 v[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = t + 256 * u;
            v[u] = j < NV - 1 ? src[j] : make_uint4(0u, 0u, 0u, 0u);
        }
        uint4* dst = reinterpret_cast<uint4*>(s_win);
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int j = t + 256 * u;
            if (j < NV - 1) dst[j] = v[u];
        }
        // the last, partial piece word by word (hist ends at MT_HIST: no read beyond it)
        if (t < MT_JWIN - 4 * (NV - 1)) s_win[4 * (NV - 1) + t] = hist[k0 + 4 * (NV - 1) + t];
    }
    __syncthreads();
    uint32_t acc[MT_JW];
#pragma unroll
    for (int m = 0; m < MT_JW; ++m) acc[m] = 0u;
    const int sh = lane & 31;
#pragma unroll
    for (int r = 0; r < MT_JQ / 4; ++r) {  // (fully unrolled: the polynomial words stay in registers)
        const int i = 64 * (wave + 4 * r) + lane;
        if (((pw[r] >> sh) & 1u) && i < MT_DEG) {  // (a polynomial has no term at or beyond t^19937: guards the window)
            const uint32_t* w = s_win + i;
#pragma unroll
            for (int m = 0; m < MT_JW; ++m) acc[m] ^= w[m];
        }
    }
#pragma unroll
    for (int m = 0; m < MT_JW; ++m) {
#pragma unroll
        for (int o = 32; o >= 1; o >>= 1) acc[m] ^= (uint32_t)__shfl_xor((int)acc[m], o);
    }
    __syncthreads();  // every wave is done with the window
    if (lane == 0) {
#pragma unroll
        for (int m = 0; m < MT_JW; ++m) s_win[wave * MT_JW + m] = acc[m];
    }
    __syncthreads();
    if (t < MT_JW) starts[(size_t)c * MT_N + k0 + t] = (s_win[t] ^ s_win[MT_JW + t]) ^ (s_win[2 * MT_JW + t] ^ s_win[3 * MT_JW + t]);
}

// Piece c = blocks [c bpc, (c + 1) bpc) of the call's nblocks blocks of 624 words (block 0 starts at the call's first word): its first
// block is starts[c], the others follow by the block recurrence of k_mt_blocks (same column walk, one barrier a block).  The piece
// that holds the call's last word leaves the generator's state: that block and the words consumed of it.
__global__ __launch_bounds__(MT_THREADS) void k_mt_chunks(const uint32_t* __restrict__ starts, long long nwords, int bpc, uint32_t* __restrict__ raw,
                                                          uint32_t* __restrict__ state) {
    __shared__ uint32_t s_mt[2][MT_N];
    const int c = threadIdx.x;
    const long long nblocks = (nwords + MT_N - 1) / MT_N;
    const long long fb = (long long)blockIdx.x * bpc;
    if (fb >= nblocks) return;
    const long long lb = (fb + bpc < nblocks ? fb + bpc : nblocks) - 1;  // last block of this piece
    for (int k = c; k < MT_N; k += MT_THREADS) {
        const uint32_t v = starts[(size_t)blockIdx.x * MT_N + k];
        s_mt[0][k] = v;
        raw[fb * MT_N + k] = v;
    }
    __syncthreads();
    int par = 0;
    uint32_t* r = raw + (fb + 1) * MT_N;
    for (int done = 0, k = (int)(lb - fb); done < k; ++done, par ^= 1, r += MT_N) {
        const uint32_t* __restrict__ o = s_mt[par];
        uint32_t* __restrict__ n = s_mt[par ^ 1];
        if (c < MT_COLS) {  // (k_mt_blocks' block step, see there)
            const bool third = c + 2 * MT_COLS < MT_N;
            const bool last = c + 2 * MT_COLS == MT_N - 1;
            const int i2 = third ? c + 2 * MT_COLS : 0, i3 = (third && !last) ? c + 2 * MT_COLS + 1 : 0;
            const uint32_t a0 = o[c], a1 = o[c + 1], far = o[c + MT_M];
            const uint32_t b0w = o[c + MT_COLS], b1w = o[c + MT_COLS + 1];
            const uint32_t c0w = o[i2], c1r = o[i3];
            uint32_t z0 = o[0], z1 = o[1], zm = o[MT_M];
            asm volatile("" : "+v"(z0), "+v"(z1), "+v"(zm));
            uint32_t nz = zm ^ mt_twist(z0, z1);
            asm volatile("" : "+v"(nz));
            const uint32_t c1w = last ? nz : c1r;
            const uint32_t n0 = far ^ mt_twist(a0, a1);
            const uint32_t n1 = n0 ^ mt_twist(b0w, b1w);
            const uint32_t n2 = n1 ^ mt_twist(c0w, c1w);
            n[c] = n0;
            n[c + MT_COLS] = n1;
            if (third) n[c + 2 * MT_COLS] = n2;
            r[c] = n0;
            r[c + MT_COLS] = n1;
            if (third) r[c + 2 * MT_COLS] = n2;
        }
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    }
    if (lb == nblocks - 1) {
        const uint32_t* fin = s_mt[par];
        for (int k = c; k < MT_N; k += MT_THREADS) state[k] = fin[k];
        if (c == 0) { state[MT_N] = (uint32_t)(nwords - lb * MT_N); state[MT_N + 1] = 0; }
    }
}

int launch_mt_seed(midas_ctx* ctx, uint64_t seed, uint32_t* state) {
    hipLaunchKernelGGL(k_mt_seed, dim3(1), dim3(1), 0, ctx->stream, (uint32_t)(seed & 0xffffffffull), state);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// torch.normal(mean, std, size) of float32 values from the stream's words (ATen normal_fill, modules/particle_filter.py:326-335):
// sixteen at a time, u1 = 1 - data[j], u2 = data[j + 8], data[j] = (radius(u1) cos(theta(u2))) std + mean, data[j + 8] = (radius
// sin) std + mean; when numel is not a multiple of 16 the last sixteen values are drawn AGAIN from sixteen further words.  radius /
// cos / sin are TABLES over the 2^24 values a float32 uniform takes, read off torch.normal itself on the host
// (midastouch_amd/torch_normal.py): bit-identical to whatever math library ATen dispatches to.
constexpr uint32_t MT_U24 = (1u << 24) - 1u;
__global__ __launch_bounds__(256) void k_mt_normal(const uint32_t* __restrict__ raw, const int32_t* __restrict__ meta, long long numel, long long nwords,
                                                   const float* __restrict__ R, const float* __restrict__ C, const float* __restrict__ S,
                                                   float mean, float std, float* __restrict__ out, uint32_t* __restrict__ hist, long long off = 0) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const uint32_t* w = raw + (meta ? meta[0] : 0) + off;
    if (hist && i < MT_HIST && i < nwords) hist[i] = w[i];
    if (i >= numel) return;
    const bool tail = (numel & 15) && i >= numel - 16;        // redrawn from the sixteen words behind the first numel
    const long long base = tail ? numel : (i & ~15ll);
    const int j = (int)(tail ? i - (numel - 16) : (i & 15));
    const uint32_t k1 = mt_temper(w[base + (j & 7)]) & MT_U24, k2 = mt_temper(w[base + 8 + (j & 7)]) & MT_U24;
    const float n = R[k1] * (j < 8 ? C[k2] : S[k2]);
    out[i] = __builtin_fmaf(n, std, mean);
}

// The words of a call: sequential walk (k_mt_blocks; raw + meta offset) or in pieces (k_mt_jump + k_mt_chunks; offset 0, meta null).
// `polys` = G polynomials of 624 words (t^J_c mod phi, J_c = distance from the first word in `hist` to the first word of piece c - the
// host knows both); skip_words is part of J_c already: in pieces the state is replaced, not advanced.
static int mt_words(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t nwords, const uint32_t* hist, const uint32_t* polys, int32_t G,
                    uint32_t** raw_out, int32_t** meta_out) {
    void* p;
    int rc;
    *raw_out = nullptr;
    *meta_out = nullptr;
    if (polys && G > 0) {
        const int64_t nblocks = ceil_div(nwords, MT_N);
        const int bpc = (int)ceil_div(nblocks, G);
        if ((rc = midas_scratch(ctx, ((size_t)nblocks + 1) * MT_N * sizeof(uint32_t), &p))) return rc;
        uint32_t* raw = (uint32_t*)p;
        if ((rc = midas_scratch(ctx, (size_t)G * MT_N * sizeof(uint32_t), &p))) return rc;
        uint32_t* starts = (uint32_t*)p;
        static bool attr_set[64] = {};
        const int di = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
        const int lds = MT_JWIN * (int)sizeof(uint32_t);
        if (!attr_set[di] || ctx->device != di) {
            MIDAS_HIP_CHECK(ctx, hipFuncSetAttribute((const void*)k_mt_jump, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
            attr_set[di] = true;
        }
        hipLaunchKernelGGL(k_mt_jump, dim3(MT_N / MT_JW, (unsigned)G), dim3(256), lds, ctx->stream, hist, polys, starts);
        hipLaunchKernelGGL(k_mt_chunks, dim3((unsigned)G), dim3(MT_THREADS), 0, ctx->stream, (const uint32_t*)starts, (long long)nwords, bpc, raw, state);
        *raw_out = raw;
        return MIDAS_OK;
    }
    if (nwords > 0) {
        // blocks b0 .. bl: at most the words, the rest of the block they start in and of the one they end in
        if ((rc = midas_scratch(ctx, ((size_t)nwords + 3 * MT_N) * sizeof(uint32_t), &p))) return rc;
        *raw_out = (uint32_t*)p;
        if ((rc = midas_scratch(ctx, 64, &p))) return rc;
        *meta_out = (int32_t*)p;
    }
    hipLaunchKernelGGL(k_mt_blocks, dim3(1), dim3(MT_THREADS), 0, ctx->stream, state, (long long)skip_words, (long long)nwords, *raw_out, *meta_out);
    return MIDAS_OK;
}

int launch_mt_rand64_chunked(midas_ctx* ctx, uint32_t* state, int64_t N, double* out, uint32_t* hist, const uint32_t* polys, int32_t G) {
    uint32_t* raw;
    int32_t* meta;
    int rc = mt_words(ctx, state, 0, 2 * N, hist, polys, G, &raw, &meta);
    if (rc) return rc;
    hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw, (const int32_t*)meta,
                       (long long)N, out, hist);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

int launch_mt_rand64(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t N, double* out, uint32_t* hist) {
    if (N == 0 && skip_words == 0) return MIDAS_OK;
    uint32_t* raw;
    int32_t* meta;
    int rc = mt_words(ctx, state, skip_words, 2 * N, nullptr, nullptr, 0, &raw, &meta);
    if (rc) return rc;
    if (N > 0)
        hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw, (const int32_t*)meta, (long long)N,
                           out, 2 * N >= MT_HIST ? hist : (uint32_t*)nullptr);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// numel float32 normals (numel >= 16); polys / G as in mt_words (nullptr / 0: the sequential walk)
int launch_mt_normal32(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int64_t numel, float mean, float std, const float* R, const float* C,
                       const float* S, float* out, uint32_t* hist, const uint32_t* polys, int32_t G) {
    const int64_t nwords = numel + ((numel & 15) ? 16 : 0);
    uint32_t* raw;
    int32_t* meta;
    int rc = mt_words(ctx, state, skip_words, nwords, hist, polys, G, &raw, &meta);
    if (rc) return rc;
    const int64_t threads = numel > MT_HIST ? numel : (nwords < MT_HIST ? nwords : MT_HIST);  // (the history's words have a thread each)
    hipLaunchKernelGGL(k_mt_normal, dim3((unsigned)ceil_div(threads, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw, (const int32_t*)meta,
                       (long long)numel, (long long)nwords, R, C, S, mean, std, out, nwords >= MT_HIST ? hist : (uint32_t*)nullptr);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

// ---- several draws of one frame from ONE walk of the generator -----------------------------------------------------------------
// A seeded frame of the reference takes torch.normal (N, 3) twice and torch.rand N float64 (particle_filter.py:326-335, :245): as
// three calls that is three jumps, three walks, three launches' worth of Python.  Here the segments' words are generated together
// (one jump + one set of pieces over the sum of their words), each segment is then transformed from its place in the raw words; the
// history the next call's jump reads is copied once.
__global__ __launch_bounds__(256) void k_mt_hist(const uint32_t* __restrict__ raw, const int32_t* __restrict__ meta, uint32_t* __restrict__ hist) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < MT_HIST) hist[i] = raw[(meta ? meta[0] : 0) + i];
}

int launch_mt_draws(midas_ctx* ctx, uint32_t* state, int64_t skip_words, int32_t nseg, const midas_mt_segment* segs, const float* R,
                    const float* C, const float* S, uint32_t* hist, const uint32_t* polys, int32_t G) {
    int64_t total = 0;
    for (int i = 0; i < nseg; ++i)
        total += segs[i].kind == MIDAS_MT_SEGMENT_RAND64 ? 2 * segs[i].count : segs[i].count + ((segs[i].count & 15) ? 16 : 0);
    uint32_t* raw;
    int32_t* meta;
    int rc = mt_words(ctx, state, skip_words, total, hist, polys, G, &raw, &meta);
    if (rc) return rc;
    int64_t off = 0;
    for (int i = 0; i < nseg; ++i) {
        const int64_t n = segs[i].count;
        if (segs[i].kind == MIDAS_MT_SEGMENT_RAND64) {
            if (n > 0)
                hipLaunchKernelGGL(k_mt_emit, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw, (const int32_t*)meta,
                                   (long long)n, (double*)segs[i].out_dev, (uint32_t*)nullptr, (long long)off);
            off += 2 * n;
        } else {
            const int64_t nw = n + ((n & 15) ? 16 : 0);
            hipLaunchKernelGGL(k_mt_normal, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw, (const int32_t*)meta,
                               (long long)n, (long long)nw, R, C, S, segs[i].mean, segs[i].std, (float*)segs[i].out_dev, (uint32_t*)nullptr,
                               (long long)off);
            off += nw;
        }
    }
    if (hist && total >= MT_HIST)
        hipLaunchKernelGGL(k_mt_hist, dim3((unsigned)ceil_div((int64_t)MT_HIST, 256)), dim3(256), 0, ctx->stream, (const uint32_t*)raw,
                           (const int32_t*)meta, hist);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

MIDAS_WARM_TU(mt19937, k_mt_seed)

}  // namespace midas
