// resample.hip - weights (K5), float64 CDF in the fixed blocked order (K6), inverse-CDF search (K7),
// gather-resample (K8) and the fused tail of the per-frame step.
//
// Summation order (DESIGN.md "Summation order", oracle/midas_oracle.c mo_blocked_scan): 16 values =
// chunk (one lane), 16 chunks = group (a quarter-wave), 16 groups = block (one 256-thread workgroup,
// 4096 values); every level is a sequential float64 sum in index order starting from +0.0.
#include "midas_internal.hpp"
#include "midas_math.hpp"
#include "peer_row.hpp"
#include "resample_search.hpp"
#include "tail_block.hpp"
#include "tail_group.hpp"

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "the peer-mapped route kernel's completion protocol relies on gfx942 / gfx950 write-through store acknowledgement (see k_shard_route_*)"
#endif

namespace midas {

#ifdef MIDAS_DEBUG_CLOCKS
__device__ long long g_tg_clk[64];
__device__ long long g_tg_w[8192];
#endif
constexpr double ISCLOSE_ATOL = 1e-8;  // torch.isclose default atol (particle_filter.py:460-463)

MD double wsum_shuffles(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
MD double wsum(double v) { return wave_sum_ordered(v); }  // (the same additions in the same order: midas_math.hpp)
// self-check of the register-move sums against the shuffle butterflies: 64 doubles in, per lane {shuffle wave sum, ordered wave
// sum, shuffle quarter sum, ordered quarter sum} out (256 doubles)
__global__ __launch_bounds__(64) void k_debug_wave_sum(const double* __restrict__ in, double* __restrict__ out) {
    const int l = threadIdx.x;
    const double v = in[l];
    double q = v;
    q += __shfl_xor(q, 8); q += __shfl_xor(q, 4); q += __shfl_xor(q, 2); q += __shfl_xor(q, 1);
    out[4 * l] = wsum_shuffles(v);
    out[4 * l + 1] = wave_sum_ordered(v);
    out[4 * l + 2] = q;
    out[4 * l + 3] = quarter_sum_ordered(v);
}
int launch_selftest_wave_sums(midas_ctx* ctx, const double* in64, double* out256) {
    hipLaunchKernelGGL(k_debug_wave_sum, dim3(1), dim3(64), 0, ctx->stream, in64, out256);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}
MD double wmax(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = t > v ? t : v; }
    return v;
}
MD double wmin(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { double t = __shfl_xor(v, o); v = t < v ? t : v; }
    return v;
}

// Block-cooperative version of seq_totals: the partials are fetched in parallel into LDS (one round trip
// instead of nb dependent ones), then every thread adds them in order from LDS.  s_buf holds >= nb doubles.
MD void seq_totals_lds(const double* __restrict__ part, int nb, int b, double* s_buf, double& bp, double& total) {
    for (int i = threadIdx.x; i < nb; i += blockDim.x) s_buf[i] = part[i];
    __syncthreads();
    double acc = 0.0, pre = 0.0;
    for (int i = 0; i < nb; ++i) {
        if (i == b) pre = acc;
        acc = acc + s_buf[i];
    }
    bp = pre;
    total = acc;
    __syncthreads();
}

// sequential sum of per-block totals part[0..nb), exclusive prefix up to `b` returned in bp
MD void seq_totals(const double* __restrict__ part, int nb, int b, double& bp, double& total) {
    double acc = 0.0, pre = 0.0;
    for (int i = 0; i < nb; ++i) {
        if (i == b) pre = acc;
        acc = acc + part[i];
    }
    bp = pre;
    total = acc;
}

// block-wide max/min over an array of partials
MD void block_extrema(const double* __restrict__ pmax, const double* __restrict__ pmin, int np, int stride,
                      double* s_red, double& mx, double& mn) {
    double a = -INFINITY, b = INFINITY;
    bool nan = false;
    for (int i = threadIdx.x; i < np; i += blockDim.x) {
        double u = pmax[(int64_t)i * stride], v = pmin[(int64_t)i * stride];
        nan |= (u != u) || (v != v);
        a = u > a ? u : a;
        b = v < b ? v : b;
    }
    a = wmax(a);
    b = wmin(b);
    const bool wnan = __any(nan);
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[w] = a; s_red[8 + w] = b; s_red[16 + w] = wnan ? 1.0 : 0.0; }
    __syncthreads();
    a = s_red[0]; b = s_red[8];
    double f = s_red[16];
    for (int i = 1; i < nw; ++i) {
        a = s_red[i] > a ? s_red[i] : a;
        b = s_red[8 + i] < b ? s_red[8 + i] : b;
        f += s_red[16 + i];
    }
    __syncthreads();
    if (f != 0.0) { a = NAN; b = NAN; }  // torch.max/min propagate NaN
    mx = a;
    mn = b;
}

// ------------------------------------------------------------------------------------------------
// standalone kernels
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_gather_f64(int64_t N, const double* __restrict__ table,
                                                    const int32_t* __restrict__ idx, double* __restrict__ out) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n < N) out[n] = table[idx[n]];
}

// per-block extrema of x (4096 values per block)
__global__ __launch_bounds__(256) void k_extrema(int64_t N, const double* __restrict__ x, double* __restrict__ pmax,
                                                 double* __restrict__ pmin) {
    __shared__ double s_red[24];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
    double a = -INFINITY, b = INFINITY;
    bool nan = false;
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        if (i < N) {
            double v = x[i];
            nan |= v != v;
            a = v > a ? v : a;
            b = v < b ? v : b;
        }
    }
    a = wmax(a);
    b = wmin(b);
    const bool wnan = __any(nan);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { s_red[w] = a; s_red[8 + w] = b; s_red[16 + w] = wnan ? 1.0 : 0.0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double f = 0.0;
        for (int i = 0; i < 4; ++i) {
            a = s_red[i] > a ? s_red[i] : a;
            b = s_red[8 + i] < b ? s_red[8 + i] : b;
            f += s_red[16 + i];
        }
        pmax[blockIdx.x] = f != 0.0 ? NAN : a;
        pmin[blockIdx.x] = f != 0.0 ? NAN : b;
    }
}

// e = exp(x - max) (or x when the softmax is skipped) ; per-block totals of e in the spec order.
// valid (nullable) is NOT applied here.  flag_out[0] = 1 when the softmax is applied.
__global__ __launch_bounds__(256) void k_exp_partial(int64_t N, const double* __restrict__ x, int np,
                                                     const double* __restrict__ pmax, const double* __restrict__ pmin,
                                                     int32_t softmax, double* __restrict__ e_out,
                                                     double* __restrict__ part_sum, int32_t* __restrict__ flag_out) {
    __shared__ double s_red[24];
    __shared__ double s_gtot[16];
    double mx, mn;
    block_extrema(pmax, pmin, np, 1, s_red, mx, mn);
    // not isclose(max - min, 0): |max-min| > atol, or NaN (isclose(NaN, 0) is False)
    const double spread = mx - mn;
    const bool apply = softmax && !(__builtin_fabs(spread) <= ISCLOSE_ATOL);
    if (blockIdx.x == 0 && threadIdx.x == 0 && flag_out) flag_out[0] = apply ? 1 : 0;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_CHUNK;
    double v[SCAN_CHUNK], l[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        double e = 0.0;
        if (i < N) {
            const double xi = x[i];
            e = apply ? exp_spec(xi - mx) : xi;
            e_out[i] = e;
        }
        v[j] = e;
    }
    const double W = block_scan(v, l, s_gtot);
    if (threadIdx.x == 0) part_sum[blockIdx.x] = W;
}

// w = e / S with S = sequential sum of the block totals (only when flag[0])
__global__ __launch_bounds__(256) void k_normalise(int64_t N, double* __restrict__ w, int nb,
                                                   const double* __restrict__ part_sum,
                                                   const int32_t* __restrict__ flag) {
    if (!flag[0]) return;
    double bp, S;
    seq_totals(part_sum, nb, 0, bp, S);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) w[i] = w[i] / S;
}

__global__ __launch_bounds__(256) void k_prune(int64_t N, double* __restrict__ w, const double* __restrict__ dist,
                                               double thr, int32_t* __restrict__ nvalid) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    bool keep = false;
    if (i < N) {
        keep = !(dist[i] > thr);
        w[i] = w[i] * (keep ? 1.0 : 0.0);
    }
    unsigned long long m = __ballot(keep);
    if ((threadIdx.x & 63) == 0 && m) atomicAdd(nvalid, (int32_t)__popcll(m));
}

// block-local prefix of w (spec order) -> lp ; block totals -> part ; NaN detection -> status[0] |= 2
__global__ __launch_bounds__(256) void k_scan_local(int64_t N, const double* __restrict__ w, double* __restrict__ lp,
                                                    double* __restrict__ part, int32_t* __restrict__ status) {
    __shared__ double s_gtot[16];
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_CHUNK;
    double v[SCAN_CHUNK], l[SCAN_CHUNK];
    bool nan = false;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        double t = 0.0;
        if (i < N) { t = w[i]; nan |= t != t; }
        v[j] = t;
    }
    const double W = block_scan(v, l, s_gtot);
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j)
        if (base + j < N) lp[base + j] = l[j];
    if (threadIdx.x == 0) part[blockIdx.x] = W;
    const bool wnan = __any(nan);
    if (wnan && (threadIdx.x & 63) == 0) atomicOr(status, 2);
}

// cdf_i = (BP_b + lp_i) / total ; cdf_{N-1} = 1 ; status[0] = 1 when total == 0 (2 already set on NaN)
__global__ __launch_bounds__(256) void k_cdf_final(int64_t N, double* __restrict__ cdf, int nb,
                                                   const double* __restrict__ part, int32_t* __restrict__ status) {
    double bp, total;
    seq_totals(part, nb, blockIdx.x, bp, total);
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (total != total) atomicOr(status, 2);
        else if (total == 0.0) atomicOr(status, 1);
    }
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + (int64_t)j * 256 + threadIdx.x;
        if (i < N) cdf[i] = (i == N - 1) ? 1.0 : (bp + cdf[i]) / total;
    }
}

MD int32_t search_lower(const double* __restrict__ cdf, int64_t N, double u) {
    int64_t lo = 0, hi = N;
    while (hi > lo) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] < u) lo = mid + 1; else hi = mid;
    }
    return (int32_t)(lo < N ? lo : N - 1);
}
MD int32_t search_upper(const double* __restrict__ cdf, int64_t N, double u) {
    int64_t lo = 0, hi = N;
    while (hi > lo) {
        const int64_t mid = lo + ((hi - lo) >> 1);
        if (cdf[mid] <= u) lo = mid + 1; else hi = mid;
    }
    return (int32_t)(lo < N ? lo : N - 1);
}

MD int32_t resample_slot(const double* __restrict__ cdf, int64_t N, int64_t M, int64_t i, int32_t mode,
                         const double* __restrict__ u, float u32, uint64_t seed, uint64_t step) {
    if (mode == MIDAS_RESAMPLE_MULTINOMIAL) {
        const double ui = u ? u[i] : philox_uniform53((uint64_t)i, seed, step);
        return search_lower(cdf, N, ui);
    }
    const float r = u32 >= 0.0f ? u32 : philox_uniform24(seed, step);
    const float off = r / (float)M;
    double loc = (double)i / (double)M + (double)off;
    loc = loc >= 1.0 ? loc - 1.0 : loc;  // fmod(loc, 1) for loc in [0, 2)
    return search_upper(cdf, N, loc);
}

__global__ __launch_bounds__(256) void k_search(int64_t N, const double* __restrict__ cdf, int64_t M, int32_t mode,
                                                const double* __restrict__ u, float u32, uint64_t seed, uint64_t step,
                                                int32_t* __restrict__ idx) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < M) idx[i] = resample_slot(cdf, N, M, i, mode, u, u32, seed, step);
}

// rows of 16-byte multiples: one lane per 16-byte piece ; other sizes: one lane per byte group of 4/8/1
template <typename V>
__global__ __launch_bounds__(256) void k_gather_rows(int64_t M, const int32_t* __restrict__ idx,
                                                     const V* __restrict__ src, V* __restrict__ dst, int per_row) {
    const int64_t t = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t i = t / per_row;
    const int k = (int)(t - i * per_row);
    if (i < M) dst[i * per_row + k] = src[(int64_t)idx[i] * per_row + k];
}

// ------------------------------------------------------------------------------------------------
// step tail (shared by the fused single-GPU step and the particle-sharded multi-GPU step)
// ------------------------------------------------------------------------------------------------
// A shard owns N local particles = the global slots [slot_base, slot_base + N) and the global blocks
// [block_base, block_base + nb_local) of the summation spec; arrays suffixed _all span every shard.
// Single GPU: slot_base = block_base = 0 and the _all arrays are the local ones.

// TA: e = exp(x - 1) comes from the particle update (x replaces it when the isclose guard skips the softmax);
//     em = e * valid ;
//     block-local prefix of em -> lp_out ; local block totals of e (softmax denominator) and of em
//     (CDF total) ; status[0] = 2 on NaN ; status[1] = particles kept.
//     The CDF is built from e*valid directly: the softmax normalisation cancels in prefix / total.
__global__ __launch_bounds__(256) void k_tail_a(int64_t N, const double* __restrict__ x, const uint8_t* __restrict__ valid,
                                                int np, int pstride, const double* __restrict__ pmax_all,
                                                const double* __restrict__ pmin_all, int32_t softmax,
                                                double* __restrict__ e_io, double* __restrict__ lp_out,
                                                double* __restrict__ block_sums_e, double* __restrict__ block_totals_em,
                                                double* __restrict__ flags_out, int32_t* __restrict__ flag,
                                                int32_t* __restrict__ status) {
    __shared__ double s_red[24];
    __shared__ double s_gtot[16];
    if (blockIdx.y) {  // batch of trajectories
        const int64_t b = blockIdx.y, o = b * N;
        x += o; valid += o; e_io += o; lp_out += o;
        pmax_all += b * np; pmin_all += b * np;
        block_sums_e += b * gridDim.x; block_totals_em += b * gridDim.x;
        flag += b; status += 2 * b;
    }
    double mx, mn;
    block_extrema(pmax_all, pmin_all, np, pstride, s_red, mx, mn);
    const bool apply = softmax && !(__builtin_fabs(mx - mn) <= ISCLOSE_ATOL);
    if (blockIdx.x == 0 && threadIdx.x == 0) flag[0] = apply ? 1 : 0;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK + (int64_t)threadIdx.x * SCAN_CHUNK;
    double v[SCAN_CHUNK], vm[SCAN_CHUNK], l[SCAN_CHUNK];
    bool nan = false;
    int kept = 0;
    // unconditional loads on clamped slots (conditional ones are issued one round trip at a time)
    const double* __restrict__ src = apply ? e_io : x;
    uint8_t okv[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j, ic = i < N ? i : N - 1;
        v[j] = src[ic];
        okv[j] = valid[ic];
    }
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + j;
        double e = 0.0, em = 0.0;
        if (i < N) {
            e = v[j];
            if (!apply) e_io[i] = e;
            const bool ok = okv[j] != 0;
            em = e * (ok ? 1.0 : 0.0);
            kept += ok ? 1 : 0;
            nan |= em != em;
        }
        v[j] = e;
        vm[j] = em;
    }
    const double We = block_scan(v, l, s_gtot);
    __syncthreads();
    const double Wm = block_scan(vm, l, s_gtot);
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j)
        if (base + j < N) lp_out[base + j] = l[j];
    if (threadIdx.x == 0) { block_sums_e[blockIdx.x] = We; block_totals_em[blockIdx.x] = Wm; }
    const bool wnan = __any(nan);
    if (wnan && (threadIdx.x & 63) == 0) {
        atomicOr(&status[0], 2);
        if (flags_out) atomicAdd(&flags_out[0], 1.0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if ((threadIdx.x & 63) == 0 && kept) {
        atomicAdd(&status[1], kept);
        if (flags_out) atomicAdd(&flags_out[1], (double)kept);  // exact: integers far below 2^53
    }
}

// TA2 (fused single-GPU step, deferred mode): the particle update ran concurrently with the codebook scoring,
// so this kernel gathers x = scores[nn_idx] itself, takes e = exp(x - 1) and produces what TA produces - with
// the isclose guard (a GLOBAL property of x) deferred to TB:  the guard can only fire when every block's own
// range is within the tolerance, so a block whose range is wider writes the softmax variant only; a block
// whose range is within it (rare: all its particles share one score) writes the raw variant as well, and TB
// picks one after reducing the per-block extrema.  Global traffic is coalesced (slot = k * 256 + thread); the
// chunk-per-thread view the summation spec needs goes through LDS (one pad double per 16-slot chunk).
MD int pad16(int i) { return i + (i >> 4); }

MD void scan_variant(const double* val, const uint8_t* okm, int64_t base, int64_t N, double* s_a, double* s_m,
                     double* s_gtot, double* __restrict__ lp_out, double* __restrict__ gend_out,
                     double* __restrict__ ggend_out, double& W_all, double& W_masked, bool& nan) {
    const int t = threadIdx.x;
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int s = k * 256 + t;
        const bool in = base + s < N;
        const double v = in ? val[k] : 0.0;
        const double m = in ? v * (okm[k] ? 1.0 : 0.0) : 0.0;
        nan |= m != m;
        s_a[pad16(s)] = v;
        s_m[pad16(s)] = m;
    }
    __syncthreads();
    double v[SCAN_CHUNK], vm[SCAN_CHUNK], l[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) { v[j] = s_a[t * 17 + j]; vm[j] = s_m[t * 17 + j]; }
    W_all = block_scan(v, l, s_gtot);
    __syncthreads();
    W_masked = block_scan(vm, l, s_gtot);
    if (base + (int64_t)t * SCAN_CHUNK < N) gend_out[(base >> 4) + t] = l[SCAN_CHUNK - 1];  // block-local prefix at the chunk end
    if ((t & 15) == 15) ggend_out[(base >> 8) + (t >> 4)] = l[SCAN_CHUNK - 1];               // ... at the end of each 256-slot group
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) s_m[t * 17 + j] = l[j];  // own chunk only
    __syncthreads();
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int s = k * 256 + t;
        if (base + s < N) lp_out[base + s] = s_m[pad16(s)];
    }
    __syncthreads();
}

__global__ __launch_bounds__(256) void k_tail_a2(int64_t N, const double* __restrict__ scores,
                                                 const int32_t* __restrict__ nn_idx, const uint8_t* __restrict__ valid,
                                                 int32_t softmax, double* __restrict__ e_out, double* __restrict__ x_out,
                                                 double* __restrict__ lp_soft, double* __restrict__ lp_raw,
                                                 double* __restrict__ gend_soft, double* __restrict__ gend_raw,
                                                 double* __restrict__ ggend_soft, double* __restrict__ ggend_raw,
                                                 double* __restrict__ bsum_e, double* __restrict__ btot_soft,
                                                 double* __restrict__ btot_raw, double* __restrict__ bmax,
                                                 double* __restrict__ bmin, int32_t* __restrict__ status,
                                                 double* __restrict__ flags_out, int64_t score_stride) {
    if (blockIdx.y) {  // batch of trajectories: every per-trajectory array is (B, ...) contiguous, plain strides
        const int64_t b = blockIdx.y, o = b * N, ng = (N + SCAN_CHUNK - 1) / SCAN_CHUNK, nb = gridDim.x;
        scores += b * score_stride; nn_idx += o; valid += o; e_out += o; x_out += o; lp_soft += o; lp_raw += o;
        gend_soft += b * ng; gend_raw += b * ng; ggend_soft += b * 16 * nb; ggend_raw += b * 16 * nb;
        bsum_e += b * nb; btot_soft += b * nb; btot_raw += b * nb; bmax += b * nb; bmin += b * nb;
        status += 2 * b;
    }
    __shared__ double s_a[SCAN_BLOCK + SCAN_BLOCK / 16];
    __shared__ double s_m[SCAN_BLOCK + SCAN_BLOCK / 16];
    __shared__ double s_gtot[16];
    __shared__ double s_red[24];
    const int t = threadIdx.x;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
    // loads are unconditional on clamped slots (a conditional load per slot makes the compiler wait for each
    // one before issuing the next); out-of-range slots are neutralised afterwards
    int32_t nn[SCAN_CHUNK];
    uint8_t okm[SCAN_CHUNK];
    double x[SCAN_CHUNK];
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const int64_t i = base + k * 256 + t, ic = i < N ? i : N - 1;
        nn[k] = nn_idx[ic];
        okm[k] = valid[ic];
    }
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) x[k] = scores[nn[k]];
    double mx = -INFINITY, mn = INFINITY;
    bool xnan = false;
    int kept = 0;
#pragma unroll
    for (int k = 0; k < SCAN_CHUNK; ++k) {
        const bool in = base + k * 256 + t < N;
        xnan |= in && x[k] != x[k];
        mx = in && x[k] > mx ? x[k] : mx;
        mn = in && x[k] < mn ? x[k] : mn;
        kept += in && okm[k] ? 1 : 0;
        if (!in) { x[k] = 0.0; okm[k] = 0; }
    }
    // block extrema (NaN propagates, as torch.max / torch.min do)
    mx = wmax(mx);
    mn = wmin(mn);
    const bool wxnan = __any(xnan);
    if ((t & 63) == 0) { s_red[t >> 6] = mx; s_red[8 + (t >> 6)] = mn; s_red[16 + (t >> 6)] = wxnan ? 1.0 : 0.0; }
    __syncthreads();
    mx = s_red[0]; mn = s_red[8];
    double f = s_red[16];
#pragma unroll
    for (int w = 1; w < 4; ++w) {
        mx = s_red[w] > mx ? s_red[w] : mx;
        mn = s_red[8 + w] < mn ? s_red[8 + w] : mn;
        f += s_red[16 + w];
    }
    if (f != 0.0) { mx = NAN; mn = NAN; }
    if (t == 0) { bmax[blockIdx.x] = mx; bmin[blockIdx.x] = mn; }
    const bool close = __builtin_fabs(mx - mn) <= ISCLOSE_ATOL;  // false on NaN
    const bool need_soft = softmax != 0, need_raw = !softmax || close;
    bool nan = false;
    double Wa = 0.0, Wm = 0.0;
    if (need_soft) {
        double e[SCAN_CHUNK];
#pragma unroll
        for (int k = 0; k < SCAN_CHUNK; ++k) {
            e[k] = exp_spec(x[k] - 1.0);
            const int64_t i = base + k * 256 + t;
            if (i < N) e_out[i] = e[k];
        }
        scan_variant(e, okm, base, N, s_a, s_m, s_gtot, lp_soft, gend_soft, ggend_soft, Wa, Wm, nan);
        if (t == 0) { bsum_e[blockIdx.x] = Wa; btot_soft[blockIdx.x] = Wm; }
    }
    if (need_raw) {
#pragma unroll
        for (int k = 0; k < SCAN_CHUNK; ++k) {
            const int64_t i = base + k * 256 + t;
            if (i < N) x_out[i] = x[k];
        }
        bool nan_raw = false;
        scan_variant(x, okm, base, N, s_a, s_m, s_gtot, lp_raw, gend_raw, ggend_raw, Wa, Wm, nan_raw);
        if (t == 0) btot_raw[blockIdx.x] = Wm;
        if (!need_soft) nan = nan_raw;  // with the softmax on, x NaN <=> e NaN: counted once
    } else if (t == 0) {
        btot_raw[blockIdx.x] = 0.0;
    }
    const bool wnan = __any(nan);
    if (wnan && (t & 63) == 0) {
        atomicOr(&status[0], 2);
        if (flags_out) atomicAdd(&flags_out[0], 1.0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) kept += __shfl_xor(kept, o);
    if ((t & 63) == 0 && kept) {
        atomicAdd(&status[1], kept);
        if (flags_out) atomicAdd(&flags_out[1], (double)kept);  // exact: integers far below 2^53
    }
}

// k_tail_a2 in the chunk-per-thread view (tail_block.hpp): every thread reads and writes its own 16-slot chunk from one
// address, nothing goes through LDS.  Same outputs, 5.5 us instead of 8.0 at N = 100k (two barriers and two LDS round
// trips fewer on a latency-bound kernel).  Single trajectory, N >= 16 (smaller sets: k_tail_a2).
// Extra workgroups of the tail (blockIdx.x >= nb): the prediction list of the NEXT frame's sparse scoring.  Every row whose
// stamp is this frame's epoch was somebody's nearest entry in this frame (claimed by its first particle, or confirmed from
// the previous list): it goes on the list - order is immaterial, one counter bump per wave - and is re-stamped epoch + 1,
// the tag the next front (epoch + 2) honours as "being scored by my streaming waves".  A cloud moves a fraction of the
// codebook's spacing per frame, so most of the rows it needs were needed the frame before: they are then scored by
// balanced, coalesced streaming waves instead of by whichever particle wave touches them first (in the frames after a
// wide start a wave claimed up to 64 rows = 16 rounds of cold 8 KB fetches, and the kernel ends with its slowest wave).
constexpr int PREDICT_PER_THREAD = 16;
// One workgroup lists the rows of its 256 x PREDICT_PER_THREAD stamps that carry this frame's epoch: counts per thread, a prefix
// over the wave (DPP), the waves' totals through LDS, ONE bump of the list's counter per workgroup.  (One bump per wave with four
// stamps a thread was 196 serialised read-modify-writes of one address whenever most waves had rows to list - the frames after
// a wide start, 10^4 rows in use: 3 - 5 us of the tail kernel there.)  s_wt: 8 ints of LDS.
MD void predict_scan(const ScorePredict& pr, int blk, int* s_wt) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int64_t k0 = ((int64_t)blk * 256 + threadIdx.x) * PREDICT_PER_THREAD;
    uint32_t st[PREDICT_PER_THREAD];
    if (k0 + PREDICT_PER_THREAD <= pr.K) {
#pragma unroll
        for (int q = 0; q < PREDICT_PER_THREAD / 4; ++q) {
            const uint4 v = reinterpret_cast<const uint4*>(pr.stamps + k0)[q];
            st[4 * q] = v.x; st[4 * q + 1] = v.y; st[4 * q + 2] = v.z; st[4 * q + 3] = v.w;
        }
    } else {
#pragma unroll
        for (int j = 0; j < PREDICT_PER_THREAD; ++j) st[j] = k0 + j < pr.K ? pr.stamps[k0 + j] : 0u;
    }
    // listed: the rows this frame used (stamp = epoch) and the rows that were on this frame's list unused (stamp = epoch - 1, the
    // frame's pred_tag: their second chance, ScorePredict); a second chance that was not taken (bit 31 set) is dropped
    const uint32_t unused = pr.epoch - 1u;
    // keep(st): the row goes on the next list
    auto keep = [&](uint32_t v) { return v == pr.epoch || ((v & ~PRED_SECOND) == unused && (v >> 30) < (uint32_t)MIDAS_PRED_CHANCES); };
    int n = 0;
#pragma unroll
    for (int j = 0; j < PREDICT_PER_THREAD; ++j) n += (k0 + j < pr.K && keep(st[j])) ? 1 : 0;
    const int incl = wave_iscan_dpp(n);
    if (lane == 63) s_wt[w] = incl;
    __syncthreads();
    const int t0 = s_wt[0], t1 = s_wt[1], t2 = s_wt[2], t3 = s_wt[3];
    const int total = t0 + t1 + t2 + t3;
    if (total == 0) return;  // (uniform over the workgroup)
    if (threadIdx.x == 0) s_wt[4] = atomicAdd(pr.count, total);
    __syncthreads();
    int pos = s_wt[4] + (w > 0 ? t0 : 0) + (w > 1 ? t1 : 0) + (w > 2 ? t2 : 0) + incl - n;
#pragma unroll
    for (int j = 0; j < PREDICT_PER_THREAD; ++j)
        if (k0 + j < pr.K && keep(st[j])) {
            // (distinct rows, counter zeroed by the front: pos < K; a row that would not fit simply stays unlisted and untagged -
            // its first particle of the next frame claims it)
            if (pos < pr.K) {
                pr.list[pos] = (int32_t)(k0 + j);
                pr.stamps[k0 + j] = st[j] == pr.epoch ? pr.epoch + 1u : ((pr.epoch + 1u) | ((st[j] & PRED_SECOND) + PRED_AGE1));
            }
            ++pos;
        }
}

// prediction list from scratch (the particle set was replaced: projection onto the codebook, filter/filter.py:159-160): the rows
// idx[n] are stamped `epoch`, predict_scan then lists them and tags them epoch + 1 for the frame with epoch + 2
__global__ __launch_bounds__(256) void k_predict_mark(int64_t N, const int32_t* __restrict__ idx, uint32_t* __restrict__ stamps, int64_t K, uint32_t epoch) {
    const int64_t n = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (n >= N) return;
    const int32_t r = idx[n];
    if (r >= 0 && r < K) stamps[r] = epoch;  // every marker of a row stores the same value
}
__global__ __launch_bounds__(256) void k_predict_scan(ScorePredict pr) {
    __shared__ int s_wt[8];
    predict_scan(pr, (int)blockIdx.x, s_wt);
}

int launch_predict_seed(midas_ctx* ctx, int64_t N, const int32_t* idx, const ScorePredict& pr) {
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(pr.count, 0, sizeof(int32_t), ctx->stream));
    hipLaunchKernelGGL(k_predict_mark, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, idx, pr.stamps, pr.K, pr.epoch);
    hipLaunchKernelGGL(k_predict_scan, dim3((unsigned)ceil_div(pr.K, 256 * PREDICT_PER_THREAD)), dim3(256), 0, ctx->stream, pr);
    MIDAS_HIP_CHECK(ctx, hipGetLastError());
    return MIDAS_OK;
}

__global__ __launch_bounds__(256) void k_tail_a2d(int64_t N, const double* __restrict__ scores, const int32_t* __restrict__ nn_idx,
                                                  const uint8_t* __restrict__ valid, int32_t softmax, TailTables tb, bool padded,
                                                  int32_t* __restrict__ status, double* __restrict__ flags_out,
                                                  const double* __restrict__ part_rmse, int nrm, double* __restrict__ rmse_out,
                                                  int64_t score_stride = 0, int64_t tstride = 0, int nb_tail = 0x7fffffff,
                                                  ScorePredict pr = ScorePredict(), bool rmse_raw = false) {
    __shared__ double s_gtot[32];
    __shared__ double s_red[24];
    __shared__ uint32_t s_gh[TAIL_GUIDE_LDS];
    if ((int)blockIdx.x >= nb_tail) {  // (single trajectory only: the launcher adds these workgroups when pr.stamps is set)
        predict_scan(pr, (int)blockIdx.x - nb_tail, reinterpret_cast<int*>(s_red));
        return;
    }
    if (blockIdx.y) {  // pipelined batch: trajectory blockIdx.y - its own table block (tables_of layout), scores, arrays, rmse triple
        const int64_t b = blockIdx.y, o = b * N, ts = b * tstride;
        scores += b * score_stride; nn_idx += o; valid += o; status += 2 * b;
        tb.e += ts; tb.x_raw += ts; tb.lp += ts; tb.lp_raw += ts; tb.gend += ts; tb.gend_raw += ts; tb.ggend += ts; tb.ggend_raw += ts;
        tb.bsum_e += ts; tb.btot += ts; tb.btot_raw += ts; tb.bmax += ts; tb.bmin += ts;
        if (part_rmse) { part_rmse += 2 * b * nrm; rmse_out += 3 * b; }
        tb.guide = nullptr; tb.guide_raw = nullptr;  // (single trajectory only)
    }
    int kept = 0;
    bool nan = false;
    tail_a_direct(N, (int)blockIdx.x, scores, nn_idx, valid, softmax, tb, padded, s_gtot, s_red, kept, nan, tb.guide ? s_gh : nullptr);
    const int t = threadIdx.x;
    if (t == 0) {
        if (nan) atomicOr(&status[0], 2);
        if (kept) atomicAdd(&status[1], kept);
        if (flags_out) {  // sharded exchange record: NaN marker (any non-zero) and kept count (exact: integers far below 2^53)
            if (nan) atomicAdd(&flags_out[0], 1.0);
            if (kept) atomicAdd(&flags_out[1], (double)kept);
        }
    }
    if (rmse_out && blockIdx.x == 0) {  // the frame's rmse from the front kernel's per-wave sums (same order as k_tail_b2)
        double p = 0.0, q = 0.0;
        for (int k = t; k < nrm; k += 256) { p += part_rmse[2 * k]; q += part_rmse[2 * k + 1]; }
        p = wsum(p);
        q = wsum(q);
        __syncthreads();
        if ((t & 63) == 0) { s_red[t >> 6] = p; s_red[4 + (t >> 6)] = q; }
        __syncthreads();
        if (t == 0) {
            p = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
            q = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
            if (rmse_raw) {  // a shard's sums (added up across the ranks by whoever reads the exchange records): k_reduce_partials' order
                rmse_out[0] = p;
                rmse_out[1] = q;
                return;
            }
            rmse_out[0] = __builtin_sqrt(p / (double)N);
            rmse_out[1] = __builtin_sqrt(q / (double)N);
            rmse_out[2] = (double)wall_clock64() * 0.01;  // device wall clock (100 MHz) in us: where this frame ended
        }
    }
}

// The tail with one WAVE per 256-slot group (tail_group.hpp): workgroups [0, nwg) hold four groups each, workgroup nwg (when
// rmse_out is set) adds up the front's rmse sums as k_tail_a2d's block 0 does, the workgroups behind it list the next frame's rows.
__global__ __launch_bounds__(256) void k_tail_a3(TailGroupArgs a, int ngroups, int nwg, const double* __restrict__ part_rmse, int nrm,
                                                 double* __restrict__ rmse_out, bool rmse_raw, ScorePredict pr) {
    __shared__ double s_E[4][2 * TG_GROUP / GUIDE_UNIT];
    __shared__ double s_red[24];
    int b = (int)blockIdx.x;
    const int t = threadIdx.x;
    if (b < nwg) {
        const int wv = __builtin_amdgcn_readfirstlane(t >> 6), G = 4 * b + wv;  // (wave-uniform: the group's own conditions are scalar branches)
        if (G < ngroups) tail_group_wave(a, G, s_E[wv]);
        return;
    }
    b -= nwg;
    if (rmse_out) {
        if (b == 0) {  // the frame's rmse from the front kernel's per-wave sums (same order as k_tail_b2 / k_tail_a2d)
            TG_SPAN(2048, wall_clock64());
            double p = 0.0, q = 0.0;
            for (int k = t; k < nrm; k += 256) { p += part_rmse[2 * k]; q += part_rmse[2 * k + 1]; }
            p = wsum(p);
            q = wsum(q);
            if ((t & 63) == 0) { s_red[t >> 6] = p; s_red[4 + (t >> 6)] = q; }
            __syncthreads();
            if (t == 0) {
                p = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
                q = (s_red[4] + s_red[5]) + (s_red[6] + s_red[7]);
                if (rmse_raw) {
                    rmse_out[0] = p;
                    rmse_out[1] = q;
                } else {
                    rmse_out[0] = __builtin_sqrt(p / (double)a.N);
                    rmse_out[1] = __builtin_sqrt(q / (double)a.N);
                    rmse_out[2] = (double)wall_clock64() * 0.01;  // device wall clock (100 MHz) in us
                }
            }
            TG_SPAN(2049, wall_clock64());
            return;
        }
        b -= 1;
    }
    predict_scan(pr, b, reinterpret_cast<int*>(s_red));
}

constexpr int TB_MAX_BLOCKS = 1024;  // 4 M particles (per GPU in the fused step, in total in the sharded step)

// TF (sharded path): every rank holds the gathered exchange records r1_all (one per rank, rec = 5 nb + 4 doubles:
//     nb block sums of e | nb block totals of e*valid | nb block totals of x*valid | nb block max x | nb block min x |
//     NaN count | kept count | sum |dt|^2 | sum angle^2 ) written by TA2 on each shard.  The isclose guard is decided
//     from the gathered extrema (as TB2 does on one GPU) and picks the softmax or the raw variant; then
//     weights = e / S * valid and cdf = (BP + lp) / total with S, BP, total summed sequentially over ALL shards'
//     blocks in global block order; the globally last slot is forced to 1; block 0 finalises status and rmse.
__global__ __launch_bounds__(256) void k_tail_fin(int64_t N, const double* __restrict__ e, const double* __restrict__ x_raw,
                                                  const double* __restrict__ lp, const double* __restrict__ lp_raw,
                                                  const uint8_t* __restrict__ valid,
                                                  double* __restrict__ weights, double* __restrict__ cdf_io, int G, int nb,
                                                  const double* __restrict__ r1_all, int rank, double n_total,
                                                  int32_t softmax, double* __restrict__ rmse_out,
                                                  int32_t* __restrict__ status) {
    __shared__ double s_w[TB_MAX_BLOCKS];
    __shared__ double s_se[TB_MAX_BLOCKS];
    __shared__ double s_ex[12];
    __shared__ double s_tot[3];
    __shared__ int s_apply;
    const int t = threadIdx.x;
    const int nb_all = G * nb, rec = 5 * nb + 4;
    const int my = rank * nb + (int)blockIdx.x;
    auto field = [&](int f, int i) { return r1_all[(int64_t)(i / nb) * rec + (int64_t)f * nb + (i % nb)]; };
    double mx = -INFINITY, mn = INFINITY;
    bool nan = false;
    for (int i = t; i < nb_all; i += 256) {
        const double u = field(3, i), v = field(4, i);
        nan |= (u != u) || (v != v);
        mx = u > mx ? u : mx;
        mn = v < mn ? v : mn;
        s_w[i] = field(1, i);
        s_se[i] = field(0, i);
    }
    mx = wmax(mx);
    mn = wmin(mn);
    const bool wn = __any(nan);
    if ((t & 63) == 0) { s_ex[t >> 6] = mx; s_ex[4 + (t >> 6)] = mn; s_ex[8 + (t >> 6)] = wn ? 1.0 : 0.0; }
    __syncthreads();
    if (t == 0) {
        mx = s_ex[0]; mn = s_ex[4];
        double f = s_ex[8];
        for (int w = 1; w < 4; ++w) { mx = s_ex[w] > mx ? s_ex[w] : mx; mn = s_ex[4 + w] < mn ? s_ex[4 + w] : mn; f += s_ex[8 + w]; }
        if (f != 0.0) { mx = NAN; mn = NAN; }
        s_apply = (softmax && !(__builtin_fabs(mx - mn) <= ISCLOSE_ATOL)) ? 1 : 0;
    }
    __syncthreads();
    const bool apply = s_apply != 0;
    if (!apply) {
        for (int i = t; i < nb_all; i += 256) s_w[i] = field(2, i);
        __syncthreads();
    }
    if (t == 0) {
        double bp = 0.0, total = 0.0, S = 0.0;
        for (int i = 0; i < nb_all; ++i) { if (i == my) bp = total; total = total + s_w[i]; S = S + s_se[i]; }
        s_tot[0] = bp; s_tot[1] = total; s_tot[2] = apply ? S : 1.0;
    }
    __syncthreads();
    const double bp = s_tot[0], total = s_tot[1], S = s_tot[2];
    if (blockIdx.x == 0 && t == 0) {
        double nans = 0.0, kept = 0.0, st2 = 0.0, sr2 = 0.0;
        for (int r = 0; r < G; ++r) {
            const double* fl = r1_all + (int64_t)r * rec + 5 * nb;
            nans += fl[0]; kept += fl[1]; st2 += fl[2]; sr2 += fl[3];
        }
        int st = nans != 0.0 ? 2 : 0;
        if (total != total) st |= 2;
        else if (total == 0.0) st |= 1;
        status[0] = st;
        status[1] = (int32_t)kept;
        if (rmse_out) { rmse_out[0] = __builtin_sqrt(st2 / n_total); rmse_out[1] = __builtin_sqrt(sr2 / n_total); }
    }
    const double* __restrict__ ev = apply ? e : x_raw;
    const double* __restrict__ lpv = apply ? lp : lp_raw;
    const bool is_last = rank == G - 1;
    const int64_t base = (int64_t)blockIdx.x * SCAN_BLOCK;
    double ee[SCAN_CHUNK], ll[SCAN_CHUNK];
    uint8_t ok[SCAN_CHUNK];
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {  // unconditional loads on clamped slots
        const int64_t i = base + (int64_t)j * 256 + t, ic = i < N ? i : N - 1;
        ee[j] = ev[ic]; ll[j] = lpv[ic]; ok[j] = valid[ic];
    }
#pragma unroll
    for (int j = 0; j < SCAN_CHUNK; ++j) {
        const int64_t i = base + (int64_t)j * 256 + t;
        if (i < N) {
            weights[i] = (ee[j] / S) * (ok[j] ? 1.0 : 0.0);
            cdf_io[i] = (is_last && i == N - 1) ? 1.0 : (bp + ll[j]) / total;
        }
    }
}

// TB (single-GPU path): everything after TA in one kernel.  Each workgroup rebuilds the small tables (S,
// total, BP[b], cdf at the block ends) in LDS, writes the weights of its own 256 slots, then resamples
// them: the block holding the draw is found in the LDS table, the slot by a 12-step search over the
// block-local prefix with cdf(i) = (BP_b + lp_i) / total evaluated on the fly - the same values the
// sharded path materialises, so both give identical indices.
MD double cdf_at(const double* __restrict__ lp, const double* s_bp, double total, int64_t i, int64_t N) {
    return (i == N - 1) ? 1.0 : (s_bp[i >> 12] + lp[i]) / total;
}

struct TailBArgs {
    int64_t N;
    int nb;
    const double* e;
    const uint8_t* valid;
    const double* lp;
    const double* block_sums_e;
    const double* block_totals_em;
    const int32_t* flag;
    int32_t* status;
    double* weights;       // out: e/S*valid of the own slots
    int32_t mode;
    const double* u;
    float u32;
    uint64_t seed, step;
    int32_t* ridx;
    const float* poses_prop;
    float* poses_out;
    double* weights_out;
    const int32_t* nn_idx;
    int32_t* hint_out;
    const double* part_rmse;
    int nrm;
    double* rmse_out;
    int64_t slot_base;     // Philox key offset of slot 0 (b * N for trajectory b of a batch)
};

__global__ __launch_bounds__(256) void k_tail_b(TailBArgs a) {
    __shared__ double s_bp[TB_MAX_BLOCKS];
    __shared__ double s_end[TB_MAX_BLOCKS];
    __shared__ double s_tot[2];
    if (blockIdx.y) {  // batch of trajectories
        const int64_t b = blockIdx.y, o = b * a.N;
        a.e += o; a.valid += o; a.lp += o; a.block_sums_e += b * a.nb; a.block_totals_em += b * a.nb;
        a.flag += b; a.status += 2 * b; a.weights += o;
        if (a.u) a.u += o;
        a.ridx += o; a.poses_prop += o * 16; a.poses_out += o * 16; a.weights_out += o; a.nn_idx += o; a.hint_out += o;
        if (a.part_rmse) { a.part_rmse += 2 * b * a.nrm; a.rmse_out += 2 * b; }
        a.slot_base += o;
    }
    const bool apply = a.flag[0] != 0;
    // fetch the block partials in parallel (s_bp <- totals of e*valid, s_end <- sums of e), then one thread
    // turns them into the sequential prefixes the spec asks for - no dependent global loads
    for (int b = threadIdx.x; b < a.nb; b += 256) { s_bp[b] = a.block_totals_em[b]; s_end[b] = apply ? a.block_sums_e[b] : 0.0; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double acc = 0.0;
        for (int b = 0; b < a.nb; ++b) { const double w = s_bp[b]; s_bp[b] = acc; acc = acc + w; }
        s_tot[0] = acc;
        double S = 1.0;
        if (apply) {
            S = 0.0;
            for (int b = 0; b < a.nb; ++b) S = S + s_end[b];
        }
        s_tot[1] = S;
    }
    __syncthreads();
    const double total = s_tot[0], S = s_tot[1];
    const int64_t N = a.N;
    // cdf at the last slot of every block (the last block ends at N-1, forced to 1)
    for (int b = threadIdx.x; b < a.nb; b += 256) {
        const int64_t last = ((int64_t)(b + 1) << 12) - 1 < N - 1 ? ((int64_t)(b + 1) << 12) - 1 : N - 1;
        s_end[b] = cdf_at(a.lp, s_bp, total, last, N);
    }
    __syncthreads();
    const bool bad_total = !(total == total) || total == 0.0;
    const int st0 = a.status[0];
    const bool usable = st0 == 0 && !bad_total;
    if (blockIdx.x == 0 && threadIdx.x == 0 && bad_total) a.status[0] = st0 | ((total != total) ? 2 : 1);
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < N) {
        a.weights[i] = (a.e[i] / S) * (a.valid[i] ? 1.0 : 0.0);
        int32_t src = (int32_t)i;
        if (usable) {
            double t;
            bool upper;
            if (a.mode == MIDAS_RESAMPLE_MULTINOMIAL) {
                t = a.u ? a.u[i] : philox_uniform53((uint64_t)(a.slot_base + i), a.seed, a.step);
                upper = false;
            } else {
                const float r = a.u32 >= 0.0f ? a.u32 : philox_uniform24(a.seed + (uint64_t)blockIdx.y, a.step);
                const float off = r / (float)N;
                t = (double)i / (double)N + (double)off;
                t = t >= 1.0 ? t - 1.0 : t;
                upper = true;
            }
            // block: first b whose end value is >= t (lower) / > t (upper)
            int lo = 0, hi = a.nb;
            while (hi > lo) {
                const int mid = lo + ((hi - lo) >> 1);
                const double c = s_end[mid];
                if (upper ? (c <= t) : (c < t)) lo = mid + 1; else hi = mid;
            }
            if (lo >= a.nb) {
                src = (int32_t)(N - 1);
            } else {
                // Inside the 4096-slot block: a binary search on the division-free comparison (BP + lp_i) vs t * total
                // locates the slot to within rounding (probes are what this kernel pays for: 12 is the minimum);
                // the exact predicate cdf_i = (BP + lp_i) / total < t (<= for the systematic mode) then walks to the
                // true boundary - it is monotone in i, so the result is exactly the lower/upper bound over the cdf
                // values the sharded path materialises.
                const int64_t b_lo = (int64_t)lo << 12, b_hi = b_lo + SCAN_BLOCK < N ? b_lo + SCAN_BLOCK : N;
                const double bp = s_bp[lo], tt = t * total;
                int64_t l2 = b_lo, h2 = b_hi;
                while (h2 > l2) {
                    const int64_t mid = l2 + ((h2 - l2) >> 1);
                    const double c = bp + a.lp[mid];
                    // (a negative total - raw weights of a negative cosine - turns the division-free comparison round)
                    const bool lft = total < 0.0 ? (upper ? (c >= tt) : (c > tt)) : (upper ? (c <= tt) : (c < tt));
                    if (lft) l2 = mid + 1; else h2 = mid;
                }
                if (l2 >= b_hi) l2 = b_hi - 1;
                // exact fix-up
                while (l2 > b_lo) {
                    const double c = cdf_at(a.lp, s_bp, total, l2 - 1, N);
                    if (upper ? (c <= t) : (c < t)) break;
                    --l2;
                }
                while (l2 < b_hi - 1) {
                    const double c = cdf_at(a.lp, s_bp, total, l2, N);
                    if (!(upper ? (c <= t) : (c < t))) break;
                    ++l2;
                }
                src = (int32_t)(l2 < N ? l2 : N - 1);
            }
        }
        a.ridx[i] = src;
        const float4* ps = reinterpret_cast<const float4*>(a.poses_prop + (int64_t)src * 16);
        float4* pd = reinterpret_cast<float4*>(a.poses_out + i * 16);
        float4 r0 = ps[0], r1 = ps[1], r2 = ps[2], r3 = ps[3];
        pd[0] = r0; pd[1] = r1; pd[2] = r2; pd[3] = r3;
        a.weights_out[i] = (a.e[src] / S) * (a.valid[src] ? 1.0 : 0.0);
        a.hint_out[i] = a.nn_idx[src];
    }
    if (a.part_rmse && blockIdx.x == 0) {
        __shared__ double sa[4], sb[4];
        double p = 0.0, q = 0.0;
        for (int k = threadIdx.x; k < a.nrm; k += 256) { p += a.part_rmse[2 * k]; q += a.part_rmse[2 * k + 1]; }
        p = wsum(p);
        q = wsum(q);
        if ((threadIdx.x & 63) == 0) { sa[threadIdx.x >> 6] = p; sb[threadIdx.x >> 6] = q; }
        __syncthreads();
        if (threadIdx.x == 0) {
            p = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            q = (sb[0] + sb[1]) + (sb[2] + sb[3]);
            a.rmse_out[0] = __builtin_sqrt(p / (double)N);
            a.rmse_out[1] = __builtin_sqrt(q / (double)N);
        }
    }
}

// TB2 (fused single-trajectory step, after TA2): decides the isclose guard from TA2's per-block extrema, then
// weights + resample + gather like TB, with the search restructured around round trips: the cumulative value
// at the end of every 16-slot chunk (TA2's chunk-end table + block prefix) sits in LDS, so a draw is located
// to its chunk without touching memory; four probes inside the chunk and the exact fix-up follow, and the
// gathers of the winner's pose / weight / hint travel together.  Above TB2_TAB chunks per LDS table the table
// holds every 2^cshift-th chunk end and the chunk is found with cshift probes of the global table.
constexpr int TB2_TAB = 8192;
#ifdef MIDAS_DEBUG_CLOCKS  // phase clocks of one workgroup (tools/variants.sh dbg "-DMIDAS_DEBUG_CLOCKS"; tools/tb2_clocks.py)
__device__ long long g_tb2_clk[16];
__device__ long long g_ta_clk[16];
#define TB2_CLK(k) if (blockIdx.x == 97 && threadIdx.x == 64) g_tb2_clk[k] = clock64();
#define TB2_WALL(k) if (threadIdx.x == 64) { if (blockIdx.x == 0) g_tb2_clk[8 + k] = wall_clock64(); if (blockIdx.x == 195) g_tb2_clk[10 + k] = wall_clock64(); if (blockIdx.x == 390) g_tb2_clk[12 + k] = wall_clock64(); }
#else
#define TB2_CLK(k)
#define TB2_WALL(k)
#endif

struct TailB2Args {
    int64_t N;
    int nb, ng, nt, cshift;   // blocks, chunks, table entries, chunks per table entry (log2)
    const double *e, *x_raw, *lp, *lp_raw, *gend, *gend_raw, *bsum_e, *btot, *btot_raw, *bmax, *bmin;
    const uint8_t* valid;
    int32_t softmax;
    int32_t* status;
    double* weights;
    int32_t mode;
    const double* u;
    float u32;
    uint64_t seed, step;
    int32_t* ridx;
    const float* poses_prop;
    float* poses_out;
    double* weights_out;
    const int32_t* nn_idx;
    int32_t* hint_out;
    const double* part_rmse;
    int nrm;
    double* rmse_out;
    int64_t tstride;  // > 0: batch with one table block per trajectory (pipelined batch), 0: array-major batch tables
};

__global__ __launch_bounds__(256) void k_tail_b2(TailB2Args a) {
    // dynamic LDS sized to this launch (nt + 3 nb doubles) so that small N keeps several workgroups per CU
    extern __shared__ double s_dyn[];
    double* s_tab = s_dyn;            // [nt] block-local prefix at the chunk ends (TA2's table, every 2^cshift-th)
    double* s_bp = s_tab + a.nt;      // [nb] exclusive prefix of the block totals of e*valid
    double* s_w = s_bp + a.nb;        // [nb] block totals of e*valid
    double* s_se = s_w + a.nb;        // [nb] block sums of e
    __shared__ double s_ex[12];
    __shared__ double s_tot[2];
    __shared__ int s_apply;
    if (blockIdx.y) {  // batch of trajectories (plain strides)
        const int64_t b = blockIdx.y, o = b * a.N;
        if (a.tstride) {
            const int64_t ts = b * a.tstride;
            a.e += ts; a.x_raw += ts; a.lp += ts; a.lp_raw += ts; a.gend += ts; a.gend_raw += ts;
            a.bsum_e += ts; a.btot += ts; a.btot_raw += ts; a.bmax += ts; a.bmin += ts;
        } else {
            a.e += o; a.x_raw += o; a.lp += o; a.lp_raw += o; a.gend += b * a.ng; a.gend_raw += b * a.ng;
            a.bsum_e += b * a.nb; a.btot += b * a.nb; a.btot_raw += b * a.nb; a.bmax += b * a.nb; a.bmin += b * a.nb;
        }
        a.valid += o; a.status += 2 * b; a.weights += o;
        if (a.u) a.u += o;
        a.ridx += o; a.poses_prop += o * 16; a.poses_out += o * 16; a.weights_out += o; a.nn_idx += o; a.hint_out += o;
        if (a.part_rmse) { a.part_rmse += 2 * b * a.nrm; a.rmse_out += 2 * b; }
    }
    const int64_t slot_base = (int64_t)blockIdx.y * a.N;  // Philox key offset of slot 0
    const int t = threadIdx.x;
    const int64_t N = a.N;
    const int64_t i = (int64_t)blockIdx.x * 256 + t, ic = i < N ? i : N - 1;
    TB2_CLK(0)
    TB2_WALL(0)
    // ---- round trip 1: everything that does not depend on the guard (the softmax variant is the common one).
    // Loads sit in uniform branches only (a per-lane conditional load would be waited for one at a time).
    constexpr int TPT = TB2_TAB / 256, BPT = TB_MAX_BLOCKS / 256;
    const int nk = (a.nt + 255) >> 8, nbk = (a.nb + 255) >> 8;
    const int ng1 = a.ng - 1, nb1 = a.nb - 1, cs = a.cshift;
    double gv[TPT], bt[BPT], bs[BPT], bx[BPT], bn[BPT];
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        gv[k] = 0.0;
        if (k < nk) {
            int c = ((k * 256 + t + 1) << cs) - 1;
            c = c < ng1 ? c : ng1;
            gv[k] = a.gend[c];
        }
    }
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        bt[k] = 0.0; bs[k] = 0.0; bx[k] = 0.0; bn[k] = 0.0;
        if (k < nbk) {
            const int b = k * 256 + t, bc = b < nb1 ? b : nb1;
            bt[k] = a.btot[bc];
            bs[k] = a.bsum_e[bc];
            bx[k] = a.bmax[bc];
            bn[k] = a.bmin[bc];
        }
    }
    double e_i = a.e[ic];
    const bool ok_i = a.valid[ic] != 0;
    const double u_i = a.u ? a.u[ic] : 0.0;
    TB2_CLK(1)
    // ---- guard: global extrema of x from TA2's per-block ones (NaN propagates)
    double mx = -INFINITY, mn = INFINITY;
    bool nan = false;
#pragma unroll
    for (int k = 0; k < BPT; ++k) {
        const int b = k * 256 + t;
        const bool in = b < a.nb;
        nan |= in && ((bx[k] != bx[k]) || (bn[k] != bn[k]));
        mx = in && bx[k] > mx ? bx[k] : mx;
        mn = in && bn[k] < mn ? bn[k] : mn;
        if (in) { s_w[b] = bt[k]; s_se[b] = bs[k]; }
    }
#pragma unroll
    for (int k = 0; k < TPT; ++k) {
        const int j = k * 256 + t;
        if (j < a.nt) s_tab[j] = gv[k];
    }
    mx = wmax(mx);
    mn = wmin(mn);
    const bool wn = __any(nan);
    if ((t & 63) == 0) { s_ex[t >> 6] = mx; s_ex[4 + (t >> 6)] = mn; s_ex[8 + (t >> 6)] = wn ? 1.0 : 0.0; }
    __syncthreads();
    if (t == 0) {
        mx = s_ex[0]; mn = s_ex[4];
        double f = s_ex[8];
        for (int w = 1; w < 4; ++w) { mx = s_ex[w] > mx ? s_ex[w] : mx; mn = s_ex[4 + w] < mn ? s_ex[4 + w] : mn; f += s_ex[8 + w]; }
        if (f != 0.0) { mx = NAN; mn = NAN; }
        const bool apply = a.softmax && !(__builtin_fabs(mx - mn) <= ISCLOSE_ATOL);
        s_apply = apply ? 1 : 0;
        if (apply) {  // sequential sums in block order (the spec); reads and writes on different arrays so they pipeline
            double acc = 0.0, S = 0.0;
            for (int b = 0; b < a.nb; ++b) { s_bp[b] = acc; acc = acc + s_w[b]; S = S + s_se[b]; }
            s_tot[0] = acc;
            s_tot[1] = S;
        }
    }
    __syncthreads();
    const bool apply = s_apply != 0;
    const double* __restrict__ lp = a.lp;
    const double* __restrict__ esrc = a.e;
    const double* __restrict__ gend = a.gend;
    if (!apply) {
        // rare: every particle has the same score (or the softmax is off) - switch to the raw variant TA2 wrote
        lp = a.lp_raw; esrc = a.x_raw; gend = a.gend_raw;
        for (int b = t; b < a.nb; b += 256) s_w[b] = a.btot_raw[b];
        for (int j = t; j < a.nt; j += 256) {
            int c = ((j + 1) << cs) - 1;
            c = c < ng1 ? c : ng1;
            s_tab[j] = gend[c];
        }
        e_i = esrc[ic];
        __syncthreads();
        if (t == 0) {
            double acc = 0.0;
            for (int b = 0; b < a.nb; ++b) { s_bp[b] = acc; acc = acc + s_w[b]; }
            s_tot[0] = acc;
            s_tot[1] = 1.0;
        }
        __syncthreads();
    }
    TB2_CLK(2)
    const double total = s_tot[0], S = s_tot[1];
    const bool bad_total = !(total == total) || total == 0.0;
    const int st0 = a.status[0];
    const bool usable = st0 == 0 && !bad_total;
    if (blockIdx.x == 0 && t == 0 && bad_total) a.status[0] = st0 | ((total != total) ? 2 : 1);
    TB2_CLK(3)
    if (i < N) {
        a.weights[i] = (e_i / S) * (ok_i ? 1.0 : 0.0);
        int64_t src = i;
        if (usable) {
            double tq;
            bool upper;
            if (a.mode == MIDAS_RESAMPLE_MULTINOMIAL) {
                tq = a.u ? u_i : philox_uniform53((uint64_t)(slot_base + i), a.seed, a.step);
                upper = false;
            } else {
                const float r = a.u32 >= 0.0f ? a.u32 : philox_uniform24(a.seed + (uint64_t)blockIdx.y, a.step);
                const float off = r / (float)N;
                tq = (double)i / (double)N + (double)off;
                tq = tq >= 1.0 ? tq - 1.0 : tq;
                upper = true;
            }
            const double tt = tq * total;
            // "still left of the answer": cumulative value < tt (multinomial, lower bound) / <= tt (systematic, upper bound)
            // (a negative total - raw weights of a negative cosine - turns the division-free comparison round: resample_search.hpp)
            const bool neg = total < 0.0;
            auto left = [&](double c) { return neg ? (upper ? (c >= tt) : (c > tt)) : (upper ? (c <= tt) : (c < tt)); };
            auto left_exact = [&](double c) { return upper ? (c <= tq) : (c < tq); };
            // cumulative e*valid at the end of table entry j
            auto tab = [&](int j) {
                int c = ((j + 1) << cs) - 1;
                c = c < ng1 ? c : ng1;
                return s_bp[c >> 8] + s_tab[j];
            };
            // table entry: first j with !left(tab(j)); 4-ary rounds (three independent LDS probes each), then binary
            int lo = 0, hi = a.nt;
            while (hi - lo >= 4) {
                const int q = (hi - lo) >> 2;
                const int m1 = lo + q, m2 = m1 + q, m3 = m2 + q;
                const bool p1 = left(tab(m1)), p2 = left(tab(m2)), p3 = left(tab(m3));
                if (p3) lo = m3 + 1;
                else if (p2) { lo = m2 + 1; hi = m3; }
                else if (p1) { lo = m1 + 1; hi = m2; }
                else hi = m1;
            }
            while (hi > lo) {
                const int mid = lo + ((hi - lo) >> 1);
                if (left(tab(mid))) lo = mid + 1; else hi = mid;
            }
            TB2_CLK(4)
            if (lo >= a.nt) lo = a.nt - 1;
            // chunk inside the entry (only when one entry spans several chunks)
            int64_t c_lo = (int64_t)lo << cs, c_hi = c_lo + ((int64_t)1 << cs);
            c_hi = c_hi < a.ng ? c_hi : a.ng;
            while (c_hi - c_lo > 1) {
                const int64_t mid = c_lo + ((c_hi - c_lo) >> 1);
                const double c = s_bp[(mid - 1) >> 8] + gend[mid - 1];
                if (left(c)) c_lo = mid; else c_hi = mid;
            }
            // the chunk's sixteen prefix values in one round trip; slot = number of them still left of the answer
            const int64_t s0 = c_lo << 4;
            const double bp = s_bp[c_lo >> 8];
            double v[SCAN_CHUNK];
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK; ++j) {
                const int64_t sj = s0 + j;
                v[j] = lp[sj < N ? sj : N - 1];
            }
            const double v_prev = lp[s0 > 0 ? s0 - 1 : 0];  // last slot of the previous chunk (its own block prefix)
            const double bp_prev = s_bp[(s0 > 0 ? s0 - 1 : 0) >> 12];
            int pos = 0;
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK; ++j) pos += (s0 + j < N && left(bp + v[j])) ? 1 : 0;
            int64_t l2 = s0 + pos;
            // exact fix-up: the predicate on cdf_i = (BP + lp_i) / total is monotone in i over the whole array; the two
            // neighbours of the boundary are normally inside the chunk just fetched
            double vm = 0.0, vp = 0.0;
#pragma unroll
            for (int j = 0; j < SCAN_CHUNK; ++j) { vm = (j == pos - 1) ? v[j] : vm; vp = (j == pos) ? v[j] : vp; }
            bool walk = false;
            if (pos > 0) walk |= !left_exact((l2 - 1 == N - 1) ? 1.0 : (bp + vm) / total);
            else if (l2 > 0) walk |= !left_exact((bp_prev + v_prev) / total);
            if (pos < SCAN_CHUNK && l2 < N) walk |= left_exact((l2 == N - 1) ? 1.0 : (bp + vp) / total);
            else walk = true;
            TB2_CLK(5)
            if (walk) {
                if (l2 >= N) l2 = N - 1;
                while (l2 > 0) {
                    if (left_exact(cdf_at(lp, s_bp, total, l2 - 1, N))) break;
                    --l2;
                }
                while (l2 < N - 1) {
                    if (!left_exact(cdf_at(lp, s_bp, total, l2, N))) break;
                    ++l2;
                }
            }
            src = l2;
        }
        TB2_CLK(6)
        a.ridx[i] = (int32_t)src;
        const float4* ps = reinterpret_cast<const float4*>(a.poses_prop + src * 16);
        const float4 r0 = ps[0], r1 = ps[1], r2 = ps[2], r3 = ps[3];
        const double e_s = esrc[src];
        const uint8_t ok_s = a.valid[src];
        const int32_t nn_s = a.nn_idx[src];
        float4* pd = reinterpret_cast<float4*>(a.poses_out + i * 16);
        pd[0] = r0; pd[1] = r1; pd[2] = r2; pd[3] = r3;
        a.weights_out[i] = (e_s / S) * (ok_s ? 1.0 : 0.0);
        a.hint_out[i] = nn_s;
        TB2_CLK(7)
        TB2_WALL(1)
    }
    if (a.part_rmse && blockIdx.x == 0) {
        __shared__ double sa[4], sb[4];
        double p = 0.0, q = 0.0;
        for (int k = t; k < a.nrm; k += 256) { p += a.part_rmse[2 * k]; q += a.part_rmse[2 * k + 1]; }
        p = wsum(p);
        q = wsum(q);
        if ((t & 63) == 0) { sa[t >> 6] = p; sb[t >> 6] = q; }
        __syncthreads();
        if (t == 0) {
            p = (sa[0] + sa[1]) + (sa[2] + sa[3]);
            q = (sb[0] + sb[1]) + (sb[2] + sb[3]);
            a.rmse_out[0] = __builtin_sqrt(p / (double)N);
            a.rmse_out[1] = __builtin_sqrt(q / (double)N);
        }
    }
}

// T4 (sharded path): resample local slot i (global slot slot_base + i) from the GLOBAL cdf and gather pose /
//     weight / hint rows of any shard; identity when the weights are unusable.  The shards' data arrive as ONE
//     all_gather of a packed per-rank record block:
//        [ cdf: n x f64 | weights: n x f64 | poses: n x 16 f32 | nn_idx: n x i32 ]   (n = particles per rank)
//     so global particle p lives in block p / n at row p % n.
struct PackView {
    const char* base;
    int64_t stride;  // bytes between rank blocks
    int64_t n;       // particles per rank
    MD const char* blk(int64_t p, int64_t& row) const {
        const int64_t r = p / n;
        row = p - r * n;
        return base + r * stride;
    }
    MD double cdf(int64_t p) const { int64_t row; const char* b = blk(p, row); return reinterpret_cast<const double*>(b)[row]; }
    MD double weight(int64_t p) const { int64_t row; const char* b = blk(p, row); return reinterpret_cast<const double*>(b + 8 * n)[row]; }
    MD const float4* pose(int64_t p) const { int64_t row; const char* b = blk(p, row); return reinterpret_cast<const float4*>(b + 16 * n) + 4 * row; }
    MD int32_t nn(int64_t p) const { int64_t row; const char* b = blk(p, row); return reinterpret_cast<const int32_t*>(b + 80 * n)[row]; }
};

struct TailResampleArgs {
    int64_t N;            // local slots
    int64_t N_all;        // global particles
    int64_t slot_base;
    PackView pk;
    const int32_t* status;
    int32_t mode;
    const double* u;      // local uniforms or null
    float u32;
    uint64_t seed, step;
    int32_t* ridx;        // local out: global source index
    float* poses_out;
    double* weights_out;
    int32_t* hint_out;
};

__global__ __launch_bounds__(256) void k_tail_resample(TailResampleArgs a) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= a.N) return;
    const int64_t slot = a.slot_base + i, N = a.N_all;
    int64_t src = slot;
    if (a.status[0] == 0) {
        double t;
        bool upper;
        if (a.mode == MIDAS_RESAMPLE_MULTINOMIAL) {
            t = a.u ? a.u[i] : philox_uniform53((uint64_t)slot, a.seed, a.step);
            upper = false;
        } else {
            const float r = a.u32 >= 0.0f ? a.u32 : philox_uniform24(a.seed, a.step);
            const float off = r / (float)N;
            t = (double)slot / (double)N + (double)off;
            t = t >= 1.0 ? t - 1.0 : t;
            upper = true;
        }
        int64_t lo = 0, hi = N;
        while (hi > lo) {
            const int64_t mid = lo + ((hi - lo) >> 1);
            const double c = a.pk.cdf(mid);
            if (upper ? (c <= t) : (c < t)) lo = mid + 1; else hi = mid;
        }
        src = lo < N ? lo : N - 1;
    }
    a.ridx[i] = (int32_t)src;
    const float4* ps = a.pk.pose(src);
    float4* pd = reinterpret_cast<float4*>(a.poses_out + i * 16);
    float4 r0 = ps[0], r1 = ps[1], r2 = ps[2], r3 = ps[3];
    pd[0] = r0; pd[1] = r1; pd[2] = r2; pd[3] = r3;
    a.weights_out[i] = a.pk.weight(src);
    a.hint_out[i] = a.pk.nn(src);
}

// ------------------------------------------------------------------------------------------------
// sharded step, owner-side resample: rows travel instead of whole shards
// ------------------------------------------------------------------------------------------------
// After the exchange of the per-block records (r1_all) every rank knows the GLOBAL block prefix, and the draw of
// a slot is a pure function of the slot (Philox keyed by the global slot, the replicated host uniforms, or the
// systematic comb) - so every rank can tell, for every slot of the whole filter, which rank OWNS the source particle
// (the block the draw falls in, on the exact predicate).  The owner then resolves the exact source inside its own
// tables and sends that one row (pose, weight, NN index) to the rank holding the slot: an all_to_all of N rows per
// rank in total, where gathering every shard's packed block moved G-1 times as much.
//   pass COUNT: for every global slot: owner o, destination d  ->  counts[d] += (o == me), counts[G + o] += (d == me)
//               (the split sizes of the all_to_all; the caller reads them back) ; status / rmse finalised
//   pass PACK : the slots this rank owns: source slot (search_in_block), record -> send buffer, segment d at the
//               exclusive prefix of the send counts ; the rank's own masked weights
// Record (88 bytes): int32 slot (local at the destination) | int32 global source | int32 NN index | pad | f64 weight |
// 16 x f32 pose.
constexpr int ROUTE_REC = 88;
struct ShardRouteArgs {
    int64_t N;        // particles per rank
    int G, rank, nb;  // ranks, this rank, blocks per rank
    const double* r1_all;
    const double *e, *x_raw, *lp, *lp_raw, *gend, *gend_raw, *ggend, *ggend_raw;
    const uint8_t* valid;
    const int32_t* nn_idx;
    const float* poses_prop;
    int32_t* status;
    double* rmse_out;
    double n_total;
    int32_t softmax, mode;
    const double* u_all;
    float u32;
    uint64_t seed, step;
    int32_t* counts;  // [2 G]
    int32_t* cursor;  // [G]
    char* send;
    double* weights;
    // fixed-capacity form (no count pass, no read-back): segment of destination d = rows [d * fixed_cap, + fixed_cap) of
    // `send`, unused rows keep slot -1 (the buffers were set to 0xFF); rows that do not fit go to `ovf` (gathered by all)
    int64_t fixed_cap = 0, ovf_cap = 0;
    char* ovf = nullptr;
    int32_t* ovf_count = nullptr;
    char* self_rows = nullptr;  // [N] the rows this rank owns AND needs (they never travel)
    // peer-mapped form: row `slot` of the destination's inbox, written in place (fine-grained memory, system-scope stores)
    char* const* peers = nullptr;
    // ... with the completion protocol inside this kernel (the C-side frame across processes): the LAST workgroup to finish
    // publishes frame `tag` in slot `rank` of every inbox's flag block and then waits (bounded) until the own inbox carries
    // every rank's tag - when the kernel ends this rank's inbox holds all N rows, and ONE wave polled for it.  (Polling from
    // every workgroup of the consumer was measured: 1563 waves reading one address until it changes cost 26 us per frame.)
    const char* own_inbox = nullptr;
    long long flag_off = 0;
    unsigned long long tag = 0;
    unsigned* done = nullptr;  // workgroups finished, zero between launches (reset by the last one)
    const guide_t *guide = nullptr, *guide_raw = nullptr;  // the shard's guide tables (GUIDE_BINS, midas_internal.hpp) or null
};

// (sys_store8 / sys_load8 / pack2: peer_row.hpp - the rows of the peer-mapped form are 128-byte lines written by sixteen lanes)
template <bool PACK>
__global__ __launch_bounds__(256) void k_shard_route(ShardRouteArgs a) {
    __shared__ double s_bp[TB_MAX_BLOCKS];
    __shared__ double s_end[TB_MAX_BLOCKS];
    __shared__ double s_se[TB_MAX_BLOCKS];
    __shared__ double s_ex[12];
    __shared__ double s_tot[3];
    __shared__ int s_apply, s_cnt[2], s_base[2];
    __shared__ int s_hist[128], s_soff[64];
    const int t = threadIdx.x;
    const int nb_all = a.G * a.nb, rec = 5 * a.nb + 4;
    auto field = [&](int f, int i) { return a.r1_all[(int64_t)(i / a.nb) * rec + (int64_t)f * a.nb + (i % a.nb)]; };
    // ---- tables (as k_tail_fin): guard, sequential prefix over all blocks, exact cdf at the block ends
    double mx = -INFINITY, mn = INFINITY;
    bool nan = false;
    for (int i = t; i < nb_all; i += 256) {
        const double u = field(3, i), v = field(4, i);
        nan |= (u != u) || (v != v);
        mx = u > mx ? u : mx;
        mn = v < mn ? v : mn;
        s_end[i] = field(1, i);
        s_se[i] = field(0, i);
    }
    mx = wmax(mx);
    mn = wmin(mn);
    const bool wn = __any(nan);
    if ((t & 63) == 0) { s_ex[t >> 6] = mx; s_ex[4 + (t >> 6)] = mn; s_ex[8 + (t >> 6)] = wn ? 1.0 : 0.0; }
    if (t < 128) s_hist[t] = 0;
    if (t < 2) s_cnt[t] = 0;
    __syncthreads();
    if (t == 0) {
        mx = s_ex[0]; mn = s_ex[4];
        double f = s_ex[8];
        for (int w = 1; w < 4; ++w) { mx = s_ex[w] > mx ? s_ex[w] : mx; mn = s_ex[4 + w] < mn ? s_ex[4 + w] : mn; f += s_ex[8 + w]; }
        if (f != 0.0) { mx = NAN; mn = NAN; }
        s_apply = (a.softmax && !(__builtin_fabs(mx - mn) <= ISCLOSE_ATOL)) ? 1 : 0;
    }
    __syncthreads();
    const bool apply = s_apply != 0;
    if (!apply) {
        for (int i = t; i < nb_all; i += 256) s_end[i] = field(2, i);
        __syncthreads();
    }
    if (t == 0) {
        double acc = 0.0, S = 0.0;
        for (int i = 0; i < nb_all; ++i) { s_bp[i] = acc; acc = acc + s_end[i]; S = S + s_se[i]; }
        s_tot[0] = acc; s_tot[1] = apply ? S : 1.0;
        double nans = 0.0;
        for (int r = 0; r < a.G; ++r) nans += a.r1_all[(int64_t)r * rec + 5 * a.nb];
        s_tot[2] = nans;
        if (PACK) {  // segment offsets of the send buffer = exclusive prefix of the send counts (or fixed segments)
            int o = 0;
            for (int g = 0; g < a.G; ++g) { s_soff[g] = a.fixed_cap ? (int)(g * a.fixed_cap) : o; o += a.fixed_cap ? 0 : a.counts[g]; }
        }
    }
    __syncthreads();
    const double total = s_tot[0], S = s_tot[1];
    {   // exact cdf at the last slot of every block: (BP_b + W_b) / total (the block total is the block-local prefix at
        // the block's last slot); the globally last block ends at the last particle, forced to 1
        double wv[TB_MAX_BLOCKS / 256];
#pragma unroll
        for (int k = 0; k < TB_MAX_BLOCKS / 256; ++k) wv[k] = (k * 256 + t < nb_all) ? s_end[k * 256 + t] : 0.0;
        __syncthreads();
#pragma unroll
        for (int k = 0; k < TB_MAX_BLOCKS / 256; ++k) {
            const int b = k * 256 + t;
            if (b < nb_all) {
                s_end[b] = (b == nb_all - 1) ? 1.0 : (s_bp[b] + wv[k]) / total;
                s_se[b] = wv[k];  // (the sums of e are summed up: the array now holds the block totals, the guide tables' bin width)
            }
        }
        __syncthreads();
    }
    const bool bad_total = !(total == total) || total == 0.0;
    const bool usable = s_tot[2] == 0.0 && !bad_total;
    if ((!PACK || a.fixed_cap || a.peers) && blockIdx.x == 0 && t == 0) {
        double kept = 0.0, st2 = 0.0, sr2 = 0.0;
        for (int r = 0; r < a.G; ++r) {
            const double* fl = a.r1_all + (int64_t)r * rec + 5 * a.nb;
            kept += fl[1]; st2 += fl[2]; sr2 += fl[3];
        }
        int st = s_tot[2] != 0.0 ? 2 : 0;
        if (total != total) st |= 2;
        else if (total == 0.0) st |= 1;
        if (a.own_inbox) atomicOr(&a.status[0], st);  // (zeroed by the front; the last workgroup may add the "flag late" bit)
        else a.status[0] = st;
        a.status[1] = (int32_t)kept;
        if (a.rmse_out) { a.rmse_out[0] = __builtin_sqrt(st2 / a.n_total); a.rmse_out[1] = __builtin_sqrt(sr2 / a.n_total); }
    }
    // ---- per global slot
    const int64_t N = a.N, N_all = (int64_t)a.G * N;
    const int64_t i = (int64_t)blockIdx.x * 256 + t;
    const bool live = i < N_all;
    const int d = live ? (int)(i / N) : 0;
    int o = d, b = 0;
    double tq = 0.0;
    bool upper = false, past = false;
    if (live && usable) {
        if (a.mode == MIDAS_RESAMPLE_MULTINOMIAL) {
            tq = a.u_all ? a.u_all[i] : philox_uniform53((uint64_t)i, a.seed, a.step);
        } else {
            const float r = a.u32 >= 0.0f ? a.u32 : philox_uniform24(a.seed, a.step);
            const float off = r / (float)N_all;
            tq = (double)i / (double)N_all + (double)off;
            tq = tq >= 1.0 ? tq - 1.0 : tq;
            upper = true;
        }
        int lo = 0, hi = nb_all;
        while (hi > lo) {
            const int mid = lo + ((hi - lo) >> 1);
            const double c = s_end[mid];
            if (upper ? (c <= tq) : (c < tq)) lo = mid + 1; else hi = mid;
        }
        past = lo >= nb_all;  // beyond every block end: the last particle
        b = past ? nb_all - 1 : lo;
        o = b / a.nb;
    }
    if (!PACK) {
        if (live) {
            if (o == a.rank) atomicAdd(&s_hist[d], 1);
            if (d == a.rank) atomicAdd(&s_hist[64 + o], 1);
        }
        __syncthreads();
        if (t < a.G) {
            if (s_hist[t]) atomicAdd(&a.counts[t], s_hist[t]);
            if (s_hist[64 + t]) atomicAdd(&a.counts[a.G + t], s_hist[64 + t]);
        }
        return;
    }
    // ---- PACK: this rank's own masked weights, then the rows it owns
    const double* __restrict__ ev = apply ? a.e : a.x_raw;
    if (live && d == a.rank) {
        const int64_t il = i - (int64_t)a.rank * N;
        a.weights[il] = (ev[il] / S) * (a.valid[il] ? 1.0 : 0.0);
    }
    const bool mine = live && o == a.rank;
    if (a.peers) {
        // straight into the slot's row of the destination's inbox: the rows of a wave are staged in LDS and go out sixteen
        // lanes per row - whole 128-byte lines (peer_row.hpp)
        __shared__ unsigned long long s_stage[4][64][PEER_PIECES];
        __shared__ char* s_dst[4][64];
        const int w = t >> 6, lane = t & 63;
        const unsigned long long mm = __ballot(mine);
        if (mine) {
            int64_t src;
            if (!usable) src = i - (int64_t)a.rank * N;  // the resampler keeps the particles
            else if (past) src = N - 1;
            else
                src = search_in_block_t<const double*, const double*>(apply ? a.lp : a.lp_raw, apply ? a.gend : a.gend_raw, apply ? a.ggend : a.ggend_raw,
                                                                      b - a.rank * a.nb, N, a.rank == a.G - 1 ? N - 1 : -1, s_bp[b], total, tq, upper,
                                                                      (const double*)nullptr, apply ? a.guide : a.guide_raw, s_se[b]);
            const float4* ps = reinterpret_cast<const float4*>(a.poses_prop + src * 16);
            const float4 r0 = ps[0], r1 = ps[1], r2 = ps[2], r3 = ps[3];
            const double wgt = (ev[src] / S) * (a.valid[src] ? 1.0 : 0.0);
            const int k = __popcll(mm & ((1ull << lane) - 1ull));
            unsigned long long* st = s_stage[w][k];
            st[0] = pack2((int)(i - (int64_t)d * N), (int)((int64_t)a.rank * N + src));
            st[1] = pack2(a.nn_idx[src], d);
            st[2] = (unsigned long long)__double_as_longlong(wgt);
            st[3] = 0ull;
            st[4] = pack2f(r0.x, r0.y); st[5] = pack2f(r0.z, r0.w);
            st[6] = pack2f(r1.x, r1.y); st[7] = pack2f(r1.z, r1.w);
            st[8] = pack2f(r2.x, r2.y); st[9] = pack2f(r2.z, r2.w);
            st[10] = pack2f(r3.x, r3.y); st[11] = pack2f(r3.z, r3.w);
            s_dst[w][k] = a.peers[d] + (size_t)(i - (int64_t)d * N) * PEER_ROW;
        }
        __syncthreads();  // (uniform branch: every thread of the workgroup is here)
        peer_rows_store(s_stage[w], s_dst[w], __popcll(mm));
        if (a.own_inbox) {
            // The barrier waits for every wave's outstanding stores (s_waitcnt vmcnt(0) in front of s_barrier), and the row
            // stores are system-scope write-through stores: acknowledged = performed at the destination.  (That is a property of
            // gfx942 / gfx950's memory system, not of the programming model - which would want a release on every workgroup's
            // add: the build refuses any other target, below.)  So the count needs
            // no release of its own (a release fence here is a write-back of the XCD's whole L2 per workgroup: measured,
            // +7 us per launch); the workgroup that sees the full count publishes with a system-scope release store.
            __syncthreads();
            __shared__ int s_last;
            if (t == 0) s_last = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == gridDim.x - 1 ? 1 : 0;
            __syncthreads();
            if (s_last && w == 0) {
                if (lane == 0) __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                PeerInboxSrc f;
                f.rows = a.own_inbox; f.flag_off = a.flag_off; f.tag = a.tag; f.G = a.G; f.rank = a.rank; f.peers = a.peers; f.status = a.status;
                peer_flags_publish_wait(f, true);
            }
        }
        return;
    }
    const int d0 = (int)(((int64_t)blockIdx.x * 256) / N);  // a workgroup spans at most two destinations (N >= 256)
    int pos = 0;
    {
        if (mine) pos = atomicAdd(&s_cnt[d - d0], 1);
        __syncthreads();
        if (t < 2 && s_cnt[t]) s_base[t] = atomicAdd(&a.cursor[d0 + t], s_cnt[t]);
        __syncthreads();
    }
    if (mine) {
        int64_t src;
        if (!usable) src = i - (int64_t)a.rank * N;  // the resampler keeps the particles
        else if (past) src = N - 1;
        else
            src = search_in_block(apply ? a.lp : a.lp_raw, apply ? a.gend : a.gend_raw, apply ? a.ggend : a.ggend_raw,
                                  b - a.rank * a.nb, N, a.rank == a.G - 1 ? N - 1 : -1, s_bp[b], total, tq, upper);
        char* rp = a.send + (size_t)(s_soff[d] + s_base[d - d0] + pos) * ROUTE_REC;
        if (a.fixed_cap && d == a.rank) {  // own slot, own source: stays here (systematic draws are mostly of this kind)
            rp = a.self_rows + (size_t)(s_base[d - d0] + pos) * ROUTE_REC;
        } else if (a.fixed_cap && s_base[d - d0] + pos >= a.fixed_cap) {  // segment full: the row travels in the overflow block
            const int q = atomicAdd(a.ovf_count, 1);
            if (q >= a.ovf_cap) rp = nullptr;  // lost: the caller sees ovf_count > ovf_cap (counts_dev[3 G])
            else rp = a.ovf + (size_t)q * ROUTE_REC;
        }
        if (rp) {
        const float4* ps = reinterpret_cast<const float4*>(a.poses_prop + src * 16);
        const float4 r0 = ps[0], r1 = ps[1], r2 = ps[2], r3 = ps[3];
        const double w = (ev[src] / S) * (a.valid[src] ? 1.0 : 0.0);
        const int32_t nn = a.nn_idx[src];
        // records are 8-byte aligned (88 = 8 x 11): everything goes out as 8-byte pieces
        reinterpret_cast<int2*>(rp)[0] = make_int2((int)(i - (int64_t)d * N), (int)((int64_t)a.rank * N + src));
        reinterpret_cast<int2*>(rp)[1] = make_int2(nn, d);  // d: the destination rank (read from overflow rows)
        *reinterpret_cast<double*>(rp + 16) = w;
        float2* p2 = reinterpret_cast<float2*>(rp + 24);
        p2[0] = make_float2(r0.x, r0.y); p2[1] = make_float2(r0.z, r0.w);
        p2[2] = make_float2(r1.x, r1.y); p2[3] = make_float2(r1.z, r1.w);
        p2[4] = make_float2(r2.x, r2.y); p2[5] = make_float2(r2.z, r2.w);
        p2[6] = make_float2(r3.x, r3.y); p2[7] = make_float2(r3.z, r3.w);
        }
    }
}

// rows: records to look at; dest >= 0: only rows addressed to that rank (overflow block), rows with slot -1 are padding
__global__ __launch_bounds__(256) void k_shard_unpack(int64_t N, const char* __restrict__ recv, int32_t* __restrict__ ridx,
                                                      float* __restrict__ poses_out, double* __restrict__ weights_out,
                                                      int32_t* __restrict__ hint_out, int32_t dest) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    const char* rp = recv + (size_t)r * ROUTE_REC;
    const int2 h0 = *reinterpret_cast<const int2*>(rp), h1 = *reinterpret_cast<const int2*>(rp + 8);
    if (h0.x < 0 || (dest >= 0 && h1.y != dest)) return;
    const double w = *reinterpret_cast<const double*>(rp + 16);
    const float2* p2 = reinterpret_cast<const float2*>(rp + 24);
    float2 v[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) v[k] = p2[k];
    const int64_t slot = h0.x;
    ridx[slot] = h0.y;
    hint_out[slot] = h1.x;
    weights_out[slot] = w;
    float2* pd = reinterpret_cast<float2*>(poses_out + slot * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) pd[k] = v[k];
}

// the N rows other ranks stored into this rank's inbox (row r = slot r)
__device__ __forceinline__ void unpack_peer_row(const char* __restrict__ inbox, int64_t r, int32_t* __restrict__ ridx, float* __restrict__ poses_out,
                                                double* __restrict__ weights_out, int32_t* __restrict__ hint_out) {
    const PeerRow v = peer_row_load(inbox, r);
    ridx[r] = (int32_t)(v.head[0] >> 32);
    hint_out[r] = (int32_t)(v.head[1] & 0xFFFFFFFFull);
    weights_out[r] = __longlong_as_double((long long)v.head[2]);
    unsigned long long* pd = reinterpret_cast<unsigned long long*>(poses_out + r * 16);
#pragma unroll
    for (int k = 0; k < 8; ++k) pd[k] = v.pose[k];
}

__global__ __launch_bounds__(256) void k_shard_unpack_peer(int64_t N, const char* __restrict__ inbox, int32_t* __restrict__ ridx,
                                                           float* __restrict__ poses_out, double* __restrict__ weights_out,
                                                           int32_t* __restrict__ hint_out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    unpack_peer_row(inbox, r, ridx, poses_out, weights_out, hint_out);
}

// Device-side completion flags of the peer-mapped exchange (the C-side sharded frame, midas_shard_step): behind its route
// kernel - a kernel boundary, so every row it stored is out - a rank stores the frame's tag into slot `rank` of the flag
// block that follows the N rows of every inbox; the unpack kernel of a rank waits until all G slots of its OWN inbox carry the
// tag.  That replaces the 4-byte collective the host-driven frame used as a barrier.  No cycle: a rank's flag kernel sits
// behind its record all_gather, which completes only when every rank has joined it - and every rank enqueues its join before
// its own waiting kernel.  The wait is bounded (2 s of the 100 MHz wall clock): on expiry status[0] gets bit 16 and the
// kernel goes on - a stuck peer must not hang the device.
__global__ void k_peer_flag_write(char* const* peers, int G, int rank, long long flag_off, unsigned long long tag) {
    const int d = threadIdx.x;
    if (d < G) __hip_atomic_store(reinterpret_cast<unsigned long long*>(peers[d] + flag_off) + rank, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

__global__ __launch_bounds__(256) void k_shard_unpack_peer_wait(int64_t N, const char* __restrict__ inbox, int32_t* __restrict__ ridx,
                                                                float* __restrict__ poses_out, double* __restrict__ weights_out,
                                                                int32_t* __restrict__ hint_out, int G, long long flag_off,
                                                                unsigned long long tag, int32_t* __restrict__ status,
                                                                char* const* __restrict__ peers, int rank) {
    // peers != NULL: this kernel also PUBLISHES the rank's completion flag (its first workgroup, before anybody waits): it was
    // launched behind the route kernel, so every row this rank stored is out - one launch fewer than a flag kernel of its own
    if (peers && blockIdx.x == 0 && (int)threadIdx.x < G)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(peers[threadIdx.x] + flag_off) + rank, tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if ((int)threadIdx.x < G) {
        const unsigned long long* f = reinterpret_cast<const unsigned long long*>(inbox + flag_off) + threadIdx.x;
        const long long t0 = wall_clock64();
        bool late = false;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < tag) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000ll) { late = true; break; }
        }
        if (late && status) atomicOr(&status[0], 16);
    }
    __syncthreads();
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= N) return;
    unpack_peer_row(inbox, r, ridx, poses_out, weights_out, hint_out);
}

// start-up self test of the peer data path (include/midas_hip.h)
__global__ void k_peer_probe_write(char* const* peers, int G, int rank, int nonce) {
    const int d = threadIdx.x;
    if (d < G) sys_store8(peers[d] + (size_t)rank * ROUTE_REC, pack2(nonce, rank));
}
__global__ void k_peer_probe_check(const char* inbox, int G, int nonce, int32_t* ok) {
    const int r = threadIdx.x;
    const bool good = r >= G || sys_load8(inbox + (size_t)r * ROUTE_REC) == pack2(nonce, r);
    const bool all = __all(good);
    if (r == 0) ok[0] = all ? 1 : 0;
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
#define LAUNCH_CHECK(ctx) MIDAS_HIP_CHECK(ctx, hipGetLastError())

// The grouped form needs the hand-over records, the whole grid resident and whole 16-byte pieces of the per-slot arrays.
// Resident = the launch's workgroups (four-wave group workgroups + the rmse workgroup + `extra` list workgroups) fit the DEVICE's
// compute units at the kernel's occupancy - queried, not assumed (a partitioned or smaller part takes the one-workgroup-per-block
// form earlier); the waves wait for each other, so a grid that cannot all start must not take this form.
static int tail_resident_workgroups(midas_ctx* ctx) {
    static int cap_dev[64] = {};
    const int di = ctx->device >= 0 && ctx->device < 64 ? ctx->device : 0;
    if (!cap_dev[di] || ctx->device != di) {
        hipDeviceProp_t prop;
        const int ncu = (hipGetDeviceProperties(&prop, ctx->device) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 0;
        int occ = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, (const void*)k_tail_a3, 256, 0) != hipSuccess) { occ = 0; (void)hipGetLastError(); }
        cap_dev[di] = ncu * occ > 0 ? ncu * occ : -1;
    }
    return cap_dev[di];
}
static bool tail_grouped_ok(midas_ctx* ctx, int64_t N, const int32_t* nn_idx, const uint8_t* valid, const TailTables& tb, int extra) {
    const char* env = getenv("MIDAS_TAIL_GROUPED");  // (read per launch: the tests compare both forms in one process)
    const bool on = !(env && env[0] == '0');
    if (!on || !ctx->tail_rec || N < SCAN_CHUNK || ceil_div(N, SCAN_BLOCK) > ctx->tail_rec_blocks) return false;
    if ((ceil_div(N, TG_GROUP) + 3) / 4 + 1 + extra > tail_resident_workgroups(ctx)) return false;
    const uintptr_t al = (uintptr_t)nn_idx | (uintptr_t)tb.e | (uintptr_t)tb.x_raw | (uintptr_t)tb.lp | (uintptr_t)tb.lp_raw;
    return (al & 15) == 0 && ((uintptr_t)valid & 3) == 0;
}
static int launch_tail_a3(midas_ctx* ctx, int64_t N, const double* scores, const int32_t* nn_idx, const uint8_t* valid, int32_t softmax,
                          const TailTables& tb, bool padded, int32_t* status, double* flags_out, const double* part_rmse, double* rmse_out,
                          bool rmse_raw, const ScorePredict* predict) {
    TailGroupArgs a;
    a.N = N; a.scores = scores; a.nn_idx = nn_idx; a.valid = valid; a.softmax = softmax; a.tb = tb; a.padded = padded;
    a.status = status; a.flags_out = flags_out; a.rec = ctx->tail_rec;
    if (++ctx->tail_tag == 0) ctx->tail_tag = 1;
    a.tag = ctx->tail_tag;
    const int ngroups = (int)ceil_div(N, TG_GROUP), nwg = (ngroups + 3) / 4;
    const int nscan = predict ? (int)ceil_div(predict->K, 256 * PREDICT_PER_THREAD) : 0;
    hipLaunchKernelGGL(k_tail_a3, dim3((unsigned)(nwg + (rmse_out ? 1 : 0) + nscan)), dim3(256), 0, ctx->stream, a, ngroups, nwg, part_rmse,
                       particle_update_blocks(N), rmse_out, rmse_raw, predict ? *predict : ScorePredict());
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_shard_unpack_peer(midas_ctx* ctx, int64_t N, const void* inbox, int32_t* ridx, float* poses_out, double* weights_out,
                             int32_t* hint_out) {
    hipLaunchKernelGGL(k_shard_unpack_peer, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, (const char*)inbox, ridx,
                       poses_out, weights_out, hint_out);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_peer_flag_write(midas_ctx* ctx, void* const* peers, int G, int rank, int64_t flag_off, uint64_t tag) {
    hipLaunchKernelGGL(k_peer_flag_write, dim3(1), dim3(64), 0, ctx->stream, (char* const*)peers, G, rank, (long long)flag_off, (unsigned long long)tag);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_shard_unpack_peer_wait(midas_ctx* ctx, int64_t N, const void* inbox, int32_t* ridx, float* poses_out, double* weights_out,
                                  int32_t* hint_out, int G, int64_t flag_off, uint64_t tag, int32_t* status, void* const* peers, int rank) {
    hipLaunchKernelGGL(k_shard_unpack_peer_wait, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, (const char*)inbox, ridx,
                       poses_out, weights_out, hint_out, G, (long long)flag_off, (unsigned long long)tag, status, (char* const*)peers, rank);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_peer_probe(midas_ctx* ctx, void* const* peers, const void* inbox, int G, int rank, int nonce, int32_t* ok) {
    if (peers) hipLaunchKernelGGL(k_peer_probe_write, dim3(1), dim3(64), 0, ctx->stream, (char* const*)peers, G, rank, nonce);
    else hipLaunchKernelGGL(k_peer_probe_check, dim3(1), dim3(64), 0, ctx->stream, (const char*)inbox, G, nonce, ok);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_gather_f64(midas_ctx* ctx, int64_t N, const double* table, const int32_t* idx, double* out) {
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_gather_f64, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, table, idx, out);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_softmax(midas_ctx* ctx, int64_t N, const double* x, int32_t softmax, double* w) {
    if (N == 0) return MIDAS_OK;
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    void* sc;
    int rc = midas_scratch(ctx, (size_t)nb * 3 * sizeof(double) + 64, &sc);
    if (rc) return rc;
    double* pmax = (double*)sc;
    double* pmin = pmax + nb;
    double* psum = pmin + nb;
    int32_t* flag = (int32_t*)(psum + nb);
    hipLaunchKernelGGL(k_extrema, dim3(nb), dim3(256), 0, ctx->stream, N, x, pmax, pmin);
    hipLaunchKernelGGL(k_exp_partial, dim3(nb), dim3(256), 0, ctx->stream, N, x, nb, pmax, pmin, softmax, w, psum, flag);
    hipLaunchKernelGGL(k_normalise, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, w, nb, psum, flag);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_prune(midas_ctx* ctx, int64_t N, double* w, const double* dist, double thr, int32_t* nvalid) {
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(nvalid, 0, sizeof(int32_t), ctx->stream));
    if (N == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_prune, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, w, dist, thr, nvalid);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_cdf(midas_ctx* ctx, int64_t N, const double* w, double* cdf, int32_t* status) {
    MIDAS_HIP_CHECK(ctx, hipMemsetAsync(status, 0, sizeof(int32_t), ctx->stream));
    if (N == 0) return MIDAS_OK;
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    void* sc;
    int rc = midas_scratch(ctx, (size_t)nb * sizeof(double), &sc);
    if (rc) return rc;
    hipLaunchKernelGGL(k_scan_local, dim3(nb), dim3(256), 0, ctx->stream, N, w, cdf, (double*)sc, status);
    hipLaunchKernelGGL(k_cdf_final, dim3(nb), dim3(256), 0, ctx->stream, N, cdf, nb, (const double*)sc, status);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_search(midas_ctx* ctx, int64_t N, const double* cdf, int64_t M, int32_t mode, const double* u, float u32,
                  uint64_t seed, uint64_t step, int32_t* idx) {
    if (M == 0) return MIDAS_OK;
    hipLaunchKernelGGL(k_search, dim3((unsigned)ceil_div(M, 256)), dim3(256), 0, ctx->stream, N, cdf, M, mode, u, u32,
                       seed, step, idx);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_gather_rows(midas_ctx* ctx, int64_t M, const int32_t* idx, const void* src, void* dst, int32_t row_bytes) {
    if (M == 0) return MIDAS_OK;
    const bool a16 = row_bytes % 16 == 0 && (uintptr_t)src % 16 == 0 && (uintptr_t)dst % 16 == 0;
    const bool a8 = row_bytes % 8 == 0 && (uintptr_t)src % 8 == 0 && (uintptr_t)dst % 8 == 0;
    const bool a4 = row_bytes % 4 == 0 && (uintptr_t)src % 4 == 0 && (uintptr_t)dst % 4 == 0;
    if (a16) {
        const int per = row_bytes / 16;
        hipLaunchKernelGGL((k_gather_rows<float4>), dim3((unsigned)ceil_div(M * per, 256)), dim3(256), 0, ctx->stream, M,
                           idx, (const float4*)src, (float4*)dst, per);
    } else if (a8) {
        const int per = row_bytes / 8;
        hipLaunchKernelGGL((k_gather_rows<double>), dim3((unsigned)ceil_div(M * per, 256)), dim3(256), 0, ctx->stream, M,
                           idx, (const double*)src, (double*)dst, per);
    } else if (a4) {
        const int per = row_bytes / 4;
        hipLaunchKernelGGL((k_gather_rows<float>), dim3((unsigned)ceil_div(M * per, 256)), dim3(256), 0, ctx->stream, M,
                           idx, (const float*)src, (float*)dst, per);
    } else {
        hipLaunchKernelGGL((k_gather_rows<uint8_t>), dim3((unsigned)ceil_div(M * row_bytes, 256)), dim3(256), 0,
                           ctx->stream, M, idx, (const uint8_t*)src, (uint8_t*)dst, row_bytes);
    }
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_tail_a(midas_ctx* ctx, int64_t N, const double* x, const uint8_t* valid, int np, int pstride,
                  const double* pmax_all, const double* pmin_all, int32_t softmax, double* e_io, double* lp_out,
                  double* block_sums_e, double* block_totals_em, double* flags_out, int32_t* flag, int32_t* status,
                  int batch) {
    hipLaunchKernelGGL(k_tail_a, dim3((unsigned)ceil_div(N, SCAN_BLOCK), (unsigned)(batch > 1 ? batch : 1)), dim3(256), 0, ctx->stream, N, x, valid, np, pstride,
                       pmax_all, pmin_all, softmax, e_io, lp_out, block_sums_e, block_totals_em, flags_out, flag, status);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_tail_fin(midas_ctx* ctx, int64_t N, const double* e, const double* x_raw, const double* lp, const double* lp_raw,
                    const uint8_t* valid, double* weights, double* cdf_io, int G, int nb, const double* r1_all, int rank,
                    double n_total, int32_t softmax, double* rmse_out, int32_t* status) {
    if ((int64_t)G * nb > TB_MAX_BLOCKS)
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "G*nb", "more than 4 M particles in total in the sharded step");
    hipLaunchKernelGGL(k_tail_fin, dim3((unsigned)ceil_div(N, SCAN_BLOCK)), dim3(256), 0, ctx->stream, N, e, x_raw, lp, lp_raw, valid,
                       weights, cdf_io, G, nb, r1_all, rank, n_total, softmax, rmse_out, status);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

// TA2 of one shard: the exchange record r1 = [bsum_e | btot | btot_raw | bmax | bmin | NaN count, kept count | ...]
int launch_shard_tail_a(midas_ctx* ctx, int64_t N, const double* scores, const int32_t* nn_idx, const uint8_t* valid,
                        int32_t softmax, const TailTables& tb, double* r1, int32_t* status, const double* part_rmse,
                        const ScorePredict* predict) {
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    constexpr bool direct = true;  // (k_tail_a2, the LDS-staged form, serves N < 16 and the batch without table strides only)
    const bool with_list = predict && predict->stamps && predict->list;
    if ((part_rmse || with_list) && !(direct && N >= SCAN_CHUNK))
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "shard tail", "rmse sums / prediction list in the tail need the direct tail kernel (N >= 16)");
    if (direct && N >= SCAN_CHUNK) {  // the shard's per-slot tables are padded (shard_tables_of, api.hip)
        TailTables t = tb;
        t.bsum_e = r1; t.btot = r1 + nb; t.btot_raw = r1 + 2 * nb; t.bmax = r1 + 3 * nb; t.bmin = r1 + 4 * nb;
        if (tail_grouped_ok(ctx, N, nn_idx, valid, t, with_list ? (int)ceil_div(predict->K, 256 * PREDICT_PER_THREAD) : 0))
            return launch_tail_a3(ctx, N, scores, nn_idx, valid, softmax, t, true, status, r1 + 5 * nb, part_rmse,
                                  part_rmse ? r1 + 5 * nb + 2 : (double*)nullptr, true, with_list ? predict : nullptr);
        const int nscan = with_list ? (int)ceil_div(predict->K, 256 * PREDICT_PER_THREAD) : 0;
        // part_rmse: the front's per-wave sums are added up here (block 0) into r1[5 nb + 2 ..] instead of by a kernel of their own
        hipLaunchKernelGGL(k_tail_a2d, dim3((unsigned)(nb + nscan)), dim3(256), 0, ctx->stream, N, scores, nn_idx, valid, softmax, t, true,
                           status, r1 + 5 * nb, part_rmse, particle_update_blocks(N), part_rmse ? r1 + 5 * nb + 2 : (double*)nullptr,
                           (int64_t)0, (int64_t)0, nb, with_list ? *predict : ScorePredict(), true);
        LAUNCH_CHECK(ctx);
        return MIDAS_OK;
    }
    if (tb.guide)  // (this form of the tail writes no guide tables: "no guide" in every entry, see tail_block.hpp)
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(tb.guide, 0xFF, (size_t)2 * nb * GUIDE_STRIDE * sizeof(guide_t), ctx->stream));
    hipLaunchKernelGGL(k_tail_a2, dim3((unsigned)nb), dim3(256), 0, ctx->stream, N, scores, nn_idx, valid, softmax, tb.e, tb.x_raw,
                       tb.lp, tb.lp_raw, tb.gend, tb.gend_raw, tb.ggend, tb.ggend_raw, r1, r1 + nb, r1 + 2 * nb, r1 + 3 * nb, r1 + 4 * nb,
                       status, r1 + 5 * nb, 0);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_tail_resample(midas_ctx* ctx, const midas_tail_resample_args& r) {
    TailResampleArgs a;
    a.N = r.N; a.N_all = r.N_all; a.slot_base = r.slot_base;
    a.pk.base = (const char*)r.pack_all_dev; a.pk.stride = r.rank_stride; a.pk.n = r.n_per_rank;
    a.status = r.status_dev; a.mode = r.mode; a.u = r.u_dev; a.u32 = r.u32;
    a.seed = r.seed; a.step = r.step; a.ridx = r.ridx_dev;
    a.poses_out = r.poses_out_dev; a.weights_out = r.weights_out_dev; a.hint_out = r.hint_out_dev;
    hipLaunchKernelGGL(k_tail_resample, dim3((unsigned)ceil_div(r.N, 256)), dim3(256), 0, ctx->stream, a);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

#ifdef MIDAS_DEBUG_CLOCKS
int debug_tb2_clocks(long long* out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_tb2_clk), 16 * sizeof(long long)) == hipSuccess ? 0 : 1; }
int debug_ta_clocks(long long* out16) { return hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_ta_clk), 16 * sizeof(long long)) == hipSuccess ? 0 : 1; }
// stamps of the grouped tail: out64 = two waves' phases, out_w (4096) = start / end per group wave [2 G, 2 G + 1], rmse workgroup
// [2048, 2049], list workgroups [2050 + 2 b, ..]; reset: all zero
int debug_tg_clocks(long long* io64, int reset) {
    if (reset) {
        static long long zero[8192];
        (void)hipMemcpyToSymbol(HIP_SYMBOL(g_tg_w), zero, sizeof(zero));
        return hipMemcpyToSymbol(HIP_SYMBOL(g_tg_clk), zero, 64 * sizeof(long long)) == hipSuccess ? 0 : 1;
    }
    return hipMemcpyFromSymbol(io64, HIP_SYMBOL(g_tg_clk), 64 * sizeof(long long)) == hipSuccess ? 0 : 1;
}
int debug_tg_waves(long long* out8192) { return hipMemcpyFromSymbol(out8192, HIP_SYMBOL(g_tg_w), 8192 * sizeof(long long)) == hipSuccess ? 0 : 1; }
#endif

int launch_tail_a2(midas_ctx* ctx, int64_t N, const double* scores, const int32_t* nn_idx, const uint8_t* valid,
                   int32_t softmax, const TailTables& tb, int32_t* status, int batch, int64_t score_stride, bool padded_tables,
                   const double* part_rmse, double* rmse_out, int64_t tstride, const ScorePredict* predict) {
    const int nb = (int)ceil_div(N, SCAN_BLOCK);
    constexpr bool direct = true;  // (k_tail_a2, the LDS-staged form, serves N < 16 and the batch without table strides only)
    const bool with_list = predict && predict->stamps && predict->list && batch <= 1;
    if (with_list && !(direct && N >= SCAN_CHUNK)) return midas_set_error(ctx, MIDAS_ERR_INVALID, "score_list", "the prediction list needs the direct tail kernel (N >= 16)");
    if (direct && batch <= 1 && tail_grouped_ok(ctx, N, nn_idx, valid, tb, with_list ? (int)ceil_div(predict->K, 256 * PREDICT_PER_THREAD) : 0))
        return launch_tail_a3(ctx, N, scores, nn_idx, valid, softmax, tb, padded_tables, status, nullptr, part_rmse,
                              part_rmse ? rmse_out : (double*)nullptr, false, with_list ? predict : nullptr);
    if (direct && (batch <= 1 || tstride > 0) && N >= SCAN_CHUNK) {
        const int nscan = with_list ? (int)ceil_div(predict->K, 256 * PREDICT_PER_THREAD) : 0;
        hipLaunchKernelGGL(k_tail_a2d, dim3((unsigned)(nb + nscan), (unsigned)(batch > 1 ? batch : 1)), dim3(256), 0, ctx->stream, N, scores, nn_idx,
                           valid, softmax, tb, padded_tables, status, (double*)nullptr, part_rmse, particle_update_blocks(N),
                           part_rmse ? rmse_out : (double*)nullptr, score_stride, tstride, nb, with_list ? *predict : ScorePredict());
        LAUNCH_CHECK(ctx);
        return MIDAS_OK;
    }
    if (tb.guide && batch <= 1) {  // (this form of the tail writes no guide tables: "no guide" in every entry, see tail_block.hpp)
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(tb.guide, 0xFF, (size_t)nb * GUIDE_STRIDE * sizeof(guide_t), ctx->stream));
        if (tb.guide_raw) MIDAS_HIP_CHECK(ctx, hipMemsetAsync(tb.guide_raw, 0xFF, (size_t)nb * GUIDE_STRIDE * sizeof(guide_t), ctx->stream));
    }
    hipLaunchKernelGGL(k_tail_a2, dim3((unsigned)nb, (unsigned)(batch > 1 ? batch : 1)), dim3(256), 0, ctx->stream, N, scores, nn_idx, valid, softmax, tb.e, tb.x_raw,
                       tb.lp, tb.lp_raw, tb.gend, tb.gend_raw, tb.ggend, tb.ggend_raw, tb.bsum_e, tb.btot, tb.btot_raw, tb.bmax, tb.bmin,
                       status, nullptr, score_stride);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_tail_b2(midas_ctx* ctx, const StepTailArgs& a, const TailTables& tb) {
    const int nb = (int)ceil_div(a.N, SCAN_BLOCK), ng = (int)ceil_div(a.N, SCAN_CHUNK);
    if (nb > TB_MAX_BLOCKS) return midas_set_error(ctx, MIDAS_ERR_INVALID, "N", "more than 4 M particles per GPU: shard them");
    // every workgroup of TB2 loads the table: keep it whole (one entry per chunk) while that is cheap, coarser
    // for large N where (N / 256 workgroups) x table bytes would dominate
    static const int tab_env = getenv("MIDAS_TB2_TAB") ? atoi(getenv("MIDAS_TB2_TAB")) : 0;
    const int tab_cap = tab_env > 0 ? (tab_env < TB2_TAB ? tab_env : TB2_TAB) : (ng <= TB2_TAB ? TB2_TAB : (ng <= 4 * TB2_TAB ? 2048 : 1024));  // measured at N = 300k / 1M
    int cshift = 0;
    while (ceil_div((int64_t)ng, (int64_t)1 << cshift) > tab_cap) ++cshift;
    const int nt = (int)ceil_div((int64_t)ng, (int64_t)1 << cshift);
    TailB2Args b;
    b.N = a.N; b.nb = nb; b.ng = ng; b.nt = nt; b.cshift = cshift;
    b.e = tb.e; b.x_raw = tb.x_raw; b.lp = tb.lp; b.lp_raw = tb.lp_raw; b.gend = tb.gend; b.gend_raw = tb.gend_raw;
    b.bsum_e = tb.bsum_e; b.btot = tb.btot; b.btot_raw = tb.btot_raw; b.bmax = tb.bmax; b.bmin = tb.bmin;
    b.valid = a.valid; b.softmax = a.softmax; b.status = a.status; b.weights = a.weights; b.mode = a.mode;
    b.u = a.u; b.u32 = a.u32; b.seed = a.seed; b.step = a.step; b.ridx = a.ridx; b.poses_prop = a.poses_prop;
    b.poses_out = a.poses_out; b.weights_out = a.weights_out; b.nn_idx = a.nn_idx; b.hint_out = a.hint_out;
    b.part_rmse = a.part_rmse; b.nrm = a.part_rmse ? particle_update_blocks(a.N) : 0; b.rmse_out = a.rmse_out;
    b.tstride = a.tstride;
    hipLaunchKernelGGL(k_tail_b2, dim3((unsigned)ceil_div(a.N, 256), (unsigned)(a.batch > 1 ? a.batch : 1)), dim3(256),
                       (size_t)(nt + 3 * nb) * sizeof(double), ctx->stream, b);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_shard_route(midas_ctx* ctx, const midas_shard_route_args& r, const TailTables& tb, bool pack, const PeerRouteSync* sync) {
    const int nb = (int)ceil_div(r.N, SCAN_BLOCK);
    if ((int64_t)r.G * nb > TB_MAX_BLOCKS)
        return midas_set_error(ctx, MIDAS_ERR_INVALID, "G*nb", "more than 4 M particles in total in the sharded step");
    ShardRouteArgs a;
    a.N = r.N; a.G = r.G; a.rank = r.rank; a.nb = nb; a.r1_all = r.r1_all_dev;
    a.e = tb.e; a.x_raw = tb.x_raw; a.lp = tb.lp; a.lp_raw = tb.lp_raw; a.gend = tb.gend; a.gend_raw = tb.gend_raw;
    a.guide = tb.guide; a.guide_raw = tb.guide_raw;
    a.ggend = tb.ggend; a.ggend_raw = tb.ggend_raw;
    a.valid = r.valid_dev; a.nn_idx = r.nn_idx_dev; a.poses_prop = r.poses_prop_dev;
    a.status = r.status_dev; a.rmse_out = r.rmse_dev; a.n_total = (double)r.G * (double)r.N;
    a.softmax = r.softmax; a.mode = r.resample_mode; a.u_all = r.u_all_dev; a.u32 = r.u32; a.seed = r.seed; a.step = r.step;
    a.counts = r.counts_dev; a.cursor = r.counts_dev + 2 * r.G; a.send = (char*)r.send_dev; a.weights = r.weights_dev;
    const unsigned grid = (unsigned)ceil_div((int64_t)r.G * r.N, 256);
    if (pack && r.peers_dev) {  // rows stored straight into the destinations' inboxes
        a.peers = (char* const*)r.peers_dev;
        if (sync) { a.own_inbox = sync->inbox; a.flag_off = sync->flag_off; a.tag = sync->tag; a.done = reinterpret_cast<unsigned*>(const_cast<char*>(sync->inbox) + sync->flag_off + 64 * 8); }
        hipLaunchKernelGGL(k_shard_route<true>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (pack && r.fixed_cap > 0) {  // one pass, no counts: padded segments + overflow block
        a.fixed_cap = r.fixed_cap; a.ovf_cap = r.ovf_cap; a.ovf = (char*)r.ovf_dev; a.ovf_count = r.counts_dev + 2 * r.G + r.G;
        a.self_rows = (char*)r.self_dev;
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(r.self_dev, 0xFF, (size_t)r.N * ROUTE_REC, ctx->stream));
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(r.counts_dev, 0, (size_t)(3 * r.G + 1) * sizeof(int32_t), ctx->stream));
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(r.send_dev, 0xFF, (size_t)r.G * r.fixed_cap * ROUTE_REC, ctx->stream));
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(r.ovf_dev, 0xFF, (size_t)r.ovf_cap * ROUTE_REC, ctx->stream));
        hipLaunchKernelGGL(k_shard_route<true>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else if (pack) {
        hipLaunchKernelGGL(k_shard_route<true>, dim3(grid), dim3(256), 0, ctx->stream, a);
    } else {
        MIDAS_HIP_CHECK(ctx, hipMemsetAsync(r.counts_dev, 0, (size_t)3 * r.G * sizeof(int32_t), ctx->stream));
        hipLaunchKernelGGL(k_shard_route<false>, dim3(grid), dim3(256), 0, ctx->stream, a);
    }
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_shard_unpack(midas_ctx* ctx, int64_t N, const void* recv, int32_t* ridx, float* poses_out, double* weights_out,
                        int32_t* hint_out, int32_t dest) {
    hipLaunchKernelGGL(k_shard_unpack, dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, ctx->stream, N, (const char*)recv, ridx,
                       poses_out, weights_out, hint_out, dest);
    LAUNCH_CHECK(ctx);
    return MIDAS_OK;
}

int launch_step_tail(midas_ctx* ctx, const StepTailArgs& a, int prof_slot_base) {
    const int nb = (int)ceil_div(a.N, SCAN_BLOCK);
    if (nb > TB_MAX_BLOCKS) return midas_set_error(ctx, MIDAS_ERR_INVALID, "N", "more than 4 M particles per GPU: shard them");
    void* sc;
    const int B = a.batch > 1 ? a.batch : 1;
    int rc = midas_scratch(ctx, (size_t)B * nb * 2 * sizeof(double) + (size_t)B * sizeof(int32_t) + 64, &sc);
    if (rc) return rc;
    double* psum = (double*)sc;
    double* pw = psum + (size_t)B * nb;
    double* e = a.e;
    int32_t* flag = (int32_t*)(pw + (size_t)B * nb);
    if (!a.x) {  // deferred mode: the tail gathers the scores (B trajectories as grid.y, plain strides)
        const int ng = (int)ceil_div(a.N, SCAN_CHUNK);
        void* sc2;
        if ((rc = midas_scratch(ctx, (size_t)B * ((size_t)nb * 35 + (size_t)ng * 2) * sizeof(double), &sc2))) return rc;
        TailTables tb;
        tb.e = e; tb.x_raw = a.x_raw; tb.lp = a.cdf; tb.lp_raw = a.lp_raw;
        tb.bsum_e = psum; tb.btot = pw;
        tb.btot_raw = (double*)sc2; tb.bmax = tb.btot_raw + (size_t)B * nb; tb.bmin = tb.bmax + (size_t)B * nb;
        tb.gend = tb.bmin + (size_t)B * nb; tb.gend_raw = tb.gend + (size_t)B * ng;
        tb.ggend = tb.gend_raw + (size_t)B * ng; tb.ggend_raw = tb.ggend + (size_t)B * 16 * nb;
        if ((rc = launch_tail_a2(ctx, a.N, a.scores, a.nn_idx, a.valid, a.softmax, tb, a.status, B, a.score_stride))) return rc;
        prof_mark(ctx, prof_slot_base + 1);
        if ((rc = launch_tail_b2(ctx, a, tb))) return rc;
        prof_mark(ctx, prof_slot_base + 2);
        return MIDAS_OK;
    }
    if ((rc = launch_tail_a(ctx, a.N, a.x, a.valid, a.npart, 1, a.part_max, a.part_min, a.softmax, e, a.cdf, psum, pw, nullptr,
                            flag, a.status, B)))
        return rc;
    prof_mark(ctx, prof_slot_base + 1);
    TailBArgs b;
    b.N = a.N; b.nb = nb; b.e = e; b.valid = a.valid; b.lp = a.cdf; b.block_sums_e = psum; b.block_totals_em = pw;
    b.flag = flag; b.status = a.status; b.weights = a.weights; b.mode = a.mode; b.u = a.u; b.u32 = a.u32;
    b.seed = a.seed; b.step = a.step; b.ridx = a.ridx; b.poses_prop = a.poses_prop; b.poses_out = a.poses_out;
    b.weights_out = a.weights_out; b.nn_idx = a.nn_idx; b.hint_out = a.hint_out;
    b.part_rmse = a.part_rmse; b.nrm = a.part_rmse ? particle_update_blocks(a.N) : 0; b.rmse_out = a.rmse_out;
    b.slot_base = 0;
    hipLaunchKernelGGL(k_tail_b, dim3((unsigned)ceil_div(a.N, 256), (unsigned)B), dim3(256), 0, ctx->stream, b);
    LAUNCH_CHECK(ctx);
    prof_mark(ctx, prof_slot_base + 2);
    return MIDAS_OK;
}

MIDAS_WARM_TU(resample, k_gather_f64)

}  // namespace midas
