// peer_row.hpp - rows of the peer-mapped exchange (include/midas_hip.h, midas_peer_*): the inbox of a rank holds one 128-byte row
// per particle slot, stored by the rank that owns the slot's source particle straight into this rank's fine-grained memory, and
// a block of completion flags behind the rows.
//   piece (8 bytes)  0: int32 slot (local at the destination) | int32 global source     1: int32 NN index | int32 destination
//                    2: float64 weight    3: 0    4 .. 11: the 4 x 4 float32 pose, row-major    12 .. 15: 0
// A row is ONE 128-byte line: the owner's wave writes it with sixteen adjacent lanes (peer_rows_store), so that what crosses the
// fabric is whole lines, not eleven 8-byte pieces of eleven different store instructions.  Readers take the pieces with
// system-scope loads (another agent wrote them: the non-coherent caches are bypassed).
#pragma once
#include "midas_internal.hpp"

namespace midas {

constexpr int PEER_ROW = 128, PEER_PIECES = 12;

__device__ __forceinline__ void sys_store8(void* p, unsigned long long v) {
    __hip_atomic_store(reinterpret_cast<unsigned long long*>(p), v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long sys_load8(const void* p) {
    return __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ unsigned long long pack2(int lo, int hi) { return (unsigned long long)(unsigned)lo | ((unsigned long long)(unsigned)hi << 32); }
__device__ __forceinline__ unsigned long long pack2f(float lo, float hi) { return pack2(__float_as_int(lo), __float_as_int(hi)); }

// One wave: (first wave of the launch only) publish this rank's completion flag to every inbox, then wait until all G flags of
// the own inbox carry `tag`.  Bounded (2 s of the 100 MHz wall clock): a stuck peer must not hang the device.
__device__ __forceinline__ void peer_flags_publish_wait(const PeerInboxSrc& s, bool publish) {
    const int lane = threadIdx.x & 63;
    if (publish && s.peers && lane < s.G)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(s.peers[lane] + s.flag_off) + s.rank, s.tag, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (lane < s.G && !(publish && lane == s.rank)) {  // (the publisher's own rows are out: nothing to wait for there)
        const unsigned long long* f = reinterpret_cast<const unsigned long long*>(s.rows + s.flag_off) + lane;
        const long long t0 = wall_clock64();
        bool late = false;
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < s.tag) {
            __builtin_amdgcn_s_sleep(8);
            if (wall_clock64() - t0 > 200000000ll) { late = true; break; }
        }
        if (late && s.status) atomicOr(&s.status[0], 16);
    }
}

// The pieces of row r a reader needs (0 .. 2 and 4 .. 11), all requested before any is looked at.
struct PeerRow { unsigned long long head[3]; unsigned long long pose[8]; };
__device__ __forceinline__ PeerRow peer_row_load(const char* rows, int64_t r) {
    const char* rp = rows + (size_t)r * PEER_ROW;
    PeerRow v;
#pragma unroll
    for (int k = 0; k < 3; ++k) v.head[k] = sys_load8(rp + 8 * k);
#pragma unroll
    for (int k = 0; k < 8; ++k) v.pose[k] = sys_load8(rp + 32 + 8 * k);
    return v;
}

// Wave-cooperative store of the rows staged in LDS: stage[r][0 .. PEER_PIECES) = the pieces of the r-th row, dst[r] = its address
// in the destination's inbox, r < count.  Sixteen lanes per row, four rows per instruction, every row a whole 128-byte line
// (pieces 12 .. 15 are written as zeros).  The caller made the staged data visible to the wave (barrier / wave fence).
__device__ __forceinline__ void peer_rows_store(const unsigned long long (*stage)[PEER_PIECES], char* const* dst, int count) {
    const int lane = threadIdx.x & 63, sub = lane >> 4, p = lane & 15;
    for (int r0 = 0; r0 < count; r0 += 4) {
        const int r = r0 + sub;
        if (r < count) sys_store8(dst[r] + 8 * p, p < PEER_PIECES ? stage[r][p] : 0ull);
    }
}

}  // namespace midas
