// kao_device.h -- wavefront primitives shared by the gfx950 translation units (kao_kernels.hip, kao_bound.hip).
// wave64 only; every lane of the wavefront must be active at the call.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace kao {

// Wavefront min (all 64 lanes active): butterfly inside each row of 16 with four fused v_min_u32_dpp
// (quad_perm xor1, xor2, row_half_mirror, row_mirror; hipcc emits mov_dpp + min pairs for the builtin form),
// then the four row minima are read with v_readlane and combined on the scalar unit.  Result is wave-uniform.
// s_nop 1 = the 2 wait states a DPP read needs after the VALU write of its source.
__device__ __forceinline__ uint32_t wave_umin(uint32_t v) {
    asm("s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_min_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    uint32_t a = (uint32_t)__builtin_amdgcn_readlane((int)v, 0);
    uint32_t b = (uint32_t)__builtin_amdgcn_readlane((int)v, 16);
    uint32_t c = (uint32_t)__builtin_amdgcn_readlane((int)v, 32);
    uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)v, 48);
    return min(min(a, b), min(c, d));
}
// Wavefront sum, same structure with fused v_add_u32_dpp.
__device__ __forceinline__ int wave_sum(int v) {
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(v));
    return __builtin_amdgcn_readlane(v, 0) + __builtin_amdgcn_readlane(v, 16) +
           __builtin_amdgcn_readlane(v, 32) + __builtin_amdgcn_readlane(v, 48);
}
// Six wavefront sums at once (round 6; K-eval folded its six partial sums one wave_sum after the other: ~100 of a small candidate's
// ~450 instructions).  v_permlane32_swap / v_permlane16_swap (gfx950) exchange the upper half / the odd rows of one register with the
// lower half / the even rows of another, so a swap and an add fold two values into one register whose halves (rows) hold the two
// partial sums: six values -> three -> two registers, and the rows of 16 are then summed by the fused DPP butterfly of wave_sum, both
// registers in one block (a DPP read wants two wait states behind the write of its source: the other chain's add is one of them).
// Totals come back wave-uniform in v[0..5].  33 instructions against 96.
__device__ __forceinline__ int fold_halves(int a, int b) {   // lanes 0..31: a[l] + a[l + 32]; lanes 32..63: b[l - 32] + b[l]
    const auto r = __builtin_amdgcn_permlane32_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ int fold_rows(int a, int b) {     // rows 0 / 2: a's row + the row above it; rows 1 / 3: b's row below + its own
    const auto r = __builtin_amdgcn_permlane16_swap((unsigned)a, (unsigned)b, false, false);
    return (int)(r[0] + r[1]);
}
__device__ __forceinline__ void wave_sum6(int (&v)[6]) {
    const int w0 = fold_halves(v[0], v[3]), w1 = fold_halves(v[1], v[4]), w2 = fold_halves(v[2], v[5]);
    int u0 = fold_rows(w0, w1);      // rows: v0 v1 v3 v4
    int u1 = fold_rows(w2, 0);       // rows: v2 0 v5 0
    asm("s_nop 1\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 0\n\t"
        "v_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "v_add_u32_dpp %1, %1, %1 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(u0), "+v"(u1));
    v[0] = __builtin_amdgcn_readlane(u0, 0); v[1] = __builtin_amdgcn_readlane(u0, 16);
    v[3] = __builtin_amdgcn_readlane(u0, 32); v[4] = __builtin_amdgcn_readlane(u0, 48);
    v[2] = __builtin_amdgcn_readlane(u1, 0); v[5] = __builtin_amdgcn_readlane(u1, 32);
}
// Four wavefront sums at once: two half folds, one row fold (rows: v0 v1 v2 v3), one DPP block, four v_readlane -- 19 instructions.
__device__ __forceinline__ void wave_sum4(int (&v)[4]) {
    int u = fold_rows(fold_halves(v[0], v[2]), fold_halves(v[1], v[3]));
    asm("s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 quad_perm:[2,3,0,1] row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_half_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1\n\tv_add_u32_dpp %0, %0, %0 row_mirror row_mask:0xf bank_mask:0xf\n\t"
        "s_nop 1"
        : "+v"(u));
    v[0] = __builtin_amdgcn_readlane(u, 0); v[1] = __builtin_amdgcn_readlane(u, 16);
    v[2] = __builtin_amdgcn_readlane(u, 32); v[3] = __builtin_amdgcn_readlane(u, 48);
}
// lanes of the wavefront (active ones) on which `c` holds: v_cmp into a scalar pair + s_bcnt1 -- a count that needs no vector register, no reduction
__device__ __forceinline__ int wave_count(bool c) { return __builtin_popcountll(__ballot(c)); }
__device__ __forceinline__ uint32_t wave_umax(uint32_t v) { return ~wave_umin(~v); }
__device__ __forceinline__ long long wave_sum64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

// largest dynamic-LDS size a kernel has been enabled for is tracked per device (function attributes are per device)
constexpr int kAttrDevices = 64;
static inline int attr_slot() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < kAttrDevices) ? d : 0; }

}  // namespace kao
