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
