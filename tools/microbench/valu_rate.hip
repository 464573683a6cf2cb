// Integer-VALU issue-rate microbenchmark for gfx950: how many wave64 instructions per second can the chip
// issue for the instruction kinds k_search is made of?  (measure, don't guess: anchors roofline_valu_issue)
//   hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip && ./valu_rate
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

#define REP 64
// clk[0] += shader-clock ticks (s_memtime), clk[1] += ticks of the constant 100 MHz reference (s_memrealtime) spent in the loop by
// one wavefront: their ratio is the REAL shader clock during the run (the table assumed 2.4 GHz instead of measuring)
template <int KIND>
__global__ __launch_bounds__(256) void k(uint32_t *out, int iters, uint32_t s0, uint32_t s1, unsigned long long *clk) {
    const unsigned long long c0 = clock64(), w0 = wall_clock64();
    uint32_t a = threadIdx.x * 2654435761u + s0, b = a ^ s1, c = a + 7, d = b + 11, e = a ^ 0x1234567u, f = b * 3u, g = c + d, h = e ^ f;
    unsigned long long p0 = ((unsigned long long)a << 32) | b, p1 = ((unsigned long long)c << 32) | d, p2 = ((unsigned long long)e << 32) | f,
                       p3 = ((unsigned long long)g << 32) | h;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int r = 0; r < REP / 8; ++r) {
            if (KIND == 0) {  // v_add_u32, 8 independent chains
                asm volatile("v_add_u32 %0, %0, %8\n v_add_u32 %1, %1, %8\n v_add_u32 %2, %2, %8\n v_add_u32 %3, %3, %8\n"
                             "v_add_u32 %4, %4, %8\n v_add_u32 %5, %5, %8\n v_add_u32 %6, %6, %8\n v_add_u32 %7, %7, %8\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else if (KIND == 1) {  // v_cmp_eq_u32 (to vcc) + v_cndmask_b32
                asm volatile("v_cmp_eq_u32 vcc, %0, %1\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_eq_u32 vcc, %4, %5\n v_cndmask_b32 %6, %6, %7, vcc\n"
                             "v_cmp_eq_u32 vcc, %1, %2\n v_cndmask_b32 %0, %0, %3, vcc\n v_cmp_eq_u32 vcc, %5, %6\n v_cndmask_b32 %4, %4, %7, vcc\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");
            } else if (KIND == 2) {  // v_mad_u32_u24
                asm volatile("v_mad_u32_u24 %0, %0, %8, %1\n v_mad_u32_u24 %1, %1, %8, %2\n v_mad_u32_u24 %2, %2, %8, %3\n v_mad_u32_u24 %3, %3, %8, %4\n"
                             "v_mad_u32_u24 %4, %4, %8, %5\n v_mad_u32_u24 %5, %5, %8, %6\n v_mad_u32_u24 %6, %6, %8, %7\n v_mad_u32_u24 %7, %7, %8, %0\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else if (KIND == 3) {  // v_mul_hi_u32_u24
                asm volatile("v_mul_hi_u32_u24 %0, %1, %8\n v_mul_hi_u32_u24 %1, %2, %8\n v_mul_hi_u32_u24 %2, %3, %8\n v_mul_hi_u32_u24 %3, %4, %8\n"
                             "v_mul_hi_u32_u24 %4, %5, %8\n v_mul_hi_u32_u24 %5, %6, %8\n v_mul_hi_u32_u24 %6, %7, %8\n v_mul_hi_u32_u24 %7, %0, %8\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else if (KIND == 4) {  // v_cmp_eq_u32_sdwa (WORD_1) + v_addc_co_u32
                asm volatile("v_cmp_eq_u32_sdwa vcc, %0, %1 src0_sel:WORD_1 src1_sel:DWORD\n v_addc_co_u32 %2, vcc, 0, %2, vcc\n"
                             "v_cmp_eq_u32_sdwa vcc, %3, %4 src0_sel:WORD_1 src1_sel:DWORD\n v_addc_co_u32 %5, vcc, 0, %5, vcc\n"
                             "v_cmp_eq_u32_sdwa vcc, %6, %7 src0_sel:WORD_1 src1_sel:DWORD\n v_addc_co_u32 %0, vcc, 0, %0, vcc\n"
                             "v_cmp_eq_u32_sdwa vcc, %1, %2 src0_sel:WORD_1 src1_sel:DWORD\n v_addc_co_u32 %3, vcc, 0, %3, vcc\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : : "vcc");
            } else if (KIND == 5) {  // v_mul_lo_u32 (full 32-bit multiply)
                asm volatile("v_mul_lo_u32 %0, %0, %8\n v_mul_lo_u32 %1, %1, %8\n v_mul_lo_u32 %2, %2, %8\n v_mul_lo_u32 %3, %3, %8\n"
                             "v_mul_lo_u32 %4, %4, %8\n v_mul_lo_u32 %5, %5, %8\n v_mul_lo_u32 %6, %6, %8\n v_mul_lo_u32 %7, %7, %8\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else if (KIND == 6) {  // v_min_u32 with DPP
                asm volatile("v_min_u32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_u32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_u32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             "v_min_u32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_min_u32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
            } else if (KIND == 8) {  // CONTROL: v_fma_f32 (the guide's table quotes 2 cycles per wave64 instruction for it)
                asm volatile("v_fma_f32 %0, %0, %8, %1\n v_fma_f32 %1, %1, %8, %2\n v_fma_f32 %2, %2, %8, %3\n v_fma_f32 %3, %3, %8, %4\n"
                             "v_fma_f32 %4, %4, %8, %5\n v_fma_f32 %5, %5, %8, %6\n v_fma_f32 %6, %6, %8, %7\n v_fma_f32 %7, %7, %8, %0\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else if (KIND == 9) {  // CONTROL: v_pk_fma_f32 (two fp32 FMAs per lane per instruction), 4 register pairs
                asm volatile("v_pk_fma_f32 %0, %0, %2, %1\n v_pk_fma_f32 %1, %1, %3, %2\n v_pk_fma_f32 %2, %2, %0, %3\n v_pk_fma_f32 %3, %3, %1, %0\n"
                             "v_pk_fma_f32 %0, %0, %2, %1\n v_pk_fma_f32 %1, %1, %3, %2\n v_pk_fma_f32 %2, %2, %0, %3\n v_pk_fma_f32 %3, %3, %1, %0\n"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3));
            } else if (KIND == 10) {  // CONTROL: fp32 FMA and integer add interleaved (do the two share one issue port?)
                asm volatile("v_fma_f32 %0, %0, %8, %1\n v_add_u32 %1, %1, %8\n v_fma_f32 %2, %2, %8, %3\n v_add_u32 %3, %3, %8\n"
                             "v_fma_f32 %4, %4, %8, %5\n v_add_u32 %5, %5, %8\n v_fma_f32 %6, %6, %8, %7\n v_add_u32 %7, %7, %8\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h) : "s"(s1));
            } else {  // v_and_b32 / v_xor_b32 / v_lshl_or_b32 mix
                asm volatile("v_and_b32 %0, %0, %1\n v_xor_b32 %1, %1, %2\n v_lshl_or_b32 %2, %3, 16, %2\n v_and_b32 %3, %3, %4\n"
                             "v_xor_b32 %4, %4, %5\n v_lshl_or_b32 %5, %6, 16, %5\n v_and_b32 %6, %6, %7\n v_xor_b32 %7, %7, %0\n"
                             : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(e), "+v"(f), "+v"(g), "+v"(h));
            }
        }
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = a ^ b ^ c ^ d ^ e ^ f ^ g ^ h ^ (uint32_t)(p0 ^ p1 ^ p2 ^ p3) ^ (uint32_t)((p0 ^ p1 ^ p2 ^ p3) >> 32);
    if (clk && blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = clock64() - c0; clk[1] = wall_clock64() - w0; }
}

template <int KIND>
double run(const char *name, uint32_t *d_out, int blocks, int threads = 256) {
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    static unsigned long long *d_clk = nullptr;
    if (!d_clk) hipMalloc(reinterpret_cast<void **>(&d_clk), 16);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, 10, 1u, 3u, d_clk);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(threads), 0, 0, d_out, iters, 1u, 3u, d_clk);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    unsigned long long clk[2] = {0, 0};
    hipMemcpy(clk, d_clk, 16, hipMemcpyDeviceToHost);
    const double ghz = clk[1] ? (double)clk[0] / (double)clk[1] * 0.1 : 0.0;   // s_memtime ticks per s_memrealtime tick x 100 MHz
    const double winst = (double)blocks * (threads / 64) * iters * REP;  // wave-instructions
    const double rate = winst / (ms * 1e-3);
    printf("%-44s %8.1f G wave-instr/s  = %.2f cycles per wave64 instruction per SIMD at the MEASURED shader clock %.3f GHz (%.2f at a nominal 2.4 GHz; 1024 SIMDs)\n",
           name, rate / 1e9, ghz > 0 ? 1024.0 * ghz * 1e9 / rate : 0.0, ghz, 1024.0 * 2.4e9 / rate);
    return rate;
}

int main() {
    uint32_t *d_out;
    const int blocks = 256 * 8;  // 8 waves per SIMD
    hipMalloc(&d_out, (size_t)blocks * 256 * 4);
    run<0>("v_add_u32", d_out, blocks);
    run<7>("v_and / v_xor / v_lshl_or", d_out, blocks);
    run<1>("v_cmp_eq_u32 + v_cndmask_b32", d_out, blocks);
    run<4>("v_cmp_eq_u32_sdwa + v_addc_co_u32", d_out, blocks);
    run<2>("v_mad_u32_u24", d_out, blocks);
    run<3>("v_mul_hi_u32_u24", d_out, blocks);
    run<5>("v_mul_lo_u32", d_out, blocks);
    run<6>("v_min_u32_dpp", d_out, blocks);
    // controls: does this harness reproduce the guide's 2-cycle figure for the fp32 FMA path?
    run<8>("CONTROL v_fma_f32", d_out, blocks);
    run<9>("CONTROL v_pk_fma_f32 (2 FMAs per lane)", d_out, blocks);
    run<10>("CONTROL v_fma_f32 / v_add_u32 interleaved", d_out, blocks);
    run<8>("CONTROL v_fma_f32, 1 wave per SIMD", d_out, 256, 256);
    run<0>("v_add_u32, 1 wave per SIMD", d_out, 256, 256);
    run<8>("CONTROL v_fma_f32, 2 waves per SIMD", d_out, 512, 256);
    run<0>("v_add_u32, 2 waves per SIMD", d_out, 512, 256);
    return 0;
}
