// micro-benchmark: issue cost of a few VALU instructions on gfx950 (one wave per SIMD and 8 waves per SIMD)
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(x) x x x x x x x x
template <int K>
__global__ void k(unsigned long long *out, int n) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    unsigned long long b0 = a0 * 0x9E3779B97F4A7C15ull, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    unsigned sh = threadIdx.x & 48;
    unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
        if (K == 0) { REP8(asm volatile("v_lshrrev_b32 %0, %4, %0\n v_lshrrev_b32 %1, %4, %1\n v_lshrrev_b32 %2, %4, %2\n v_lshrrev_b32 %3, %4, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));) }
        if (K == 1) { REP8(asm volatile("v_lshrrev_b64 %0, %4, %0\n v_lshrrev_b64 %1, %4, %1\n v_lshrrev_b64 %2, %4, %2\n v_lshrrev_b64 %3, %4, %3" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3) : "v"(sh));) }
        if (K == 2) { REP8(asm volatile("v_bfe_u32 %0, %0, %4, 16\n v_bfe_u32 %1, %1, %4, 16\n v_bfe_u32 %2, %2, %4, 16\n v_bfe_u32 %3, %3, %4, 16" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));) }
        if (K == 3) { REP8(asm volatile("v_lshl_add_u64 %0, %0, 2, %1\n v_lshl_add_u64 %1, %1, 2, %2\n v_lshl_add_u64 %2, %2, 2, %3\n v_lshl_add_u64 %3, %3, 2, %0" : "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3));) }
        if (K == 4) { REP8(asm volatile("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(sh));) }
        if (K == 5) { REP8(asm volatile("v_mov_b32_dpp %0, %1 row_shr:1 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_shr:2 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_shr:4 row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_shr:8 row_mask:0xf bank_mask:0xf" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (K == 6) { REP8(asm volatile("v_cmp_ne_u32 vcc, %0, %1\n v_cmp_ne_u32 vcc, %1, %2\n v_cmp_ne_u32 vcc, %2, %3\n v_cmp_ne_u32 vcc, %3, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3) : "vcc");) }
        if (K == 7) { REP8(asm volatile("s_and_b64 s[20:21], s[20:21], exec\n s_or_b64 s[22:23], s[20:21], exec\n s_and_b64 s[20:21], s[22:23], exec\n s_or_b64 s[22:23], s[20:21], exec" : : : "s20", "s21", "s22", "s23");) }
        if (K == 8) { REP8(asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3));) }
        if (K == 9) { REP8(asm volatile("v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1\n v_mad_u64_u32 %0, vcc, %2, %3, %0\n v_mad_u64_u32 %1, vcc, %2, %3, %1" : "+v"(b0), "+v"(b1) : "v"(a2), "v"(a3) : "vcc");) }
    }
    unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0 && blockIdx.x == 0) out[0] = t1 - t0;
    if (a0 + a1 + a2 + a3 + b0 + b1 + b2 + b3 == 0x12345) out[1] = 1;
}
template <int K> void run(const char *name, unsigned long long *d) {
    for (int wpb : {64, 512}) {    // 1 wave per CU-SIMD... 64 threads = 1 wave; 512 = 8 waves on one CU (2 per SIMD)
        for (int blocks : {1, 4}) {
            hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(wpb), 0, 0, d, 1000);
            hipDeviceSynchronize();
            unsigned long long h[2];
            hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
            printf("%-16s threads/block %4d blocks %d: %.2f cycles per instruction (wave 0)\n", name, wpb, blocks, h[0] / (1000.0 * 32));
        }
    }
}
int main() {
    unsigned long long *d; hipMalloc(&d, 16);
    run<0>("v_lshrrev_b32", d); run<1>("v_lshrrev_b64", d); run<2>("v_bfe_u32", d); run<3>("v_lshl_add_u64", d);
    run<4>("v_mul_lo_u32", d); run<5>("v_mov_b32_dpp", d); run<6>("v_cmp_ne_u32", d); run<7>("s_and/or_b64", d);
    run<8>("v_cvt_f32_i32", d); run<9>("v_mad_u64_u32", d);
    return 0;
}
