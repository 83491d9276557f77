// micro-benchmark: VALU issue THROUGHPUT of one SIMD on gfx950 with 1..8 waves resident (the headline kernel's question:
// is a pass VALU-issue bound?).  Each wave runs `n` iterations of 32 independent instructions of one kind (eight registers,
// four rounds); blocks of 256 threads = one wave per SIMD of a CU, `wps` blocks per CU.  Reported: wave-instructions per SIMD
// per shader cycle (s_memtime), and the time per instruction in ns at the measured clock.
//   hipcc --offload-arch=gfx950 -O2 tools/ubench_valu_peak.hip -o /tmp/ubench_valu_peak && /tmp/ubench_valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#define R4(x) x x x x
template <int K>
__global__ __launch_bounds__(256) void k(unsigned long long *out, int n) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const unsigned sh = threadIdx.x & 15;
    const unsigned long long t0 = __builtin_readcyclecounter();
    const unsigned long long w0 = wall_clock64();
    for (int i = 0; i < n; ++i) {
        if (K == 0) { R4(asm volatile("v_add_u32 %0, %8, %0\n v_add_u32 %1, %8, %1\n v_add_u32 %2, %8, %2\n v_add_u32 %3, %8, %3\n v_add_u32 %4, %8, %4\n v_add_u32 %5, %8, %5\n v_add_u32 %6, %8, %6\n v_add_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));) }
        if (K == 1) { R4(asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n v_lshlrev_b32 %4, %8, %4\n v_lshlrev_b32 %5, %8, %5\n v_lshlrev_b32 %6, %8, %6\n v_lshlrev_b32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));) }
        if (K == 2) { R4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_cndmask_b32 %1, %1, %8, vcc\n v_cndmask_b32 %2, %2, %8, vcc\n v_cndmask_b32 %3, %3, %8, vcc\n v_cndmask_b32 %4, %4, %8, vcc\n v_cndmask_b32 %5, %5, %8, vcc\n v_cndmask_b32 %6, %6, %8, vcc\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "vcc");) }
        if (K == 3) { R4(asm volatile("v_min_i32_dpp %0, %0, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %1, %1, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %2, %2, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %3, %3, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %4, %4, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %5, %5, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %6, %6, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_min_i32_dpp %7, %7, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (K == 4) { R4(asm volatile("v_cmp_lt_u32 vcc, %0, %1\n v_cmp_lt_u32 vcc, %1, %2\n v_cmp_lt_u32 vcc, %2, %3\n v_cmp_lt_u32 vcc, %3, %4\n v_cmp_lt_u32 vcc, %4, %5\n v_cmp_lt_u32 vcc, %5, %6\n v_cmp_lt_u32 vcc, %6, %7\n v_cmp_lt_u32 vcc, %7, %0" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "vcc");) }
        if (K == 5) { R4(asm volatile("v_cvt_f32_i32 %0, %0\n v_cvt_f32_i32 %1, %1\n v_cvt_f32_i32 %2, %2\n v_cvt_f32_i32 %3, %3\n v_fma_f32 %4, %4, %4, %4\n v_fma_f32 %5, %5, %5, %5\n v_fma_f32 %6, %6, %6, %6\n v_fma_f32 %7, %7, %7, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));) }
        if (K == 6) { R4(asm volatile("ds_bpermute_b32 %0, %8, %0\n ds_bpermute_b32 %1, %8, %1\n ds_bpermute_b32 %2, %8, %2\n ds_bpermute_b32 %3, %8, %3\n ds_bpermute_b32 %4, %8, %4\n ds_bpermute_b32 %5, %8, %5\n ds_bpermute_b32 %6, %8, %6\n ds_bpermute_b32 %7, %8, %7\n s_waitcnt lgkmcnt(0)" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh * 4));) }
        if (K == 7) { R4(asm volatile("s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_add_u32 s22, s22, 1\n s_and_b32 s23, s23, s22\n s_add_u32 s24, s24, 1\n s_and_b32 s25, s25, s24\n s_add_u32 s26, s26, 1\n s_and_b32 s27, s27, s26" : : : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");) }
        if (K == 8) { R4(asm volatile("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n v_readlane_b32 s24, %4, 11\n v_readlane_b32 s25, %5, 13\n v_readlane_b32 s26, %6, 15\n v_readlane_b32 s27, %7, 17" : : "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(a4), "v"(a5), "v"(a6), "v"(a7) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        // mixed streams in ONE wave: do the scalar and the vector unit overlap, or does a SIMD issue one instruction at a time?
        if (K == 9) { R4(asm volatile("v_lshlrev_b32 %0, %8, %0\n s_add_u32 s20, s20, 1\n v_lshlrev_b32 %1, %8, %1\n s_and_b32 s21, s21, s20\n v_lshlrev_b32 %2, %8, %2\n s_add_u32 s22, s22, 1\n v_lshlrev_b32 %3, %8, %3\n s_and_b32 s23, s23, s22" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "s20", "s21", "s22", "s23", "scc");) }
        if (K == 10) { R4(asm volatile("v_lshlrev_b32 %0, %8, %0\n v_lshlrev_b32 %1, %8, %1\n v_lshlrev_b32 %2, %8, %2\n v_lshlrev_b32 %3, %8, %3\n s_add_u32 s20, s20, 1\n s_and_b32 s21, s21, s20\n s_add_u32 s22, s22, 1\n s_and_b32 s23, s23, s22" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "s20", "s21", "s22", "s23", "scc");) }
        if (K == 11) { R4(asm volatile("v_and_b32 %0, %8, %0\n v_or_b32 %1, %8, %1\n v_xor_b32 %2, %8, %2\n v_sub_u32 %3, %8, %3\n v_max_i32 %4, %8, %4\n v_min_i32 %5, %8, %5\n v_bfe_u32 %6, %6, %8, 5\n v_lshl_add_u32 %7, %7, 1, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));) }
        // selects: v_cndmask with VCC, with an SGPR-pair mask, alternating with plain adds, and the mask rebuilt by a compare each time
        if (K == 12) { R4(asm volatile("v_cndmask_b32_e64 %0, %0, %8, s[20:21]\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cndmask_b32_e64 %2, %2, %8, s[20:21]\n v_cndmask_b32_e64 %3, %3, %8, s[20:21]\n v_cndmask_b32_e64 %4, %4, %8, s[20:21]\n v_cndmask_b32_e64 %5, %5, %8, s[20:21]\n v_cndmask_b32_e64 %6, %6, %8, s[20:21]\n v_cndmask_b32_e64 %7, %7, %8, s[20:21]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "s20", "s21");) }
        if (K == 13) { R4(asm volatile("v_cndmask_b32 %0, %0, %8, vcc\n v_add_u32 %1, %8, %1\n v_cndmask_b32 %2, %2, %8, vcc\n v_add_u32 %3, %8, %3\n v_cndmask_b32 %4, %4, %8, vcc\n v_add_u32 %5, %8, %5\n v_cndmask_b32 %6, %6, %8, vcc\n v_add_u32 %7, %8, %7" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "vcc");) }
        if (K == 14) { R4(asm volatile("v_cmp_lt_u32 vcc, %0, %8\n v_cndmask_b32 %1, %1, %8, vcc\n v_cmp_lt_u32 vcc, %2, %8\n v_cndmask_b32 %3, %3, %8, vcc\n v_cmp_lt_u32 vcc, %4, %8\n v_cndmask_b32 %5, %5, %8, vcc\n v_cmp_lt_u32 vcc, %6, %8\n v_cndmask_b32 %7, %7, %8, vcc" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "vcc");) }
        if (K == 15) { R4(asm volatile("v_cmp_lt_u32 s[20:21], %0, %8\n v_cndmask_b32_e64 %1, %1, %8, s[20:21]\n v_cmp_lt_u32 s[22:23], %2, %8\n v_cndmask_b32_e64 %3, %3, %8, s[22:23]\n v_cmp_lt_u32 s[24:25], %4, %8\n v_cndmask_b32_e64 %5, %5, %8, s[24:25]\n v_cmp_lt_u32 s[26:27], %6, %8\n v_cndmask_b32_e64 %7, %7, %8, s[26:27]" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");) }
        if (K == 16) { R4(asm volatile("v_min_u32 %0, %0, %8\n v_max_u32 %1, %1, %8\n v_bfi_b32 %2, %8, %2, %3\n v_and_or_b32 %3, %3, %8, %4\n v_mov_b32 %4, %8\n v_mov_b32 %5, %8\n v_mov_b32 %6, %8\n v_mov_b32 %7, %8" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh));) }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    const unsigned long long w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = t1 - t0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 == 0x12345) out[0] = 1;
}
template <int K> void run(const char *name, unsigned long long *d, int n_cu) {
    const int n = 4000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = n_cu * wps;     // 256 threads = 4 waves = one per SIMD of a CU; wps blocks per CU
        hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, 100);
        hipDeviceSynchronize();
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0, 0);
        hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, n);
        hipEventRecord(e1, 0);
        hipDeviceSynchronize();
        float ms = 0; hipEventElapsedTime(&ms, e0, e1);
        static unsigned long long h[2 * 8 * 512];
        hipMemcpy(h, d, sizeof(unsigned long long) * 2 * blocks, hipMemcpyDeviceToHost);
        double cyc = 0, wall = 0;
        for (int b = 0; b < blocks; ++b) { cyc += h[2 * b]; wall += h[2 * b + 1]; }
        cyc /= blocks; wall /= blocks;
        const double insts = (double)n * 32;                  // per wave
        const double ghz = cyc / (wall * 10.0);               // wall clock: 100 MHz
        printf("%-22s waves/SIMD %d: %.3f wave-instr / SIMD / cycle  (%.2f cycles each per SIMD; one wave: %.2f cycles per instr), clock %.2f GHz, kernel %.3f ms\n",
               name, wps, insts * wps / cyc, cyc / (insts * wps), cyc / insts, ghz, ms);
    }
}
int main() {
    int dev = 0, n_cu = 0; hipGetDevice(&dev); hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);
    unsigned long long *d; hipMalloc(&d, sizeof(unsigned long long) * 2 * 8 * 512);
    printf("CUs %d\n", n_cu);
    run<0>("v_add_u32", d, n_cu); run<1>("v_lshlrev_b32", d, n_cu); run<2>("v_cndmask_b32", d, n_cu); run<3>("v_min_i32_dpp", d, n_cu);
    run<4>("v_cmp_lt_u32", d, n_cu); run<5>("v_cvt_f32_i32+v_fma_f32", d, n_cu); run<6>("ds_bpermute_b32", d, n_cu); run<7>("s_add/s_and", d, n_cu);
    run<8>("v_readlane_b32", d, n_cu);
    run<12>("v_cndmask_e64 sgpr mask", d, n_cu); run<13>("v_cndmask / v_add altern.", d, n_cu); run<14>("v_cmp vcc + v_cndmask", d, n_cu);
    run<15>("v_cmp sgpr + v_cndmask_e64", d, n_cu); run<16>("min/max/bfi/and_or/4 v_mov", d, n_cu);
    run<9>("v_lshl / s_op alternating", d, n_cu); run<10>("4 v_lshl then 4 s_op", d, n_cu); run<11>("and/or/xor/sub/max/min/bfe/lshl_add", d, n_cu);
    return 0;
}
