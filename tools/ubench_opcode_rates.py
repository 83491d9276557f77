"""Issue rate of individual gfx950 VALU / SALU opcodes with 8 waves resident per SIMD (and with one): which integer
instructions run at the fast rate (~1.35 cycles per wave-instruction and SIMD, like v_add_u32) and which at the slow one
(~2.3-2.4, like v_lshlrev_b32)?  Generates one kernel per opcode (32 independent instructions per iteration over eight
registers), compiles with hipcc and runs.    python tools/ubench_opcode_rates.py   (GPU box)"""
import os
import subprocess
import sys
import tempfile

OPS = [  # (label, template: {d} = destination / accumulator VGPR, {s} = a second VGPR (another accumulator), {k} = a constant VGPR)
    ("v_add_u32", "v_add_u32 {d}, {k}, {d}"), ("v_sub_u32", "v_sub_u32 {d}, {d}, {k}"), ("v_and_b32", "v_and_b32 {d}, {k}, {d}"),
    ("v_or_b32", "v_or_b32 {d}, {k}, {d}"), ("v_xor_b32", "v_xor_b32 {d}, {k}, {d}"), ("v_not_b32", "v_not_b32 {d}, {d}"),
    ("v_mov_b32", "v_mov_b32 {d}, {k}"), ("v_lshlrev_b32", "v_lshlrev_b32 {d}, {k}, {d}"), ("v_lshrrev_b32", "v_lshrrev_b32 {d}, {k}, {d}"),
    ("v_ashrrev_i32", "v_ashrrev_i32 {d}, {k}, {d}"), ("v_min_i32", "v_min_i32 {d}, {k}, {d}"), ("v_max_i32", "v_max_i32 {d}, {k}, {d}"),
    ("v_min_u32", "v_min_u32 {d}, {k}, {d}"), ("v_max_u32", "v_max_u32 {d}, {k}, {d}"), ("v_bfe_u32", "v_bfe_u32 {d}, {d}, {k}, 5"),
    ("v_bfi_b32", "v_bfi_b32 {d}, {k}, {d}, {s}"), ("v_and_or_b32", "v_and_or_b32 {d}, {d}, {k}, {s}"), ("v_or3_b32", "v_or3_b32 {d}, {d}, {k}, {s}"),
    ("v_add3_u32", "v_add3_u32 {d}, {d}, {k}, {s}"), ("v_lshl_add_u32", "v_lshl_add_u32 {d}, {d}, 1, {k}"), ("v_add_lshl_u32", "v_add_lshl_u32 {d}, {d}, {k}, 1"),
    ("v_lshl_or_b32", "v_lshl_or_b32 {d}, {d}, 1, {k}"), ("v_mul_lo_u32", "v_mul_lo_u32 {d}, {d}, {k}"), ("v_mul_u32_u24", "v_mul_u32_u24 {d}, {d}, {k}"),
    ("v_mad_u32_u24", "v_mad_u32_u24 {d}, {d}, {k}, {s}"), ("v_mul_hi_u32", "v_mul_hi_u32 {d}, {d}, {k}"), ("v_bcnt_u32_b32", "v_bcnt_u32_b32 {d}, {d}, {k}"),
    ("v_mbcnt_lo_u32_b32", "v_mbcnt_lo_u32_b32 {d}, {d}, {k}"), ("v_ffbl_b32", "v_ffbl_b32 {d}, {d}"), ("v_cvt_f32_i32", "v_cvt_f32_i32 {d}, {d}"),
    ("v_cvt_f32_u32", "v_cvt_f32_u32 {d}, {d}"), ("v_mul_f32", "v_mul_f32 {d}, {k}, {d}"), ("v_fma_f32", "v_fma_f32 {d}, {d}, {k}, {s}"),
    ("v_rcp_f32", "v_rcp_f32 {d}, {d}"), ("v_cmp_lt_i32 -> vcc", "v_cmp_lt_i32 vcc, {d}, {k}"), ("v_cmp_eq_u32 -> sgpr pair", "v_cmp_eq_u32 s[20:21], {d}, {k}"),
    ("v_cndmask_b32 (sgpr mask)", "v_cndmask_b32_e64 {d}, {d}, {k}, s[20:21]"), ("v_mov_b32_dpp quad_perm", "v_mov_b32_dpp {d}, {s} quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf"),
    ("v_add_u32_dpp row_shr", "v_add_u32_dpp {d}, {s}, {d} row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1"), ("v_min_i32_dpp row_mirror", "v_min_i32_dpp {d}, {d}, {d} row_mirror row_mask:0xf bank_mask:0xf bound_ctrl:1"),
    ("v_readlane_b32", "v_readlane_b32 s20, {d}, 5"), ("v_readfirstlane_b32", "v_readfirstlane_b32 s20, {d}"), ("v_writelane_b32", "v_writelane_b32 {d}, s22, 5"),
    ("v_lshl_add_u64", "v_lshl_add_u64 {D}, {D}, 1, {D}"), ("v_mad_u64_u32", "v_mad_u64_u32 {D}, vcc, {d}, {k}, {D}"),
    ("s_add_u32", "s_add_u32 s20, s20, 1"), ("s_and_b64", "s_and_b64 s[20:21], s[20:21], exec"), ("s_bcnt1_i32_b64", "s_bcnt1_i32_b64 s22, s[20:21]"),
    ("s_ff1_i32_b64", "s_ff1_i32_b64 s22, s[20:21]"), ("s_lshl_b64", "s_lshl_b64 s[20:21], s[20:21], 1"), ("s_cselect_b32", "s_cselect_b32 s22, s20, s21"),
    ("s_mul_i32", "s_mul_i32 s22, s20, s21"), ("s_nop 0", "s_nop 0"),
]

SRC_HEAD = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
template <int K> __global__ __launch_bounds__(256) void k(unsigned long long *out, int n) {
    unsigned a0 = threadIdx.x, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    unsigned long long b0 = a0, b1 = a1, b2 = a2, b3 = a3;
    const unsigned sh = (threadIdx.x & 15) + 1;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < n; ++i) {
'''
SRC_TAIL = r'''
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[blockIdx.x] = t1 - t0;
    if (a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + b0 + b1 + b2 + b3 == 0x12345) out[0] = 1;
}
template <int K> double run(unsigned long long *d, int n_cu, int wps) {
    const int n = 2000, blocks = n_cu * wps;
    hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, 50);
    (void)hipDeviceSynchronize();
    hipLaunchKernelGGL(k<K>, dim3(blocks), dim3(256), 0, 0, d, n);
    (void)hipDeviceSynchronize();
    static unsigned long long h[8 * 512];
    (void)hipMemcpy(h, d, sizeof(unsigned long long) * blocks, hipMemcpyDeviceToHost);
    double cyc = 0;
    for (int b = 0; b < blocks; ++b) cyc += h[b];
    return cyc / blocks / (n * 32.0) / wps;      // cycles per wave-instruction and SIMD
}
'''


def body(i, tmpl):
    regs, wide = [f"%{r}" for r in range(8)], [f"%{r}" for r in range(9, 13)]
    lines = []
    for rep in range(4):
        for r in range(8):
            lines.append(tmpl.format(d=regs[r], s=regs[(r + 3) % 8], k="%8", D=wide[r % 4]))
    asm = "\\n ".join(lines)
    return (f'        if (K == {i}) asm volatile("{asm}" : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(sh), '
            f'"v"(b0), "v"(b1), "v"(b2), "v"(b3) : "s20", "s21", "s22", "s23", "vcc", "scc");\n')


def main():
    src = SRC_HEAD + "".join(body(i, t) for i, (_, t) in enumerate(OPS)) + SRC_TAIL
    src += "int main() {\n    int dev = 0, n_cu = 0; (void)hipGetDevice(&dev); (void)hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, dev);\n"
    src += "    unsigned long long *d; (void)hipMalloc(&d, sizeof(unsigned long long) * 8 * 512);\n"
    for i, (label, _) in enumerate(OPS):
        src += f'    printf("%-30s  8 waves / SIMD: %5.2f cycles per instruction and SIMD    1 wave: %5.2f\\n", "{label}", run<{i}>(d, n_cu, 8), run<{i}>(d, n_cu, 1));\n'
    src += "    return 0;\n}\n"
    tmp = tempfile.mkdtemp()
    path = os.path.join(tmp, "ops.hip")
    open(path, "w").write(src)
    if "--emit" in sys.argv:
        print(path)
        return
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O2", path, "-o", os.path.join(tmp, "ops")])
    subprocess.check_call([os.path.join(tmp, "ops")])


if __name__ == "__main__":
    main()
