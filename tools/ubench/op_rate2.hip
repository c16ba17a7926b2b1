// Microbenchmark, second set: op_rate.hip priced instruction CLASSES; this one prices single opcodes and short
// PATTERNS of the trace kernel's non-arithmetic half (compares, selects, integer address arithmetic, lane reads),
// because two readings of op_rate.hip were left open: (1) `v_cndmask_b32_e32` back to back costs 23 cycles but 2.5
// when it alternates with v_mul -- which neighbours are the expensive ones? (2) the "and/add/lshl/bfe mix" costs 4.2
// but v_xor_b32 2.3 -- which integer opcodes are in the fast class?
// 4 waves per SIMD (as the resident trace kernel), 4 independent accumulators, cycles per instruction per SIMD @ 2.4 GHz.
// Build: hipcc --offload-arch=gfx950 -O3 -o op_rate2 op_rate2.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define R8(s) s s s s s s s s
// every body is 4 instructions (or 4 groups) on x0..x3 / y0..y3; a = float operand, c = int operand
#define FOUR(op) op(%0) op(%1) op(%2) op(%3)
#define OPS_F(body) asm volatile(R8(body) : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a), "v"(b) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")
#define OPS_I(body) asm volatile(R8(body) : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3) : "v"(c), "v"(e) : "vcc", "scc", "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27")

template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b, int c, int e) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int y0 = threadIdx.x * 4, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3;
    asm volatile("v_cmp_lt_f32 vcc, %0, %1\n v_cmp_lt_f32 s[20:21], %0, %1\n v_cmp_gt_f32 s[22:23], %0, %1" :: "v"(x0), "v"(a) : "vcc", "s20", "s21", "s22", "s23");
    for (int i = 0; i < N_ITER; ++i) {
        // ---- integer opcodes, one at a time
        if (KIND == 0) OPS_I("v_and_b32 %0, %0, %4\n v_and_b32 %1, %1, %4\n v_and_b32 %2, %2, %4\n v_and_b32 %3, %3, %4\n");
        if (KIND == 1) OPS_I("v_or_b32 %0, %0, %4\n v_or_b32 %1, %1, %4\n v_or_b32 %2, %2, %4\n v_or_b32 %3, %3, %4\n");
        if (KIND == 2) OPS_I("v_add_u32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_add_u32 %2, %2, %4\n v_add_u32 %3, %3, %4\n");
        if (KIND == 3) OPS_I("v_sub_u32 %0, %0, %4\n v_sub_u32 %1, %1, %4\n v_sub_u32 %2, %2, %4\n v_sub_u32 %3, %3, %4\n");
        if (KIND == 4) OPS_I("v_lshlrev_b32 %0, 4, %0\n v_lshlrev_b32 %1, 4, %1\n v_lshlrev_b32 %2, 4, %2\n v_lshlrev_b32 %3, 4, %3\n");
        if (KIND == 5) OPS_I("v_lshrrev_b32 %0, 3, %0\n v_lshrrev_b32 %1, 3, %1\n v_lshrrev_b32 %2, 3, %2\n v_lshrrev_b32 %3, 3, %3\n");
        if (KIND == 6) OPS_I("v_bfe_u32 %0, %0, 3, 16\n v_bfe_u32 %1, %1, 3, 16\n v_bfe_u32 %2, %2, 3, 16\n v_bfe_u32 %3, %3, 3, 16\n");
        if (KIND == 7) OPS_I("v_and_b32 %0, 0xffff, %0\n v_and_b32 %1, 0xffff, %1\n v_and_b32 %2, 0xffff, %2\n v_and_b32 %3, 0xffff, %3\n");   // 32-bit literal
        if (KIND == 8) OPS_I("v_add3_u32 %0, %0, %4, %5\n v_add3_u32 %1, %1, %4, %5\n v_add3_u32 %2, %2, %4, %5\n v_add3_u32 %3, %3, %4, %5\n");
        if (KIND == 9) OPS_I("v_mul_lo_u32 %0, %0, %4\n v_mul_lo_u32 %1, %1, %4\n v_mul_lo_u32 %2, %2, %4\n v_mul_lo_u32 %3, %3, %4\n");
        if (KIND == 10) OPS_I("v_mul_u32_u24 %0, %0, %4\n v_mul_u32_u24 %1, %1, %4\n v_mul_u32_u24 %2, %2, %4\n v_mul_u32_u24 %3, %3, %4\n");
        if (KIND == 11) OPS_I("v_mad_u32_u24 %0, %0, %4, %5\n v_mad_u32_u24 %1, %1, %4, %5\n v_mad_u32_u24 %2, %2, %4, %5\n v_mad_u32_u24 %3, %3, %4, %5\n");
        if (KIND == 12) OPS_I("v_bcnt_u32_b32 %0, %0, %4\n v_bcnt_u32_b32 %1, %1, %4\n v_bcnt_u32_b32 %2, %2, %4\n v_bcnt_u32_b32 %3, %3, %4\n");
        if (KIND == 13) OPS_I("v_perm_b32 %0, %0, %4, %5\n v_perm_b32 %1, %1, %4, %5\n v_perm_b32 %2, %2, %4, %5\n v_perm_b32 %3, %3, %4, %5\n");
        if (KIND == 14) OPS_I("v_bfi_b32 %0, %0, %4, %5\n v_bfi_b32 %1, %1, %4, %5\n v_bfi_b32 %2, %2, %4, %5\n v_bfi_b32 %3, %3, %4, %5\n");
        if (KIND == 15) OPS_I("v_lshl_or_b32 %0, %0, 2, %4\n v_lshl_or_b32 %1, %1, 2, %4\n v_lshl_or_b32 %2, %2, 2, %4\n v_lshl_or_b32 %3, %3, 2, %4\n");
        if (KIND == 16) OPS_I("v_and_or_b32 %0, %0, %4, %5\n v_and_or_b32 %1, %1, %4, %5\n v_and_or_b32 %2, %2, %4, %5\n v_and_or_b32 %3, %3, %4, %5\n");
        if (KIND == 17) OPS_I("v_max_i32 %0, %0, %4\n v_min_i32 %1, %1, %4\n v_max_u32 %2, %2, %4\n v_min_u32 %3, %3, %4\n");
        if (KIND == 18) OPS_I("v_not_b32 %0, %0\n v_not_b32 %1, %1\n v_not_b32 %2, %2\n v_not_b32 %3, %3\n");
        if (KIND == 19) OPS_I("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n");
        if (KIND == 20) OPS_I("v_mbcnt_lo_u32_b32 %0, %4, %0\n v_mbcnt_hi_u32_b32 %1, %4, %1\n v_mbcnt_lo_u32_b32 %2, %4, %2\n v_mbcnt_hi_u32_b32 %3, %4, %3\n");
        if (KIND == 21) OPS_I("v_lshlrev_b16 %0, 4, %0\n v_lshlrev_b16 %1, 4, %1\n v_lshlrev_b16 %2, 4, %2\n v_lshlrev_b16 %3, 4, %3\n");
        if (KIND == 22) OPS_I("v_add_u32_sdwa %0, %0, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %1, %1, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n"
                              "v_add_u32_sdwa %2, %2, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0 src1_sel:DWORD\n v_add_u32_sdwa %3, %3, %4 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD\n");
        // ---- fp32 opcodes / encodings
        if (KIND == 30) OPS_F("v_fmac_f32 %0, %4, %5\n v_fmac_f32 %1, %4, %5\n v_fmac_f32 %2, %4, %5\n v_fmac_f32 %3, %4, %5\n");                     // VOP2 FMA
        if (KIND == 31) OPS_F("v_mul_f32_e64 %0, %0, -%4\n v_mul_f32_e64 %1, %1, -%4\n v_mul_f32_e64 %2, %2, -%4\n v_mul_f32_e64 %3, %3, -%4\n");     // VOP3 encoding of a fast op
        if (KIND == 32) OPS_F("v_add_f32_e64 %0, |%0|, %4\n v_add_f32_e64 %1, |%1|, %4\n v_add_f32_e64 %2, |%2|, %4\n v_add_f32_e64 %3, |%3|, %4\n");
        if (KIND == 33) OPS_F("v_med3_f32 %0, %0, %4, %5\n v_med3_f32 %1, %1, %4, %5\n v_med3_f32 %2, %2, %4, %5\n v_med3_f32 %3, %3, %4, %5\n");
        if (KIND == 34) OPS_F("v_max_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_max_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n");
        if (KIND == 35) OPS_F("v_cmp_class_f32 s[24:25], %0, %5\n v_cmp_class_f32 s[26:27], %1, %5\n v_cmp_class_f32 s[24:25], %2, %5\n v_cmp_class_f32 s[26:27], %3, %5\n");
        if (KIND == 36) OPS_F("v_cmp_eq_u32 vcc, %0, %4\n v_cmp_eq_u32 vcc, %1, %4\n v_cmp_eq_u32 vcc, %2, %4\n v_cmp_eq_u32 vcc, %3, %4\n");
        if (KIND == 37) OPS_F("v_mul_f32 %0, 0x40490fdb, %0\n v_mul_f32 %1, 0x40490fdb, %1\n v_mul_f32 %2, 0x40490fdb, %2\n v_mul_f32 %3, 0x40490fdb, %3\n");   // fast op + 32-bit literal
        if (KIND == 38) OPS_F("v_mul_f32 %0, s20, %0\n v_mul_f32 %1, s21, %1\n v_mul_f32 %2, s22, %2\n v_mul_f32 %3, s23, %3\n");                     // SGPR operand
        if (KIND == 39) OPS_F("v_fma_f32 %0, %0, %4, 1.0\n v_fma_f32 %1, %1, %4, 1.0\n v_fma_f32 %2, %2, %4, 1.0\n v_fma_f32 %3, %3, %4, 1.0\n");
        // ---- select / compare patterns
        if (KIND == 40) OPS_F("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n");   // e32 / e64 alternating
        if (KIND == 41) OPS_F("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n");   // two e32 selects in a row, then two muls
        if (KIND == 42) OPS_F("v_cndmask_b32_e32 %0, %0, %4, vcc\n s_nop 0\n v_cndmask_b32_e32 %1, %1, %4, vcc\n s_nop 0\n v_cndmask_b32_e32 %2, %2, %4, vcc\n s_nop 0\n v_cndmask_b32_e32 %3, %3, %4, vcc\n s_nop 0\n");   // counted as 4
        if (KIND == 43) OPS_F("v_cmp_lt_f32 vcc, %0, %4\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_cmp_lt_f32 vcc, %1, %4\n v_cndmask_b32_e32 %1, %1, %5, vcc\n");   // dependent cmp -> select through VCC (2 pairs)
        if (KIND == 44) OPS_F("v_cmp_lt_f32 s[24:25], %0, %4\n v_cndmask_b32_e64 %0, %0, %5, s[24:25]\n v_cmp_lt_f32 s[26:27], %1, %4\n v_cndmask_b32_e64 %1, %1, %5, s[26:27]\n");   // the same through an SGPR pair
        if (KIND == 45) OPS_F("v_cmp_lt_f32 vcc, %0, %4\n v_mul_f32 %2, %2, %4\n v_cndmask_b32_e32 %0, %0, %5, vcc\n v_mul_f32 %3, %3, %4\n");       // cmp, mul, select, mul
        if (KIND == 46) OPS_F("v_cndmask_b32_e64 %0, %0, %4, vcc\n v_cndmask_b32_e64 %1, %1, %4, vcc\n v_cndmask_b32_e64 %2, %2, %4, vcc\n v_cndmask_b32_e64 %3, %3, %4, vcc\n");   // VOP3 encoding, VCC as the mask
        if (KIND == 47) OPS_F("v_cndmask_b32_e32 %0, %4, %5, vcc\n v_add_f32 %1, %1, %4\n v_cndmask_b32_e32 %2, %4, %5, vcc\n v_sub_f32 %3, %3, %4\n");
        if (KIND == 48) OPS_F("v_cndmask_b32_e32 %0, %0, %4, vcc\n v_min_f32 %1, %1, %4\n v_cndmask_b32_e32 %2, %2, %4, vcc\n v_max_f32 %3, %3, %4\n");   // select beside a 4-cycle op
        if (KIND == 49) OPS_F("v_cmp_lt_f32 vcc, %0, %4\n s_and_b64 s[24:25], vcc, s[20:21]\n v_cmp_lt_f32 vcc, %1, %4\n s_and_b64 s[26:27], vcc, s[22:23]\n");   // counted as 4: compare + mask arithmetic
        // ---- lane reads, scans
        if (KIND == 50) OPS_F("v_readfirstlane_b32 s24, %0\n v_readfirstlane_b32 s25, %1\n v_readfirstlane_b32 s26, %2\n v_readfirstlane_b32 s27, %3\n");
        if (KIND == 51) OPS_F("v_mov_b32_dpp %0, %1 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %1, %2 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %2, %3 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n v_mov_b32_dpp %3, %0 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n");
        if (KIND == 52) OPS_F("v_add_u32_dpp %0, %1, %0 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %1, %2, %1 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %2, %3, %2 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:1\n v_add_u32_dpp %3, %0, %3 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:1\n");
        if (KIND == 53) OPS_F("v_permlane32_swap_b32 %0, %1\n v_permlane32_swap_b32 %2, %3\n v_permlane16_swap_b32 %0, %2\n v_permlane16_swap_b32 %1, %3\n");   // gfx950
        if (KIND == 54) OPS_F("v_writelane_b32 %0, s20, 3\n v_writelane_b32 %1, s21, 5\n v_writelane_b32 %2, s22, 7\n v_writelane_b32 %3, s23, 9\n");
        // ---- scalar side
        if (KIND == 60) OPS_F("s_and_saveexec_b64 s[24:25], s[20:21]\n s_or_b64 exec, exec, s[24:25]\n s_and_saveexec_b64 s[26:27], s[22:23]\n s_or_b64 exec, exec, s[26:27]\n");
        if (KIND == 61) OPS_F("s_bcnt1_i32_b64 s24, s[20:21]\n s_ff1_i32_b64 s25, s[22:23]\n s_bcnt1_i32_b64 s26, s[20:21]\n s_ff1_i32_b64 s27, s[22:23]\n");
        if (KIND == 62) OPS_F("s_cmp_lg_u64 s[20:21], 0\n s_cbranch_scc0 1f\n 1: s_cmp_lg_u64 s[22:23], 0\n s_cbranch_scc0 2f\n 2:\n");           // counted as 4: compare + untaken-or-taken short branch
        if (KIND == 63) OPS_F("s_mov_b64 s[24:25], exec\n s_mov_b64 exec, s[24:25]\n s_mov_b64 s[26:27], exec\n s_mov_b64 exec, s[26:27]\n");
        if (KIND == 64) OPS_F("v_mul_f32 %0, %0, %4\n s_and_b64 s[24:25], s[20:21], s[22:23]\n s_or_b64 s[26:27], s[20:21], s[22:23]\n v_mul_f32 %1, %1, %4\n");   // 2 VALU + 2 SALU, counted as 4
        if (KIND == 65) OPS_F("v_min_f32 %0, %0, %4\n s_and_b64 s[24:25], s[20:21], s[22:23]\n v_max_f32 %1, %1, %4\n s_or_b64 s[26:27], s[20:21], s[22:23]\n");   // 4-cycle VALU + SALU alternating
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + (float)(y0 + y1 + y2 + y3);
}
template <int KIND> void run(const char* name, int per_iter = 32, int waves_per_simd = 4) {
    float* d; hipMalloc(&d, 16 << 20);
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0x00ff00ff, 0x01000302); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f, 0x00ff00ff, 0x01000302); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITER * per_iter;
    printf("%-58s %6.2f cycles per instruction per SIMD\n", name, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
    hipFree(d); hipEventDestroy(e0); hipEventDestroy(e1);
}
int main() {
    printf("# op_rate2: 4 waves per SIMD, 2.4 GHz assumed\n");
    run<0>("v_and_b32"); run<1>("v_or_b32"); run<19>("v_xor_b32"); run<18>("v_not_b32"); run<2>("v_add_u32"); run<3>("v_sub_u32");
    run<4>("v_lshlrev_b32"); run<5>("v_lshrrev_b32"); run<6>("v_bfe_u32"); run<7>("v_and_b32 with a 32-bit literal"); run<8>("v_add3_u32");
    run<9>("v_mul_lo_u32"); run<10>("v_mul_u32_u24"); run<11>("v_mad_u32_u24"); run<12>("v_bcnt_u32_b32"); run<13>("v_perm_b32"); run<14>("v_bfi_b32");
    run<15>("v_lshl_or_b32"); run<16>("v_and_or_b32"); run<17>("v_min/max_i32/u32"); run<20>("v_mbcnt_lo/hi"); run<21>("v_lshlrev_b16"); run<22>("v_add_u32_sdwa (WORD source)");
    run<30>("v_fmac_f32 (VOP2)"); run<39>("v_fma_f32 with an inline constant"); run<31>("v_mul_f32_e64 with neg"); run<32>("v_add_f32_e64 with abs"); run<33>("v_med3_f32"); run<34>("v_max_f32 alone");
    run<35>("v_cmp_class_f32 -> sgpr pair"); run<36>("v_cmp_eq_u32 vcc"); run<37>("v_mul_f32 with a 32-bit literal"); run<38>("v_mul_f32 with an SGPR operand");
    run<40>("v_cndmask e32 / e64 alternating"); run<41>("2 x v_cndmask_e32 then 2 x v_mul"); run<42>("v_cndmask_e32 + s_nop 0 (per pair)", 32);
    run<43>("v_cmp vcc -> v_cndmask_e32 vcc, dependent (per instr)"); run<44>("v_cmp sgpr -> v_cndmask_e64 sgpr, dependent (per instr)"); run<45>("cmp, mul, select, mul");
    run<46>("v_cndmask_b32_e64 with VCC"); run<47>("v_cndmask_e32 (no RAW) / add / sub alternating"); run<48>("v_cndmask_e32 / v_min / v_max alternating"); run<49>("v_cmp vcc + s_and_b64 (per instr)");
    run<50>("v_readfirstlane_b32"); run<51>("v_mov_b32_dpp quad_perm"); run<52>("v_add_u32_dpp row_shr"); run<53>("v_permlane32/16_swap_b32"); run<54>("v_writelane_b32");
    run<60>("s_and_saveexec_b64 + s_or_b64 exec (per instr)"); run<61>("s_bcnt1 / s_ff1"); run<62>("s_cmp + short s_cbranch (per instr)"); run<63>("s_mov_b64 from / to exec");
    run<64>("2 v_mul + 2 s_and/s_or (per instr)"); run<65>("v_min/v_max + s_and/s_or alternating (per instr)");
    return 0;
}
