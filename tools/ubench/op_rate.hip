// Microbenchmark: what one instruction of each kind costs a SIMD of gfx950 when 1, 2 or 4 waves share it (the trace
// kernel runs 4 per SIMD), alone and interleaved with scalar instructions.  Answers "which instructions of the pooled
// trace kernel are expensive": DPP moves, v_readlane, ds_bpermute, the pieces of the IEEE divide, compares.
// Build: hipcc --offload-arch=gfx950 -O3 -o op_rate op_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 2048
#define R8(s) s s s s s s s s
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    int i0 = threadIdx.x * 4, i1 = (threadIdx.x * 7 + 3) & 255;
    __shared__ float lds[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = i;
    __syncthreads();
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0) asm volatile(R8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %1, %4\n v_mul_f32 %2, %2, %4\n v_mul_f32 %3, %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 1) asm volatile(R8("v_fma_f32 %0, %0, %4, %4\n v_fma_f32 %1, %1, %4, %4\n v_fma_f32 %2, %2, %4, %4\n v_fma_f32 %3, %3, %4, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 2) asm volatile(R8("v_cmp_lt_f32 s[20:21], %0, %4\n v_cmp_lt_f32 s[22:23], %1, %4\n v_cmp_lt_f32 s[24:25], %2, %4\n v_cmp_lt_f32 s[26:27], %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27");
        if (KIND == 3) asm volatile(R8("v_cndmask_b32 %0, %0, %4, vcc\n v_cndmask_b32 %1, %1, %4, vcc\n v_cndmask_b32 %2, %2, %4, vcc\n v_cndmask_b32 %3, %3, %4, vcc\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 4) asm volatile(R8("v_max_i32_dpp %0, %1, %1 row_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_i32_dpp %1, %2, %2 row_shr:2 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_i32_dpp %2, %3, %3 row_shr:4 row_mask:0xf bank_mask:0xf bound_ctrl:0\n v_max_i32_dpp %3, %0, %0 row_shr:8 row_mask:0xf bank_mask:0xf bound_ctrl:0\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 5) asm volatile(R8("v_mov_b32_dpp %0, %1 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %1, %2 row_bcast:31 row_mask:0xc bank_mask:0xf\n v_mov_b32_dpp %2, %3 row_bcast:15 row_mask:0xa bank_mask:0xf\n v_mov_b32_dpp %3, %0 row_bcast:31 row_mask:0xc bank_mask:0xf\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 6) asm volatile(R8("v_readlane_b32 s20, %0, 3\n v_readlane_b32 s21, %1, 5\n v_readlane_b32 s22, %2, 7\n v_readlane_b32 s23, %3, 9\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23");
        if (KIND == 7) asm volatile(R8("ds_bpermute_b32 %0, %4, %0\n ds_bpermute_b32 %1, %4, %1\n ds_bpermute_b32 %2, %4, %2\n ds_bpermute_b32 %3, %4, %3\n s_waitcnt lgkmcnt(0)\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(i1 * 4 & 255));
        if (KIND == 8) asm volatile(R8("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 9) asm volatile(R8("v_div_scale_f32 %0, vcc, %0, %4, %4\n v_div_scale_f32 %1, vcc, %1, %4, %4\n v_div_scale_f32 %2, vcc, %2, %4, %4\n v_div_scale_f32 %3, vcc, %3, %4, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 10) asm volatile(R8("v_div_fmas_f32 %0, %0, %4, %4\n v_div_fmas_f32 %1, %1, %4, %4\n v_div_fmas_f32 %2, %2, %4, %4\n v_div_fmas_f32 %3, %3, %4, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 11) asm volatile(R8("v_div_fixup_f32 %0, %0, %4, %4\n v_div_fixup_f32 %1, %1, %4, %4\n v_div_fixup_f32 %2, %2, %4, %4\n v_div_fixup_f32 %3, %3, %4, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 12) asm volatile(R8("v_lshl_add_u32 %0, %0, 4, %4\n v_lshl_add_u32 %1, %1, 4, %4\n v_lshl_add_u32 %2, %2, 4, %4\n v_lshl_add_u32 %3, %3, 4, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 13) asm volatile(R8("s_and_b64 s[20:21], s[22:23], s[24:25]\n s_and_b64 s[26:27], s[22:23], s[24:25]\n s_and_b64 s[20:21], s[22:23], s[24:25]\n s_and_b64 s[26:27], s[22:23], s[24:25]\n") ::: "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        if (KIND == 14) asm volatile(R8("s_nop 1\n s_nop 1\n s_nop 1\n s_nop 1\n"));
        if (KIND == 15) asm volatile(R8("v_mul_f32 %0, %0, %4\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_mul_f32 %1, %1, %4\n s_and_b64 s[26:27], s[22:23], s[24:25]\n v_mul_f32 %2, %2, %4\n s_and_b64 s[20:21], s[22:23], s[24:25]\n v_mul_f32 %3, %3, %4\n s_and_b64 s[26:27], s[22:23], s[24:25]\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "s22", "s23", "s24", "s25", "s26", "s27", "scc");
        if (KIND == 16) {   // one ds_read_b128 at a random 16-byte slot per lane + wait, then 4 dependent v_mul
            float4 v = *(float4*)&lds[((i1 + i) & 255) * 4 + 1024 * ((i >> 3) & 3)];
            x0 = x0 * v.x; x1 = x1 * v.y; x2 = x2 * v.z; x3 = x3 * v.w;
        }
        if (KIND == 19) asm volatile(R8("v_cndmask_b32_e64 %0, %0, %4, s[20:21]\n v_cndmask_b32_e64 %1, %1, %4, s[20:21]\n v_cndmask_b32_e64 %2, %2, %4, s[20:21]\n v_cndmask_b32_e64 %3, %3, %4, s[20:21]\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21");
        if (KIND == 20) asm volatile(R8("v_cmp_lt_f32_e32 vcc, %0, %4\n v_cmp_lt_f32_e32 vcc, %1, %4\n v_cmp_lt_f32_e32 vcc, %2, %4\n v_cmp_lt_f32_e32 vcc, %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 21) asm volatile(R8("v_mul_f32 %0, %0, %4\n v_cndmask_b32_e32 %1, %1, %4, vcc\n v_mul_f32 %2, %2, %4\n v_cndmask_b32_e32 %3, %3, %4, vcc\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 22) asm volatile(R8("v_cndmask_b32_e32 %0, %4, %4, vcc\n v_cndmask_b32_e32 %1, %4, %4, vcc\n v_cndmask_b32_e32 %2, %4, %4, vcc\n v_cndmask_b32_e32 %3, %4, %4, vcc\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "vcc");
        if (KIND == 23) asm volatile(R8("v_sub_f32 %0, %0, %4\n v_add_f32 %1, %1, %4\n v_sub_f32 %2, %2, %4\n v_add_f32 %3, %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 24) asm volatile(R8("v_min_f32 %0, %0, %4\n v_max_f32 %1, %1, %4\n v_min_f32 %2, %2, %4\n v_max_f32 %3, %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 25) asm volatile(R8("v_max3_f32 %0, %0, %4, %1\n v_min3_f32 %1, %1, %4, %2\n v_max3_f32 %2, %2, %4, %3\n v_min3_f32 %3, %3, %4, %0\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 26) asm volatile(R8("v_and_b32 %0, %0, %4\n v_add_u32 %1, %1, %4\n v_lshlrev_b32 %2, 4, %2\n v_bfe_u32 %3, %3, 16, 16\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 27) asm volatile(R8("v_mov_b32 %0, %4\n v_mov_b32 %1, %4\n v_mov_b32 %2, %4\n v_mov_b32 %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 28) asm volatile(R8("v_fma_mix_f32 %0, %4, %0, %0 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %1, %4, %1, %1 op_sel_hi:[1,0,0]\n v_fma_mix_f32 %2, %4, %2, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n v_fma_mix_f32 %3, %4, %3, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 29) { unsigned long long y0 = i0, y1 = i1; asm volatile(R8("v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n v_lshl_add_u64 %0, %0, 0, %2\n v_lshl_add_u64 %1, %1, 0, %2\n") : "+v"(y0), "+v"(y1) : "v"(y0)); x0 += (float)(y0 + y1); }
        if (KIND == 30) asm volatile(R8("v_alignbit_b32 %0, %0, %1, 7\n v_alignbit_b32 %1, %1, %2, 9\n v_alignbit_b32 %2, %2, %3, 11\n v_alignbit_b32 %3, %3, %0, 13\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 31) asm volatile(R8("v_xor_b32 %0, %0, %4\n v_xor_b32 %1, %1, %4\n v_xor_b32 %2, %2, %4\n v_xor_b32 %3, %3, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 32) asm volatile(R8("v_cvt_f32_u32 %0, %0\n v_cvt_f32_u32 %1, %1\n v_cvt_f32_u32 %2, %2\n v_cvt_f32_u32 %3, %3\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        if (KIND == 17) asm volatile(R8("v_mul_f32 %0, %0, %4\n v_mul_f32 %1, %0, %4\n v_mul_f32 %2, %1, %4\n v_mul_f32 %3, %2, %4\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));   // dependent chain
        if (KIND == 18) asm volatile(R8("v_mul_f32 %0, %0, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_mul_f32 %1, %1, %4\n s_or_b64 exec, exec, s[20:21]\n v_mul_f32 %2, %2, %4\n s_and_saveexec_b64 s[20:21], vcc\n v_mul_f32 %3, %3, %4\n s_or_b64 exec, exec, s[20:21]\n") : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a) : "s20", "s21", "scc", "vcc");
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + i0;
}
template <int KIND> void run(const char* name, int per_iter, int waves_per_simd) {
    float* d; hipMalloc(&d, 16 << 20);
    int blocks = 256 * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITER * per_iter;
    printf("%-34s waves/SIMD=%d: %.2f cycles per instruction per SIMD @2.4GHz\n", name, waves_per_simd, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    for (int w : {4}) {
        run<0>("v_mul_f32", 32, w); run<1>("v_fma_f32", 32, w); run<17>("v_mul_f32 dependent chain", 32, w);
        run<2>("v_cmp_lt_f32 -> sgpr pair", 32, w); run<3>("v_cndmask_b32", 32, w);
        run<4>("v_max_i32_dpp row_shr", 32, w); run<5>("v_mov_b32_dpp row_bcast", 32, w); run<6>("v_readlane_b32", 32, w);
        run<7>("ds_bpermute_b32 (4 + wait)", 32, w); run<8>("v_rcp_f32", 32, w); run<9>("v_div_scale_f32", 32, w);
        run<10>("v_div_fmas_f32", 32, w); run<11>("v_div_fixup_f32", 32, w); run<12>("v_lshl_add_u32", 32, w);
        run<13>("s_and_b64", 32, w); run<14>("s_nop 1", 32, w); run<15>("v_mul + s_and interleaved (per pair)", 32, w);
        run<18>("v_mul + exec save/restore (per pair)", 32, w);
        run<19>("v_cndmask_b32_e64 sgpr mask", 32, w); run<20>("v_cmp_lt_f32_e32 vcc", 32, w); run<21>("v_mul / v_cndmask_e32 alternating", 32, w);
        run<22>("v_cndmask_b32_e32 (no RAW chain)", 32, w); run<23>("v_sub/v_add_f32", 32, w); run<24>("v_min/v_max_f32", 32, w);
        run<25>("v_max3/v_min3_f32", 32, w); run<26>("and/add_u32/lshlrev/bfe mix", 32, w);
        run<16>("ds_read_b128 random + 4 v_mul", 1, w);
        run<27>("v_mov_b32", 32, w); run<28>("v_fma_mix_f32 (binary16 source)", 32, w); run<29>("v_lshl_add_u64", 32, w);
        run<30>("v_alignbit_b32", 32, w); run<31>("v_xor_b32", 32, w); run<32>("v_cvt_f32_u32", 32, w);
    }
    for (int w : {1, 2, 8}) {   // how the price of the main classes depends on the waves that share a SIMD
        run<0>("v_mul_f32", 32, w); run<1>("v_fma_f32", 32, w); run<2>("v_cmp_lt_f32 -> sgpr pair", 32, w); run<3>("v_cndmask_b32", 32, w);
        run<24>("v_min/v_max_f32", 32, w); run<26>("and/add_u32/lshlrev/bfe mix", 32, w); run<27>("v_mov_b32", 32, w);
    }
    return 0;
}
