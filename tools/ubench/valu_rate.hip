// Microbenchmark: sustained wave64 VALU issue rate on gfx950 for the instruction mix the trace kernel uses.
// Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int KIND>
__global__ void __launch_bounds__(256) k(float* out, float a, float b) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0) {   // v_fma_f32
            x0 = __builtin_fmaf(x0, a, b); x1 = __builtin_fmaf(x1, a, b); x2 = __builtin_fmaf(x2, a, b); x3 = __builtin_fmaf(x3, a, b);
            x4 = __builtin_fmaf(x4, a, b); x5 = __builtin_fmaf(x5, a, b); x6 = __builtin_fmaf(x6, a, b); x7 = __builtin_fmaf(x7, a, b);
        } else if (KIND == 1) {   // v_mul_f32 (no contraction)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
        } else if (KIND == 2) {   // v_cmp + v_cndmask pairs
            asm volatile("v_cmp_le_f32 vcc, %0, %8\n v_cndmask_b32 %0, %0, %8, vcc\n v_cmp_le_f32 vcc, %1, %8\n v_cndmask_b32 %1, %1, %8, vcc\n"
                         "v_cmp_le_f32 vcc, %2, %8\n v_cndmask_b32 %2, %2, %8, vcc\n v_cmp_le_f32 vcc, %3, %8\n v_cndmask_b32 %3, %3, %8, vcc\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a) : "vcc");
        } else {   // v_pk_mul_f32 (2 floats per lane per instruction)
            asm volatile("v_pk_mul_f32 %0, %0, %4\n v_pk_mul_f32 %1, %1, %4\n v_pk_mul_f32 %2, %2, %4\n v_pk_mul_f32 %3, %3, %4\n"
                         : "+v"(*(double*)&x0), "+v"(*(double*)&x2), "+v"(*(double*)&x4), "+v"(*(double*)&x6) : "v"(*(double*)&a));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int KIND> void run(const char* name, int per_iter, int waves_per_simd) {
    int dev_cus = 256; float* d; hipMalloc(&d, 4 << 20);
    int blocks = dev_cus * waves_per_simd;   // 256 threads = 4 waves = 1 per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double wave_instr = (double)blocks * 4 * N_ITER * per_iter;
    printf("%-22s waves/SIMD=%d: %.3f ms, %.3e wave-instr/s chip, %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, waves_per_simd, ms,
           wave_instr / (ms * 1e-3), 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    for (int w : {1, 2, 4, 8}) { run<0>("v_fma_f32", 8, w); run<1>("v_mul_f32", 8, w); run<2>("v_cmp+v_cndmask (x2)", 8, w); run<3>("v_pk_mul_f32", 4, w); }
    return 0;
}
