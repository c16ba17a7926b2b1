// Microbenchmark: issue rate of the integer instructions Threefish-256 is made of on gfx950
// (64-bit add as one v_lshl_add_u64 or as a v_add_co / v_addc_co pair, v_alignbit_b32, v_xor_b32).
// Build: hipcc --offload-arch=gfx950 -O3 -o int_rate int_rate.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int KIND>
__global__ void __launch_bounds__(256) k(unsigned long long* out, unsigned long long a) {
    unsigned long long x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3;
    unsigned int y0 = threadIdx.x, y1 = y0 + 1, y2 = y0 + 2, y3 = y0 + 3, y4 = y0 + 4, y5 = y0 + 5, y6 = y0 + 6, y7 = y0 + 7;
    for (int i = 0; i < N_ITER; ++i) {
        if (KIND == 0) {          // 4 x v_lshl_add_u64
            asm volatile("v_lshl_add_u64 %0, %0, 0, %4\n v_lshl_add_u64 %1, %1, 0, %4\n v_lshl_add_u64 %2, %2, 0, %4\n v_lshl_add_u64 %3, %3, 0, %4\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3) : "v"(a));
        } else if (KIND == 1) {   // 4 x (v_add_co_u32 + v_addc_co_u32)
            asm volatile("v_add_co_u32 %0, vcc, %0, %8\n v_addc_co_u32 %1, vcc, %1, %8, vcc\n v_add_co_u32 %2, vcc, %2, %8\n v_addc_co_u32 %3, vcc, %3, %8, vcc\n"
                         "v_add_co_u32 %4, vcc, %4, %8\n v_addc_co_u32 %5, vcc, %5, %8, vcc\n v_add_co_u32 %6, vcc, %6, %8\n v_addc_co_u32 %7, vcc, %7, %8, vcc\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"((unsigned int)a) : "vcc");
        } else if (KIND == 2) {   // 8 x v_alignbit_b32
            asm volatile("v_alignbit_b32 %0, %0, %1, 13\n v_alignbit_b32 %1, %1, %2, 13\n v_alignbit_b32 %2, %2, %3, 13\n v_alignbit_b32 %3, %3, %4, 13\n"
                         "v_alignbit_b32 %4, %4, %5, 13\n v_alignbit_b32 %5, %5, %6, 13\n v_alignbit_b32 %6, %6, %7, 13\n v_alignbit_b32 %7, %7, %0, 13\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7));
        } else {                  // 8 x v_xor_b32
            asm volatile("v_xor_b32 %0, %0, %8\n v_xor_b32 %1, %1, %8\n v_xor_b32 %2, %2, %8\n v_xor_b32 %3, %3, %8\n"
                         "v_xor_b32 %4, %4, %8\n v_xor_b32 %5, %5, %8\n v_xor_b32 %6, %6, %8\n v_xor_b32 %7, %7, %8\n"
                         : "+v"(y0), "+v"(y1), "+v"(y2), "+v"(y3), "+v"(y4), "+v"(y5), "+v"(y6), "+v"(y7) : "v"((unsigned int)a));
        }
    }
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + y0 + y1 + y2 + y3 + y4 + y5 + y6 + y7;
}
template <int KIND> void run(const char* name, int per_iter) {
    unsigned long long* d; (void)hipMalloc(&d, 8 << 20);
    const int blocks = 256 * 8;
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull); (void)hipDeviceSynchronize();
    (void)hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d, 0x9E3779B97F4A7C15ull); (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * N_ITER * per_iter;
    printf("%-36s %.3f ms, %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, ms, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
    (void)hipFree(d);
}
int main() {
    run<0>("v_lshl_add_u64 (one 64-bit add)", 4); run<1>("v_add_co_u32 + v_addc_co_u32 (x1 each)", 8);
    run<2>("v_alignbit_b32", 8); run<3>("v_xor_b32", 8);
    return 0;
}
