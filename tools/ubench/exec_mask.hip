// Microbenchmark: does a wave64 VALU instruction on gfx950 issue faster when part of the wave is masked off?
// (If halves or quarters with EXEC == 0 were skipped, packing a wave's busy lanes together would pay.)
// Build: hipcc --offload-arch=gfx950 -O3 -o exec_mask exec_mask.hip ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int PATTERN>
__global__ void __launch_bounds__(256) k(float* out, float a) {
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    const int lane = threadIdx.x & 63;
    bool on;
    if (PATTERN == 0) on = true;                 // 64 lanes
    else if (PATTERN == 1) on = lane < 32;       // lower half
    else if (PATTERN == 2) on = lane < 16;       // lower quarter
    else if (PATTERN == 3) on = (lane & 1) == 0; // every other lane (32 lanes, spread)
    else on = lane == 0;                         // one lane
    if (on)
        for (int i = 0; i < N_ITER; ++i)
            asm volatile("v_mul_f32 %0, %0, %8\n v_mul_f32 %1, %1, %8\n v_mul_f32 %2, %2, %8\n v_mul_f32 %3, %3, %8\n"
                         "v_mul_f32 %4, %4, %8\n v_mul_f32 %5, %5, %8\n v_mul_f32 %6, %6, %8\n v_mul_f32 %7, %7, %8\n"
                         : "+v"(x0), "+v"(x1), "+v"(x2), "+v"(x3), "+v"(x4), "+v"(x5), "+v"(x6), "+v"(x7) : "v"(a));
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
}
template <int PATTERN> void run(const char* name, int waves_per_simd) {
    float* d; hipMalloc(&d, 4 << 20);
    const int blocks = 256 * waves_per_simd;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<PATTERN>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<PATTERN>, dim3(blocks), dim3(256), 0, 0, d, 1.0001f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double wave_instr = (double)blocks * 4 * N_ITER * 8;
    printf("%-28s waves/SIMD=%d: %.3f ms, %.2f cycles per wave-instr per SIMD @2.4GHz\n", name, waves_per_simd, ms, 1024.0 * 2.4e9 / (wave_instr / (ms * 1e-3)));
    hipFree(d);
}
int main() {
    for (int w : {1, 4}) {
        run<0>("v_mul_f32, 64 lanes", w); run<1>("v_mul_f32, lanes 0-31", w); run<2>("v_mul_f32, lanes 0-15", w);
        run<3>("v_mul_f32, even lanes", w); run<4>("v_mul_f32, lane 0", w);
    }
    return 0;
}
