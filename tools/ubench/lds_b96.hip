// Microbenchmark: can gfx950 read 12-byte records from LDS at 4-byte alignment with ONE ds_read_b96, and what does it cost
// against ds_read_b128 of 16-byte records and against three ds_read_b32?  Random record per lane (the vertex-table pattern).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_b96 lds_b96.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#define N_ITER 4096
template <int KIND>
__global__ void __launch_bounds__(1024) k(float* out, int* bad) {
    extern __shared__ float lds[];
    const int n = 3200;
    for (int i = threadIdx.x; i < n * 4; i += 1024) lds[i] = (float)i;     // element i = i, so a record's content says where it came from
    __syncthreads();
    unsigned idx = (threadIdx.x * 2654435761u) % n;
    float a0 = 0, a1 = 0, a2 = 0; int wrong = 0;
    for (int i = 0; i < N_ITER; ++i) {
        float x, y, z;
        float4 q;
        if (KIND == 0) { unsigned addr = idx * 16; asm volatile("ds_read_b128 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(q) : "v"(addr)); x = q.x; y = q.y; z = q.z; wrong += (x != (float)(idx * 4)); }
        if (KIND == 1) { unsigned addr = idx * 12; typedef float f3v __attribute__((ext_vector_type(3))); f3v r; asm volatile("ds_read_b96 %0, %1\n s_waitcnt lgkmcnt(0)" : "=v"(r) : "v"(addr)); x = r.x; y = r.y; z = r.z; wrong += (x != (float)(idx * 3)) + (z != (float)(idx * 3 + 2)); }
        if (KIND == 2) { unsigned addr = idx * 12; asm volatile("ds_read_b32 %0, %3\n ds_read_b32 %1, %3 offset:4\n ds_read_b32 %2, %3 offset:8\n s_waitcnt lgkmcnt(0)" : "=v"(x), "=v"(y), "=v"(z) : "v"(addr)); wrong += (x != (float)(idx * 3)); }
        a0 += x; a1 += y; a2 += z;
        idx = (idx * 1664525u + 1013904223u + (unsigned)a0) % n;
    }
    out[blockIdx.x * 1024 + threadIdx.x] = a0 + a1 + a2;
    if (wrong) atomicAdd(bad, wrong);
}
template <int KIND> void run(const char* name) {
    float* d; int* bad; hipMalloc(&d, 4 << 20); hipMalloc(&bad, 4); hipMemset(bad, 0, 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 3200 * 16, 0, d, bad); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL(k<KIND>, dim3(256), dim3(1024), 3200 * 16, 0, d, bad); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); int h = 0; hipMemcpy(&h, bad, 4, hipMemcpyDeviceToHost);
    printf("%-28s %.3f ms, %.1f cycles per wave-read per CU @2.4GHz, wrong values: %d\n", name, ms, ms * 1e-3 * 2.4e9 / (16.0 * N_ITER), h);
}
int main() { run<0>("ds_read_b128 (16 B records)"); run<1>("ds_read_b96 (12 B, align 4)"); run<2>("3 x ds_read_b32 (12 B)"); return 0; }
