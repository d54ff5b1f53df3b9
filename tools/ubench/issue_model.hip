// issue_model.hip -- how many waves per SIMD and how much instruction-level parallelism per wave does gfx950 need to
// keep its VALU issue port busy?  Chains of dependent v_fma_f32 (or an fma / cvt mix), C independent chains per wave,
// W waves per SIMD (occupancy capped through dynamic LDS).  Prints lane-instructions per second.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int C, int MIX>
__global__ __launch_bounds__(256) void k(float *out, int iters, float a, float b)
{
    extern __shared__ float sm[];
    float x[C];
#pragma unroll
    for (int c = 0; c < C; c++) x[c] = threadIdx.x + c;
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int r = 0; r < 16 / C; r++) {
#pragma unroll
            for (int c = 0; c < C; c++) {
                if (MIX && (r & 1)) asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(x[c]));
                else asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[c]) : "v"(a), "v"(b));
            }
        }
    }
    float s = 0;
#pragma unroll
    for (int c = 0; c < C; c++) s += x[c];
    out[blockIdx.x * 256 + threadIdx.x] = s + (threadIdx.x == 9999 ? sm[0] : 0.f);
}
template <int C, int MIX> void run(float *out, int wps)
{
    // one 256-thread block = 4 waves = 1 wave per SIMD; `wps` blocks per CU via LDS: 160 KB / wps
    const size_t lds = wps >= 8 ? 160 * 1024 / 8 : 160 * 1024 / wps - 1024;
    hipFuncSetAttribute((const void *)k<C, MIX>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    const int blocks = 256 * wps, iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipLaunchKernelGGL((k<C, MIX>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipEventRecord(e0); hipLaunchKernelGGL((k<C, MIX>), dim3(blocks), dim3(256), lds, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("chains %2d mix %d waves/SIMD %d : %6.2f T lane-instr/s\n", C, MIX, wps, 16.0 * iters * blocks * 256.0 / ms / 1e9);
}
int main()
{
    float *out; hipMalloc(&out, 256 * 8 * 256 * 4);
    for (int w : {1, 2, 3, 4, 5, 6, 8}) { run<1, 0>(out, w); run<2, 0>(out, w); run<4, 0>(out, w); run<8, 0>(out, w); run<4, 1>(out, w); run<8, 1>(out, w); }
    return 0;
}
