// valu_rate.hip -- gfx950 micro-benchmark: scalar vs packed fp32 VALU issue rate, and
// ds_read_b128 rate for broadcast / distinct-address patterns.  Informs DESIGN.md's op budget.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__device__ unsigned long long g_clk[2];
template <int PK>
__global__ __launch_bounds__(256) void k_valu(float *out, int iters, float a, float b)
{
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7;
    if (PK) {
        v2f p0 = {x0, x1}, p1 = {x2, x3}, p2 = {x4, x5}, p3 = {x6, x7};
        v2f p4 = {x0 + 8, x1 + 8}, p5 = {x2 + 8, x3 + 8}, p6 = {x4 + 8, x5 + 8}, p7 = {x6 + 8, x7 + 8};
        const v2f va = {a, a}, vb = {b, b};
        for (int i = 0; i < iters; i++) {
#define P(p) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p) : "v"(va), "v"(vb));
            P(p0) P(p1) P(p2) P(p3) P(p4) P(p5) P(p6) P(p7)
#undef P
        }
        v2f s = p0 + p1 + p2 + p3 + p4 + p5 + p6 + p7;
        out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
    } else {
        for (int i = 0; i < iters; i++) {
#define S(x) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
            S(x0) S(x1) S(x2) S(x3) S(x4) S(x5) S(x6) S(x7)
#undef S
        }
        out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7;
    }
    if (blockIdx.x == 777 && threadIdx.x == 0) {
        g_clk[0] = __builtin_amdgcn_s_memtime() - t0;
        g_clk[1] = __builtin_amdgcn_s_memrealtime() - r0;
    }
}

// MODE 0: all lanes same address; 1: 4 distinct addresses spread over lanes (stride 272 B);
// 2: 64 random addresses in a 16 KB window
template <int MODE>
__global__ __launch_bounds__(256) void k_lds(float *out, int iters, const int *perm)
{
    extern __shared__ float4 win[];
    for (int i = threadIdx.x; i < 1024; i += 256) win[i] = make_float4(i, i + 1, i + 2, i + 3);
    __syncthreads();
    int idx;
    if (MODE == 0) idx = 5;
    else if (MODE == 1) idx = (threadIdx.x & 3) * 17 + 5;
    else idx = perm[threadIdx.x];
    float4 acc = make_float4(0, 0, 0, 0);
    for (int i = 0; i < iters; i++) {
#pragma unroll
        for (int j = 0; j < 8; j++) {
            const float4 v = win[(idx + j * 16) & 1023];
            acc.x += v.x; acc.y += v.y; acc.z += v.z; acc.w += v.w;
        }
        idx = (idx + 1) & 1023;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

int main()
{
    float *out; hipMalloc(&out, 4096 * 256 * 4);
    int *perm; hipMalloc(&perm, 256 * 4);
    std::vector<int> hp(256);
    unsigned s = 12345; for (int i = 0; i < 256; i++) { s = s * 1664525u + 1013904223u; hp[i] = (s >> 8) & 1023; }
    hipMemcpy(perm, hp.data(), 1024, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 256 * 8, iters = 20000;
    auto run = [&](const char *name, auto launch, double ops_per_lane) {
        launch(); hipDeviceSynchronize();
        hipEventRecord(e0); launch(); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double total = ops_per_lane * blocks * 256.0;
        printf("%-28s %8.3f ms  %8.2f T lane-instr/s  -> per CU per clk @2.4GHz: %.1f\n", name, ms, total / ms / 1e9,
               total / (ms * 1e-3) / 256 / 2.4e9);
    };
    run("v_fma_f32 (8 chains)", [&] { hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); }, 8.0 * iters);
    run("v_pk_fma_f32 (8 chains)", [&] { hipLaunchKernelGGL(k_valu<1>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); }, 8.0 * iters);
    unsigned long long hc[2]; hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16);
    printf("in-kernel clock during v_pk loop: %.3f GHz (memtime %llu / realtime %llu @100MHz)\n", (double)hc[0] / hc[1] * 0.1, hc[0], hc[1]);
    hipLaunchKernelGGL(k_valu<0>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); hipDeviceSynchronize();
    hipMemcpyFromSymbol(hc, HIP_SYMBOL(g_clk), 16);
    printf("in-kernel clock during v_fma loop: %.3f GHz\n", (double)hc[0] / hc[1] * 0.1);
    const int li = 4000;
    run("ds_read_b128 broadcast", [&] { hipLaunchKernelGGL(k_lds<0>, dim3(blocks), dim3(256), 16384, 0, out, li, perm); }, 8.0 * li);
    run("ds_read_b128 4 addrs", [&] { hipLaunchKernelGGL(k_lds<1>, dim3(blocks), dim3(256), 16384, 0, out, li, perm); }, 8.0 * li);
    run("ds_read_b128 random", [&] { hipLaunchKernelGGL(k_lds<2>, dim3(blocks), dim3(256), 16384, 0, out, li, perm); }, 8.0 * li);
    return 0;
}
