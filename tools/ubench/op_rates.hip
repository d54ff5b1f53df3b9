// op_rates.hip -- per-opcode VALU issue rate on gfx950 (8 waves/SIMD, 8 independent regs per wave).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP8(X) X(x0) X(x1) X(x2) X(x3) X(x4) X(x5) X(x6) X(x7)
#define KERNEL(NAME, ASMSTR) \
__global__ __launch_bounds__(256) void k_##NAME(float *out, int iters, float a, float b) { \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    float sa = __builtin_amdgcn_readfirstlane(__float_as_int(a)), sb = b; (void)sa; (void)sb; \
    unsigned long long m = 0x5555555555555555ull; (void)m; \
    for (int i = 0; i < iters; i++) { REP8(ASMSTR) } \
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }
#define A_FMA(x)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MUL(x)    asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MULS(x)   asm volatile("v_mul_f32_e32 %0, %1, %0" : "+v"(x) : "s"(a));
#define A_ADD(x)    asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MED3(x)   asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MED3S(x)  asm volatile("v_med3_f32 %0, %0, %1, 0" : "+v"(x) : "s"(a));
#define A_FLOOR(x)  asm volatile("v_floor_f32_e32 %0, %0" : "+v"(x));
#define A_TRUNC(x)  asm volatile("v_trunc_f32_e32 %0, %0" : "+v"(x));
#define A_MAX3(x)   asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_CVTU(x)   asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(x));
#define A_CVTF(x)   asm volatile("v_cvt_f32_u32_e32 %0, %0" : "+v"(x));
#define A_SDWA(x)   asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_0" : "+v"(x));
#define A_CMPCND(x) asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1\n\tv_cndmask_b32_e64 %0, %0, %2, s[20:21]" : "+v"(x) : "v"(a), "v"(b) : "s20", "s21");
#define A_CND(x)    asm volatile("v_cndmask_b32_e64 %0, %0, %1, %2" : "+v"(x) : "v"(a), "s"(m));
#define A_CMP(x)    asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(x), "v"(a) : "s20", "s21");
#define A_ADDU(x)   asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MINU(x)   asm volatile("v_min_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_LSHLOR(x) asm volatile("v_lshl_or_b32 %0, %0, 16, %1" : "+v"(x) : "v"(a));
#define A_SUB(x)    asm volatile("v_sub_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_FMAC(x)   asm volatile("v_fmac_f32_e32 %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MOV(x)    asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(a));
KERNEL(fma, A_FMA) KERNEL(mul, A_MUL) KERNEL(mul_s, A_MULS) KERNEL(add, A_ADD) KERNEL(sub, A_SUB) KERNEL(fmac, A_FMAC)
KERNEL(med3, A_MED3) KERNEL(med3_s, A_MED3S) KERNEL(floor, A_FLOOR) KERNEL(trunc, A_TRUNC) KERNEL(max3, A_MAX3)
KERNEL(cvt_u32, A_CVTU) KERNEL(cvt_f32, A_CVTF) KERNEL(sdwa, A_SDWA) KERNEL(cmp_cnd, A_CMPCND) KERNEL(cnd, A_CND) KERNEL(cmp, A_CMP)
KERNEL(add_u32, A_ADDU) KERNEL(min_u32, A_MINU) KERNEL(lshl_or, A_LSHLOR) KERNEL(mov, A_MOV)
int main() {
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, iters = 10000;
#define RUN(NAME, PER) { hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); hipDeviceSynchronize(); \
    hipEventRecord(e0); hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, out, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); printf("%-10s %7.3f ms  %6.2f T lane-instr/s\n", #NAME, ms, PER * 8.0 * iters * blocks * 256.0 / ms / 1e9); }
    RUN(fma,1) RUN(mul,1) RUN(mul_s,1) RUN(add,1) RUN(sub,1) RUN(fmac,1) RUN(med3,1) RUN(med3_s,1) RUN(floor,1) RUN(trunc,1) RUN(max3,1)
    RUN(cvt_u32,1) RUN(cvt_f32,1) RUN(sdwa,1) RUN(cmp_cnd,2) RUN(cnd,1) RUN(cmp,1) RUN(add_u32,1) RUN(min_u32,1) RUN(lshl_or,1) RUN(mov,1)
    return 0;
}
