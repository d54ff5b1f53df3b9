// op_rates2.hip -- round 2: issue cost of more gfx950 VALU opcodes, mixes of the two issue classes, and two
// hardware facts the tile kernels lean on (v_fma_mix_f32 with an f16 operand, LDS reads past the allocation).
// 8 waves/SIMD, 8 independent registers per wave; reports lane-instructions per second and shader cycles per
// wave-instruction per SIMD (s_memtime over one wave's loop / instructions issued by the SIMD's 8 waves).
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
#include <cstring>
#include <vector>
#define REP8(X) X(x0) X(x1) X(x2) X(x3) X(x4) X(x5) X(x6) X(x7)
#define KERNEL(NAME, BODY) \
__global__ __launch_bounds__(256) void k_##NAME(float *out, unsigned long long *clk, int iters, float a, float b) { \
    float x0 = threadIdx.x, x1 = x0 + 1, x2 = x0 + 2, x3 = x0 + 3, x4 = x0 + 4, x5 = x0 + 5, x6 = x0 + 6, x7 = x0 + 7; \
    unsigned long long m = 0x5555555555555555ull; (void)m; \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime(); \
    for (int i = 0; i < iters; i++) { BODY } \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime(); \
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0; \
    out[blockIdx.x * 256 + threadIdx.x] = x0 + x1 + x2 + x3 + x4 + x5 + x6 + x7; }

#define A_FMA(x)     asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_FMAS(x)    asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x) : "s"(a), "v"(b));
#define A_FMAK(x)    asm volatile("v_fma_f32 %0, %0, 0.5, %1" : "+v"(x) : "v"(b));
#define A_MULK(x)    asm volatile("v_mul_f32_e32 %0, 0.5, %0" : "+v"(x));
#define A_MULL(x)    asm volatile("v_mul_f32_e32 %0, 0x3f800347, %0" : "+v"(x));
#define A_ADD(x)     asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_ADDS(x)    asm volatile("v_add_f32_e32 %0, %1, %0" : "+v"(x) : "s"(a));
#define A_MINF(x)    asm volatile("v_min_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MAXF(x)    asm volatile("v_max_f32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MIN3F(x)   asm volatile("v_min3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_FMAMIX(x)  asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel_hi:[0,1,0]" : "+v"(x) : "v"(a), "v"(b));
#define A_FMAMIXH(x) asm volatile("v_fma_mix_f32 %0, %0, %1, %2 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "+v"(x) : "v"(a), "v"(b));
#define A_PKFMA(x)   asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(p##x) : "v"(pa), "v"(pb));
#define A_AND(x)     asm volatile("v_and_b32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_OR(x)      asm volatile("v_or_b32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_LSHL(x)    asm volatile("v_lshlrev_b32_e32 %0, 3, %0" : "+v"(x));
#define A_LSHR(x)    asm volatile("v_lshrrev_b32_e32 %0, 16, %0" : "+v"(x));
#define A_BFE(x)     asm volatile("v_bfe_u32 %0, %0, 16, 10" : "+v"(x));
#define A_LSHLADD(x) asm volatile("v_lshl_add_u32 %0, %0, 3, %1" : "+v"(x) : "v"(a));
#define A_ADD3(x)    asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_ANDOR(x)   asm volatile("v_and_or_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MADU24(x)  asm volatile("v_mad_u32_u24 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_MULU24(x)  asm volatile("v_mul_u32_u24_e32 %0, 8, %0" : "+v"(x));
#define A_UBYTE(x)   asm volatile("v_cvt_f32_ubyte1_e32 %0, %0" : "+v"(x));
#define A_PERM(x)    asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_PKMINU(x)  asm volatile("v_pk_min_u16 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_PKMAXU(x)  asm volatile("v_pk_max_u16 %0, %0, %1" : "+v"(x) : "v"(a));
#define A_FRACT(x)   asm volatile("v_fract_f32_e32 %0, %0" : "+v"(x));
#define A_RNDNE(x)   asm volatile("v_rndne_f32_e32 %0, %0" : "+v"(x));
#define A_CNDVCC(x)  asm volatile("v_cndmask_b32_e32 %0, %0, %1, vcc" : "+v"(x) : "v"(a) : "vcc");
#define A_SUBU(x)    asm volatile("v_sub_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_CVTH(x)    asm volatile("v_cvt_f16_f32_e32 %0, %0" : "+v"(x));
#define A_CVTFH(x)   asm volatile("v_cvt_f32_f16_e32 %0, %0" : "+v"(x));
#define A_DOT2(x)    asm volatile("v_dot2_f32_f16 %0, %1, %2, %0" : "+v"(x) : "v"(a), "v"(b));
#define A_SDWAADD(x) asm volatile("v_add_u32_sdwa %0, %1, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1 src1_sel:DWORD" : "+v"(x) : "v"(a));
#define A_CVTSDWA1(x) asm volatile("v_cvt_f32_u32_sdwa %0, %0 dst_sel:DWORD dst_unused:UNUSED_PAD src0_sel:WORD_1" : "+v"(x));
#define A_CVTU(x)    asm volatile("v_cvt_u32_f32_e32 %0, %0" : "+v"(x));
#define A_TRUNC(x)   asm volatile("v_trunc_f32_e32 %0, %0" : "+v"(x));
#define A_MED3(x)    asm volatile("v_med3_f32 %0, %0, %1, %2" : "+v"(x) : "v"(a), "v"(b));
#define A_CMP(x)     asm volatile("v_cmp_gt_f32_e64 s[20:21], %0, %1" : : "v"(x), "v"(a) : "s20", "s21");
#define A_CMPVCC(x)  asm volatile("v_cmp_gt_f32_e32 vcc, %0, %1" : : "v"(x), "v"(a) : "vcc");
#define A_MOV(x)     asm volatile("v_mov_b32_e32 %0, %1" : "=v"(x) : "v"(a));
#define A_MULCLAMP(x) asm volatile("v_mul_f32_e64 %0, %0, %1 clamp" : "+v"(x) : "v"(a));
#define A_FMACLAMP(x) asm volatile("v_fma_f32 %0, %0, %1, %2 clamp" : "+v"(x) : "v"(a), "v"(b));
#define A_MULNEG(x)  asm volatile("v_mul_f32_e64 %0, -%0, |%1|" : "+v"(x) : "v"(a));
#define A_RDLANE(x)  asm volatile("v_readlane_b32 s20, %0, 3" : : "v"(x) : "s20");
#define A_WRLANE(x)  asm volatile("v_writelane_b32 %0, s20, 3" : "+v"(x) : : );
// mixes: is a slow-class op hidden behind fast-class ones (two pipes) or do the costs add?
#define A_MIX_FMA_CVT(x)   A_FMA(x) A_CVTU(x)
#define A_MIX_2FMA_CVT(x)  A_FMA(x) A_FMA(x) A_CVTU(x)
#define A_MIX_FMA_CND(x)   A_FMA(x) A_CNDVCC(x)
#define A_MIX_FMA_MED3(x)  A_FMA(x) A_MED3(x)
#define A_MIX_CVT_MED3(x)  A_CVTU(x) A_MED3(x)
#define A_MIX_FMA_LSHL(x)  A_FMA(x) A_LSHL(x)
#define A_ADDU(x)    asm volatile("v_add_u32_e32 %0, %1, %0" : "+v"(x) : "v"(a));
#define A_MIX_CVT_AND(x)   A_CVTU(x) A_AND(x)
#define A_MIX_CVT_LSHL(x)  A_CVTU(x) A_LSHL(x)
#define A_MIX_CVT_ADDU(x)  A_CVTU(x) A_ADDU(x)
#define A_MIX_CVT_LSHR(x)  A_CVTU(x) A_LSHR(x)
#define A_MIX_MIXH_FMA(x)  A_FMAMIX(x) A_FMA(x)
#define A_MIX_MIXH_AND(x)  A_FMAMIX(x) A_AND(x)
#define A_MIX_FMA_AND(x)   A_FMA(x) A_AND(x)
#define A_MIX_FMA_FMAS(x)  A_FMA(x) A_FMAS(x)
#define A_MIX_CVT_FMAS(x)  A_CVTU(x) A_FMAS(x)

#define K1(NAME, OP) KERNEL(NAME, REP8(OP))
K1(fma, A_FMA) K1(fma_s, A_FMAS) K1(fma_k, A_FMAK) K1(mul_k, A_MULK) K1(mul_lit, A_MULL) K1(add, A_ADD) K1(add_s, A_ADDS)
K1(min_f32, A_MINF) K1(max_f32, A_MAXF) K1(min3_f32, A_MIN3F) K1(fma_mix_lo, A_FMAMIX) K1(fma_mix_hi, A_FMAMIXH)
K1(and_b32, A_AND) K1(or_b32, A_OR) K1(lshl, A_LSHL) K1(lshr, A_LSHR) K1(bfe, A_BFE) K1(lshl_add, A_LSHLADD) K1(add3, A_ADD3)
K1(and_or, A_ANDOR) K1(mad_u24, A_MADU24) K1(mul_u24, A_MULU24) K1(cvt_ubyte, A_UBYTE) K1(perm, A_PERM)
K1(pk_min_u16, A_PKMINU) K1(pk_max_u16, A_PKMAXU) K1(fract, A_FRACT) K1(rndne, A_RNDNE) K1(cnd_vcc, A_CNDVCC) K1(sub_u32, A_SUBU)
K1(cvt_f16, A_CVTH) K1(cvt_f32_f16, A_CVTFH) K1(dot2_f16, A_DOT2) K1(add_sdwa, A_SDWAADD) K1(cvt_sdwa_w1, A_CVTSDWA1)
K1(cvt_u32, A_CVTU) K1(trunc, A_TRUNC) K1(med3, A_MED3) K1(cmp_sgpr, A_CMP) K1(cmp_vcc, A_CMPVCC) K1(mov, A_MOV)
K1(readlane, A_RDLANE) K1(writelane, A_WRLANE) K1(mul_clamp, A_MULCLAMP) K1(fma_clamp, A_FMACLAMP) K1(mul_negabs, A_MULNEG)
K1(mix_fma_cvt, A_MIX_FMA_CVT) K1(mix_2fma_cvt, A_MIX_2FMA_CVT) K1(mix_fma_cnd, A_MIX_FMA_CND) K1(mix_fma_med3, A_MIX_FMA_MED3)
K1(mix_cvt_med3, A_MIX_CVT_MED3) K1(mix_fma_lshl, A_MIX_FMA_LSHL)
K1(add_u32, A_ADDU) K1(mix_cvt_and, A_MIX_CVT_AND) K1(mix_cvt_lshl, A_MIX_CVT_LSHL) K1(mix_cvt_addu, A_MIX_CVT_ADDU) K1(mix_cvt_lshr, A_MIX_CVT_LSHR)
K1(mix_fmamix_fma, A_MIX_MIXH_FMA) K1(mix_fmamix_and, A_MIX_MIXH_AND) K1(mix_fma_and, A_MIX_FMA_AND) K1(mix_fma_fmas, A_MIX_FMA_FMAS) K1(mix_cvt_fmas, A_MIX_CVT_FMAS)

// packed fp32 needs register pairs
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ __launch_bounds__(256) void k_pk_fma(float *out, unsigned long long *clk, int iters, float a, float b)
{
    f2 px0 = {(float)threadIdx.x, 1.f}, px1 = px0 + 1.f, px2 = px0 + 2.f, px3 = px0 + 3.f, px4 = px0 + 4.f, px5 = px0 + 5.f,
       px6 = px0 + 6.f, px7 = px0 + 7.f;
    const f2 pa = {a, a}, pb = {b, b};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; i++) { REP8(A_PKFMA) }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;
    const f2 s = px0 + px1 + px2 + px3 + px4 + px5 + px6 + px7;
    out[blockIdx.x * 256 + threadIdx.x] = s.x + s.y;
}

// ---- fact 1: v_fma_mix_f32 with an f16 operand == fmaf(a, (float)h, c) bit for bit
__global__ void k_mixcheck(const float *a, const __half *h, const float *c, float *o, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned hw = (unsigned)__half_as_ushort(h[i]) | ((unsigned)__half_as_ushort(h[(i + 1) % n]) << 16);
    float lo, hi;
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel_hi:[0,1,0]" : "=v"(lo) : "v"(a[i]), "v"(hw), "v"(c[i]));
    asm volatile("v_fma_mix_f32 %0, %1, %2, %3 op_sel:[0,1,0] op_sel_hi:[0,1,0]" : "=v"(hi) : "v"(a[i]), "v"(hw), "v"(c[i]));
    o[2 * i] = lo; o[2 * i + 1] = hi;
}

// ---- fact 2: what does a ds_read past the workgroup's LDS allocation return?
extern __shared__ unsigned dyn_lds[];
__global__ void k_ldsoob(unsigned *o, int words)
{
    for (int i = threadIdx.x; i < words; i += blockDim.x) dyn_lds[i] = 0xabcd0000u + i;
    __syncthreads();
    const unsigned probes[8] = {0u, (unsigned)(words - 1) * 4u, (unsigned)words * 4u, (unsigned)words * 4u + 4096u, 65532u, 163836u, 163840u, 0xfffffff0u};
    if (threadIdx.x < 8) {
        unsigned v;
        const unsigned addr = probes[threadIdx.x] + (unsigned)(uintptr_t)(__attribute__((address_space(3))) char *)dyn_lds;
        asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
        o[threadIdx.x] = v;
        o[8 + threadIdx.x] = addr;
    }
}

int main()
{
    float *out; hipMalloc(&out, 2048 * 256 * 4);
    unsigned long long *clk; hipMalloc(&clk, 8);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int blocks = 2048, iters = 10000;
    // 2048 blocks x 4 waves over 256 CUs x 4 SIMDs = 8 waves per SIMD, all resident at once
#define RUN(NAME, PER) { hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f); hipDeviceSynchronize(); \
    hipEventRecord(e0); hipLaunchKernelGGL(k_##NAME, dim3(blocks), dim3(256), 0, 0, out, clk, iters, 1.0001f, 0.5f); hipEventRecord(e1); hipEventSynchronize(e1); \
    float ms; hipEventElapsedTime(&ms, e0, e1); unsigned long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost); \
    printf("%-14s %7.3f ms  %6.2f T lane-instr/s   %5.2f cyc/wave-instr/SIMD   (clock %.2f GHz)\n", #NAME, ms, PER * 8.0 * iters * blocks * 256.0 / ms / 1e9, \
           (double)c / (PER * 8.0 * iters * 8.0), (double)c / (ms * 1e6)); }
    RUN(fma,1) RUN(fma_s,1) RUN(fma_k,1) RUN(mul_k,1) RUN(mul_lit,1) RUN(add,1) RUN(add_s,1) RUN(min_f32,1) RUN(max_f32,1) RUN(min3_f32,1)
    RUN(fma_mix_lo,1) RUN(fma_mix_hi,1) RUN(pk_fma,2) RUN(and_b32,1) RUN(or_b32,1) RUN(lshl,1) RUN(lshr,1) RUN(bfe,1) RUN(lshl_add,1) RUN(add3,1)
    RUN(and_or,1) RUN(mad_u24,1) RUN(mul_u24,1) RUN(cvt_ubyte,1) RUN(perm,1) RUN(pk_min_u16,1) RUN(pk_max_u16,1) RUN(fract,1) RUN(rndne,1)
    RUN(cnd_vcc,1) RUN(sub_u32,1) RUN(cvt_f16,1) RUN(cvt_f32_f16,1) RUN(dot2_f16,1) RUN(add_sdwa,1) RUN(cvt_sdwa_w1,1) RUN(cvt_u32,1) RUN(trunc,1)
    RUN(med3,1) RUN(cmp_sgpr,1) RUN(cmp_vcc,1) RUN(mov,1) RUN(readlane,1) RUN(writelane,1) RUN(mul_clamp,1) RUN(fma_clamp,1) RUN(mul_negabs,1)
    RUN(mix_fma_cvt,2) RUN(mix_2fma_cvt,3) RUN(mix_fma_cnd,2) RUN(mix_fma_med3,2) RUN(mix_cvt_med3,2) RUN(mix_fma_lshl,2)
    RUN(add_u32,1) RUN(mix_cvt_and,2) RUN(mix_cvt_lshl,2) RUN(mix_cvt_addu,2) RUN(mix_cvt_lshr,2) RUN(mix_fmamix_fma,2) RUN(mix_fmamix_and,2)
    RUN(mix_fma_and,2) RUN(mix_fma_fmas,2) RUN(mix_cvt_fmas,2)

    {   // fma_mix exactness
        const int n = 1 << 16;
        std::vector<float> a(n), c(n), o(2 * n);
        std::vector<__half> h(n);
        unsigned s = 12345u;
        auto rnd = [&]() { s = s * 1664525u + 1013904223u; return (float)(s >> 8) / 16777216.0f; };
        for (int i = 0; i < n; i++) { a[i] = rnd(); c[i] = rnd() * 3.0f - 1.0f; h[i] = __float2half(rnd() * 1.5f - 0.25f); }
        float *da, *dc, *dout; __half *dh;
        hipMalloc(&da, n * 4); hipMalloc(&dc, n * 4); hipMalloc(&dout, 2 * n * 4); hipMalloc(&dh, n * 2);
        hipMemcpy(da, a.data(), n * 4, hipMemcpyHostToDevice); hipMemcpy(dc, c.data(), n * 4, hipMemcpyHostToDevice);
        hipMemcpy(dh, h.data(), n * 2, hipMemcpyHostToDevice);
        hipLaunchKernelGGL(k_mixcheck, dim3(n / 256), dim3(256), 0, 0, da, dh, dc, dout, n);
        hipMemcpy(o.data(), dout, 2 * n * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < n; i++) {
            const float lo = __builtin_fmaf(a[i], __half2float(h[i]), c[i]), hi = __builtin_fmaf(a[i], __half2float(h[(i + 1) % n]), c[i]);
            if (memcmp(&lo, &o[2 * i], 4) || memcmp(&hi, &o[2 * i + 1], 4)) bad++;
        }
        printf("v_fma_mix_f32 vs fmaf(a, (float)half, c): %d of %d differ\n", bad, 2 * n);
    }
    {   // LDS out-of-allocation reads
        unsigned *d; hipMalloc(&d, 64);
        for (int words : {1024, 16384}) {
            hipMemset(d, 0xff, 64);
            hipLaunchKernelGGL(k_ldsoob, dim3(1), dim3(256), words * 4, 0, d, words);
            unsigned h[16];
            hipError_t e = hipMemcpy(h, d, 64, hipMemcpyDeviceToHost);
            printf("LDS alloc %d B (%s): ", words * 4, hipGetErrorString(e));
            for (int i = 0; i < 8; i++) printf("[%u]=%08x ", h[8 + i], h[i]);
            printf("\n");
        }
    }
    return 0;
}
