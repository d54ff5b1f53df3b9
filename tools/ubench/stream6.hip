// stream6.hip -- what does the memory system give a kernel that reads three planes and writes three planes?
//
// The planar-RGB LUT kernel (gbrp10le: 6 B in + 6 B out per pixel) is memory-side: its body-less skeleton measured 5.3 TB/s in
// round 2 against the 6.29 TB/s a plain float4 copy reaches on this chip (MI355X_MICROARCH.md).  This benchmark separates the
// access pattern from the kernel: same bytes, same number of streams, no LUT.
//
//   flat   : non-persistent, thread i copies 16 B at offset i of each plane (U such triples in flight per thread)
//   tile   : the product's shape -- persistent waves (16 per CU), chunks of `ch` tiles handed out by one atomic counter, a tile =
//            (1 << lw) lanes across x (16 B each) by 64 >> lw rows, next tile's loads issued before this tile's stores
//   variants: non-temporal loads / stores, workgroup size, tile shape, chunk height, plane-staggered start
//
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench/stream6.hip -o tools/ubench/stream6
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstdint>
#include <vector>
#include <functional>

typedef unsigned v4 __attribute__((ext_vector_type(4)));

template <int NT> __device__ __forceinline__ v4 ld(const v4 *p)
{
    if constexpr (NT & 1) return __builtin_nontemporal_load(p);
    else return *p;
}
template <int NT> __device__ __forceinline__ void st(v4 *p, v4 v)
{
    if constexpr (NT & 2) __builtin_nontemporal_store(v, p);
    else *p = v;
}

struct P6 { const v4 *s[3]; v4 *d[3]; };

// ---------------------------------------------------------------- flat
template <int NT, int U>
__global__ __launch_bounds__(256) void k_flat(P6 P, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
        v4 a[U][3];
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int p = 0; p < 3; p++) a[u][p] = (i + u * stride < n16) ? ld<NT>(P.s[p] + i + u * stride) : v4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++)
#pragma unroll
            for (int p = 0; p < 3; p++) {
                a[u][p].x += 1u;
                if (i + u * stride < n16) st<NT>(P.d[p] + i + u * stride, a[u][p]);
            }
    }
}

// one buffer in, one out (the guide's reference point), same total bytes
template <int NT, int U>
__global__ __launch_bounds__(256) void k_copy(const v4 *s, v4 *d, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride * U) {
        v4 a[U];
#pragma unroll
        for (int u = 0; u < U; u++) a[u] = (i + u * stride < n16) ? ld<NT>(s + i + u * stride) : v4{0, 0, 0, 0};
#pragma unroll
        for (int u = 0; u < U; u++) { a[u].x += 1u; if (i + u * stride < n16) st<NT>(d + i + u * stride, a[u]); }
    }
}

// one thread per 16 bytes, no loop: as many blocks as it takes (the shape of a plain elementwise kernel)
template <int NT>
__global__ __launch_bounds__(256) void k_copy1(const v4 *s, v4 *d, size_t n16)
{
    const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n16) { v4 a = ld<NT>(s + i); a.x += 1u; st<NT>(d + i, a); }
}
// read only (three planes summed into one word per thread) and write only
template <int NT>
__global__ __launch_bounds__(256) void k_read(P6 P, size_t n16, unsigned *sink)
{
    const size_t stride = (size_t)gridDim.x * 256;
    unsigned acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
#pragma unroll
        for (int p = 0; p < 3; p++) { const v4 a = ld<NT>(P.s[p] + i); acc += a.x ^ a.y ^ a.z ^ a.w; }
    if (acc == 0x12345678u) *sink = acc;
}
template <int NT>
__global__ __launch_bounds__(256) void k_write(P6 P, size_t n16)
{
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride)
#pragma unroll
        for (int p = 0; p < 3; p++) st<NT>(P.d[p] + i, v4{(unsigned)i, 1u, 2u, 3u});
}

// E4: the no-loop copy with the tile kernels' occupancy: 1024-thread workgroups that own the CU's whole LDS (one workgroup per CU,
// 16 waves), each copying PER x 16 KB of the three planes and exiting
template <int NT, int PER>
__global__ __launch_bounds__(1024) void k_copy_wg(P6 P, size_t n16)
{
    extern __shared__ char lds_[];
    if (threadIdx.x == 99999) lds_[0] = 1;
#pragma unroll
    for (int j = 0; j < PER; j++) {
        const size_t i = ((size_t)blockIdx.x * PER + j) * 1024 + threadIdx.x;
        if (i < n16) {
            v4 a[3];
#pragma unroll
            for (int p = 0; p < 3; p++) a[p] = ld<NT>(P.s[p] + i);
#pragma unroll
            for (int p = 0; p < 3; p++) { a[p].x += 1u; st<NT>(P.d[p] + i, a[p]); }
        }
    }
}
// E3: persistent, every wave owns ONE contiguous run of each plane (4096 sequential streams per plane)
template <int NT>
__global__ __launch_bounds__(1024) void k_runs(P6 P, size_t n16)
{
    const size_t waves = (size_t)gridDim.x * 16, w = (size_t)blockIdx.x * 16 + (threadIdx.x >> 6);
    const size_t per = (n16 + waves - 1) / waves, lo = w * per, hi = lo + per < n16 ? lo + per : n16;
    for (size_t i = lo + (threadIdx.x & 63); i < hi; i += 64) {
        v4 a[3];
#pragma unroll
        for (int p = 0; p < 3; p++) a[p] = ld<NT>(P.s[p] + i);
#pragma unroll
        for (int p = 0; p < 3; p++) { a[p].x += 1u; st<NT>(P.d[p] + i, a[p]); }
    }
}
// E1: the persistent grid-stride copy with every wave delayed by a pseudo-random time first (breaks the lock step of the waves)
template <int NT>
__global__ __launch_bounds__(256) void k_flat_jitter(P6 P, size_t n16)
{
    const unsigned h = (blockIdx.x * 2654435761u + (threadIdx.x >> 6) * 40503u) >> 20;       // 0 .. 4095
    for (unsigned k = 0; k < (h & 1023u); k++) __builtin_amdgcn_s_sleep(8);
    const size_t stride = (size_t)gridDim.x * 256;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n16; i += stride) {
        v4 a[3];
#pragma unroll
        for (int p = 0; p < 3; p++) a[p] = ld<NT>(P.s[p] + i);
#pragma unroll
        for (int p = 0; p < 3; p++) { a[p].x += 1u; st<NT>(P.d[p] + i, a[p]); }
    }
}

// ---------------------------------------------------------------- tile walker
struct TG { int lw_log2, nsx, nry, ch, nrc, nchunks, row16, rows, frames; int stagger; unsigned *queue; size_t frame16; };

template <int NT, int WPB, int DEPTH>
__global__ __launch_bounds__(64 * WPB) void k_tile(P6 P, TG g)
{
    const int lane = threadIdx.x & 63;
    const int lw = 1 << g.lw_log2, lh = 64 >> g.lw_log2;
    const int lx = lane & (lw - 1), ly = lane >> g.lw_log2;
    unsigned c = blockIdx.x * WPB + (threadIdx.x >> 6);
    bool first = true;
    for (;;) {
        if (!first) {
            if (lane == 0) c = atomicAdd(g.queue, 1u);
            c = __builtin_amdgcn_readfirstlane(c);
        }
        first = false;
        if (c >= (unsigned)g.nchunks) break;
        const int per_frame = g.nrc * g.nsx;
        const int fr = c / per_frame, r = c - fr * per_frame, rc = r / g.nsx, sx = r - rc * g.nsx;
        const int ry0 = rc * g.ch, nt = min(g.ch, g.nry - ry0);
        // lane's 16-byte column and first row
        const int col = min(sx * lw + lx, g.row16 - 1);
        size_t base = (size_t)fr * g.frame16 + (size_t)col;
        v4 cur[DEPTH][3];
#pragma unroll
        for (int d = 0; d < DEPTH; d++) {
            const int row = min((ry0 + min(d, nt - 1)) * lh + ly, g.rows - 1);
#pragma unroll
            for (int p = 0; p < 3; p++) cur[d][p] = ld<NT>(P.s[p] + base + (size_t)row * g.row16);
        }
        for (int t = 0; t < nt; t++) {
            v4 out[3];
#pragma unroll
            for (int p = 0; p < 3; p++) { out[p] = cur[0][p]; out[p].x += 1u; }
#pragma unroll
            for (int d = 0; d + 1 < DEPTH; d++)
#pragma unroll
                for (int p = 0; p < 3; p++) cur[d][p] = cur[d + 1][p];
            {
                const int row = min((ry0 + min(t + DEPTH, nt - 1)) * lh + ly, g.rows - 1);
#pragma unroll
                for (int p = 0; p < 3; p++) cur[DEPTH - 1][p] = ld<NT>(P.s[p] + base + (size_t)row * g.row16);
            }
            const int row = min((ry0 + t) * lh + ly, g.rows - 1);
#pragma unroll
            for (int p = 0; p < 3; p++) st<NT>(P.d[p] + base + (size_t)row * g.row16, out[p]);
        }
    }
}



#include <functional>
static double run_ms(hipStream_t s, int reps, const std::function<void()> &f)
{
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(e0, s);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    hipEventDestroy(e0); hipEventDestroy(e1);
    return ms / reps;
}

int main(int argc, char **argv)
{
    const int frames = argc > 1 ? atoi(argv[1]) : 128, w = 3840, h = 2160;
    const size_t plane = (size_t)frames * w * h * 2, n16 = plane / 16;
    P6 P;
    std::vector<void *> bufs;
    // planes interleaved in allocation order like torch does for a list of tensors: s0 s1 s2 d0 d1 d2
    for (int p = 0; p < 3; p++) { void *q; if (hipMalloc(&q, plane) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(q, 1, plane); P.s[p] = (const v4 *)q; bufs.push_back(q); }
    for (int p = 0; p < 3; p++) { void *q; if (hipMalloc(&q, plane) != hipSuccess) { printf("alloc failed\n"); return 1; } P.d[p] = (v4 *)q; bufs.push_back(q); }
    unsigned *queue; hipMalloc(&queue, 4);
    const double gb = 6.0 * plane / 1e9;
    printf("3 planes in + 3 out, %d UHD 16-bit frames: %.2f GB per pass\n", frames, gb);
    hipStream_t s = 0;
    auto report = [&](const char *name, double ms) { printf("%-64s %8.3f ms  %6.2f TB/s\n", name, ms, gb / ms); fflush(stdout); };

    // the two directions alone, and the runtime's own device-to-device copy
    {
        const double gbh = 3.0 * plane / 1e9;
        auto rep1 = [&](const char *name, double ms) { printf("%-64s %8.3f ms  %6.2f TB/s (one direction: %.2f GB)\n", name, ms, gbh / ms, gbh); fflush(stdout); };
        rep1("read only, 3 planes, blocks 65536, plain", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_read<0>), dim3(65536), dim3(256), 0, s, P, n16, queue); }));
        rep1("read only, 3 planes, blocks 65536, nt", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_read<1>), dim3(65536), dim3(256), 0, s, P, n16, queue); }));
        rep1("write only, 3 planes, blocks 65536, plain", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_write<0>), dim3(65536), dim3(256), 0, s, P, n16); }));
        rep1("write only, 3 planes, blocks 65536, nt", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_write<2>), dim3(65536), dim3(256), 0, s, P, n16); }));
        report("hipMemcpyDtoDAsync x3 (the runtime's copy)", run_ms(s, 5, [&] { for (int p = 0; p < 3; p++) hipMemcpyDtoDAsync((hipDeviceptr_t)P.d[p], (hipDeviceptr_t)P.s[p], plane, s); }));
        const unsigned nblk = (unsigned)((n16 + 255) / 256);
        report("copy, one thread per 16 B (no loop) x3 launches, plain", run_ms(s, 5, [&] { for (int p = 0; p < 3; p++) hipLaunchKernelGGL((k_copy1<0>), dim3(nblk), dim3(256), 0, s, P.s[p], P.d[p], n16); }));
        report("copy, one thread per 16 B (no loop) x3 launches, nt ld+st", run_ms(s, 5, [&] { for (int p = 0; p < 3; p++) hipLaunchKernelGGL((k_copy1<3>), dim3(nblk), dim3(256), 0, s, P.s[p], P.d[p], n16); }));
    }
    {
        hipFuncSetAttribute((const void *)k_copy_wg<2, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void *)k_copy_wg<2, 4>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void *)k_copy_wg<2, 16>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        hipFuncSetAttribute((const void *)k_runs<2>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (size_t ldsb : {(size_t)0, (size_t)160 * 1024}) {
            char nm[128];
            snprintf(nm, sizeof nm, "3+3 no loop, 1024-thread blocks x 16 KB/plane, LDS %zu KB, nt st", ldsb / 1024);
            report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_copy_wg<2, 1>), dim3((unsigned)((n16 + 1023) / 1024)), dim3(1024), ldsb, s, P, n16); }));
            snprintf(nm, sizeof nm, "3+3 short loop (4), 1024-thread blocks x 64 KB/plane, LDS %zu KB, nt st", ldsb / 1024);
            report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_copy_wg<2, 4>), dim3((unsigned)((n16 + 4095) / 4096)), dim3(1024), ldsb, s, P, n16); }));
            snprintf(nm, sizeof nm, "3+3 short loop (16), 1024-thread blocks x 256 KB/plane, LDS %zu KB, nt st", ldsb / 1024);
            report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_copy_wg<2, 16>), dim3((unsigned)((n16 + 16383) / 16384)), dim3(1024), ldsb, s, P, n16); }));
        }
        report("3+3 persistent, one contiguous run per wave (256 x 16 waves), nt st", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_runs<2>), dim3(256), dim3(1024), 160 * 1024, s, P, n16); }));
        report("3+3 persistent, one contiguous run per wave (512 x 16 waves, no LDS), nt st", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_runs<2>), dim3(512), dim3(1024), 0, s, P, n16); }));
        report("flat 3+3 grid-stride, blocks 2048, waves jittered at start, nt st", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat_jitter<2>), dim3(2048), dim3(256), 0, s, P, n16); }));
        report("flat 3+3 grid-stride, blocks 1024 (16 waves/CU), jittered, nt st", run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat_jitter<2>), dim3(1024), dim3(256), 0, s, P, n16); }));
    }
    // one big buffer pair (the guide's float4 copy), same bytes
    {
        // treat s0..s2 as one region if contiguous? not guaranteed: copy plane 0 -> plane 0 three times the size is not possible,
        // so time three launches back to back instead
        for (int blocks : {2048, 8192, 65536}) {
            char nm[128];
            snprintf(nm, sizeof nm, "copy (1 in, 1 out) x3 launches, blocks %d, U4, plain", blocks);
            report(nm, run_ms(s, 5, [&] { for (int p = 0; p < 3; p++) hipLaunchKernelGGL((k_copy<0, 4>), dim3(blocks), dim3(256), 0, s, P.s[p], P.d[p], n16); }));
            snprintf(nm, sizeof nm, "copy (1 in, 1 out) x3 launches, blocks %d, U4, nt ld+st", blocks);
            report(nm, run_ms(s, 5, [&] { for (int p = 0; p < 3; p++) hipLaunchKernelGGL((k_copy<3, 4>), dim3(blocks), dim3(256), 0, s, P.s[p], P.d[p], n16); }));
        }
    }
    for (int blocks : {2048, 8192, 65536}) {
        char nm[128];
        snprintf(nm, sizeof nm, "flat 3+3, blocks %d, U1, plain", blocks); report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat<0, 1>), dim3(blocks), dim3(256), 0, s, P, n16); }));
        snprintf(nm, sizeof nm, "flat 3+3, blocks %d, U2, plain", blocks); report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat<0, 2>), dim3(blocks), dim3(256), 0, s, P, n16); }));
        snprintf(nm, sizeof nm, "flat 3+3, blocks %d, U2, nt st", blocks); report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat<2, 2>), dim3(blocks), dim3(256), 0, s, P, n16); }));
        snprintf(nm, sizeof nm, "flat 3+3, blocks %d, U2, nt ld+st", blocks); report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat<3, 2>), dim3(blocks), dim3(256), 0, s, P, n16); }));
        snprintf(nm, sizeof nm, "flat 3+3, blocks %d, U4, nt st", blocks); report(nm, run_ms(s, 5, [&] { hipLaunchKernelGGL((k_flat<2, 4>), dim3(blocks), dim3(256), 0, s, P, n16); }));
    }
    // tile walker: the product's shape
    const int row16 = w * 2 / 16;   // 480 sixteen-byte words per row
    auto tile = [&](int lw_log2, int ch, int wpb, int blocks_per_cu, int nt, int depth) {
        TG g;
        g.lw_log2 = lw_log2; g.row16 = row16; g.rows = h; g.frames = frames; g.frame16 = (size_t)row16 * h;
        const int lw = 1 << lw_log2, lh = 64 >> lw_log2;
        g.nsx = (row16 + lw - 1) / lw; g.nry = (h + lh - 1) / lh; g.ch = ch; g.nrc = (g.nry + ch - 1) / ch;
        g.nchunks = frames * g.nrc * g.nsx; g.queue = queue; g.stagger = 0;
        const int blocks = 256 * blocks_per_cu;
        char nm[160];
        snprintf(nm, sizeof nm, "tile %2dx%-2d lanes, chunk %3d tiles, %2d waves/block x %d blocks/CU, nt %d, depth %d", lw, lh, ch, wpb, blocks_per_cu, nt, depth);
        auto go = [&] {
            hipMemsetD32Async((hipDeviceptr_t)queue, blocks * wpb, 1, s);
#define LAUNCH(NT, WPB, D) hipLaunchKernelGGL((k_tile<NT, WPB, D>), dim3(blocks), dim3(64 * WPB), 0, s, P, g)
            if (wpb == 16 && depth == 1) { if (nt == 0) LAUNCH(0, 16, 1); else if (nt == 2) LAUNCH(2, 16, 1); else LAUNCH(3, 16, 1); }
            else if (wpb == 16 && depth == 2) { if (nt == 0) LAUNCH(0, 16, 2); else if (nt == 2) LAUNCH(2, 16, 2); else LAUNCH(3, 16, 2); }
            else if (wpb == 16 && depth == 4) { if (nt == 0) LAUNCH(0, 16, 4); else if (nt == 2) LAUNCH(2, 16, 4); else LAUNCH(3, 16, 4); }
            else if (wpb == 8 && depth == 1) { if (nt == 0) LAUNCH(0, 8, 1); else if (nt == 2) LAUNCH(2, 8, 1); else LAUNCH(3, 8, 1); }
            else if (wpb == 8 && depth == 2) { if (nt == 0) LAUNCH(0, 8, 2); else if (nt == 2) LAUNCH(2, 8, 2); else LAUNCH(3, 8, 2); }
            else if (wpb == 4 && depth == 2) { if (nt == 0) LAUNCH(0, 4, 2); else if (nt == 2) LAUNCH(2, 4, 2); else LAUNCH(3, 4, 2); }
            else if (wpb == 4 && depth == 1) { if (nt == 0) LAUNCH(0, 4, 1); else if (nt == 2) LAUNCH(2, 4, 1); else LAUNCH(3, 4, 1); }
        };
        report(nm, run_ms(s, 5, go));
    };
    for (int nt : {0, 2, 3}) tile(5, 16, 16, 1, nt, 1);          // the product's RGB shape: 32 x 2 lanes, 16 waves per CU
    for (int lw : {4, 5, 6}) for (int ch : {8, 16, 32, 64}) tile(lw, ch, 16, 1, 2, 1);
    for (int depth : {2, 4}) for (int lw : {4, 5, 6}) tile(lw, 32, 16, 1, 2, depth);
    for (int bpc : {2, 3, 4}) for (int depth : {1, 2}) tile(5, 32, 8, bpc, 2, depth);      // 16 / 24 / 32 waves per CU
    for (int bpc : {4, 6, 8}) for (int depth : {1, 2}) tile(5, 32, 4, bpc, 2, depth);
    for (int nt : {0, 3}) tile(6, 32, 8, 4, nt, 2);
    for (void *q : bufs) hipFree(q);
    return 0;
}
