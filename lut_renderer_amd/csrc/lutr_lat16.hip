// lutr_lat16.hip -- the lattice of the fast variant: every node as four fp16 {r, g, b, 0} of value * (2^depth - 1),
// round to nearest even.  Pre-multiplying folds FFmpeg's `v * M` (vf_lut3d.c) into the node, so the blend
// of the fast kernels is the output code before truncation.  |fp16(v * M) - v * M| <= 0.25 at 10 bit, 0.0625 at 8 bit
// for v in [0, 1].  Built once per (lattice, depth) by lutr_api.cpp.
#include <hip/hip_fp16.h>

#include "lutr_internal.h"

namespace lutr {

__global__ __launch_bounds__(256) void k_make_lat16(const float4 *__restrict__ lat, uint2 *__restrict__ out, size_t nodes, float m)
{
    const size_t i = blockIdx.x * 256ull + threadIdx.x;
    if (i >= nodes) return;
    const float4 v = lat[i];
    const unsigned r = __half_as_ushort(__float2half_rn(v.x * m)), g = __half_as_ushort(__float2half_rn(v.y * m));
    const unsigned b = __half_as_ushort(__float2half_rn(v.z * m));
    out[i] = make_uint2(r | (g << 16), b);
}

void launch_make_lat16(hipStream_t st, const float4 *lat, uint2 *out, size_t nodes, float m)
{
    hipLaunchKernelGGL(k_make_lat16, dim3((unsigned)((nodes + 255) / 256)), dim3(256), 0, st, lat, out, nodes, m);
}

}  // namespace lutr
