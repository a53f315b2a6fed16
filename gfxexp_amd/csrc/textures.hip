// textures.hip -- test / inspection entry of the software texture unit (texture.hip.h): samples one texture at a
// list of coordinates with exactly the functions the shading kernels call (tex2d, tex2d_gather_r).
#include "internal.h"
#include "shading.hip.h"

namespace gfx {

__global__ void k_texture_sample(DevScene sc, uint32_t texSlot, const float2* __restrict__ uv, uint32_t n, float4* __restrict__ out, int gather) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float2 c = uv[i];
    out[i] = gather ? tex2d_gather_r(sc, texSlot, c.x, c.y) : tex2d(sc, texSlot, c.x, c.y);
}

void texture_sample(Context& ctx, hipStream_t stream, uint32_t texSlot, const void* dUv, uint32_t n, void* dOut, int gather) {
    scene_upload(ctx, stream);
    if (texSlot == 0 || texSlot >= ctx.textures.size() || ctx.textures[texSlot].width == 0) throw HipError("gfx_texture_sample: texture slot was never set");
    if (!n) return;
    hipLaunchKernelGGL(k_texture_sample, dim3((n + 255) / 256), dim3(256), 0, stream, ctx.devScene(), texSlot,
                       static_cast<const float2*>(dUv), n, static_cast<float4*>(dOut), gather);
    GFX_HIP(hipGetLastError());
}

} // namespace gfx
