// aux_kernels.hip -- data-movement kernels around the integrator (no arithmetic on radiance).
#include <hip/hip_runtime.h>

#include "render_args.h"

// Gather the per-owner compact tile buffers into the row-major whole frame.
// gathered: tileCount consecutive buffers of map.slotsPerOwner float4 (rank order).
extern "C" __global__ void __launch_bounds__(256) kajo_compose(const float4* gathered, TileMap map, float4* frame)
{
    const int x = blockIdx.x * 64 + (threadIdx.x & 63);
    const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= map.W || y >= map.H)
        return;
    int owner;
    uint32_t slot;
    kajoTileSlot(map, x, y, &owner, &slot);
    frame[(size_t)y * map.W + x] = gathered[(size_t)owner * map.slotsPerOwner + slot];
}

extern "C" int kajo_compose_launch(const void* gathered, const TileMap* map, void* frame, void* stream)
{
    dim3 grid((map->W + 63) / 64, (map->H + 3) / 4), block(256);
    hipLaunchKernelGGL(kajo_compose, grid, block, 0, static_cast<hipStream_t>(stream),
                       static_cast<const float4*>(gathered), *map, static_cast<float4*>(frame));
    return (int)hipGetLastError();
}

// Launch tail (integrator.inc.hip PARTS, capi.cpp partedOrder): blocks rendered in 2 or 4 parts have part 0's sum -- the pixel's total so
// far plus its passes -- in `tiles` and the later parts' sums in the side buffers; the total is their sum in part order. One workgroup per
// such block, `threads` = the render kernel's workgroup size (a block's slots are block * threads ...). Plain additions, no products: the
// result does not depend on the contraction setting of this file.
extern "C" __global__ void __launch_bounds__(256) kajo_fold_parts(float4* tiles, const float4* side, uint32_t sideStride, const uint32_t* blocks)
{
    const uint32_t word = blocks[blockIdx.x];
    const uint32_t slot = (word & 0x0fffffffu) * blockDim.x + threadIdx.x;
    const int parts = 1 << (word >> 30);
    float4 t = tiles[slot];
    for (int k = 1; k < parts; k++) {
        const float4 s = side[(size_t)(k - 1) * sideStride + slot];
        t.x += s.x;
        t.y += s.y;
        t.z += s.z;
    }
    tiles[slot] = t;
}

extern "C" int kajo_fold_parts_launch(void* tiles, const void* side, uint32_t sideStride, const uint32_t* blocks, unsigned count, unsigned threads, void* stream)
{
    hipLaunchKernelGGL(kajo_fold_parts, dim3(count), dim3(threads), 0, static_cast<hipStream_t>(stream), static_cast<float4*>(tiles),
                       static_cast<const float4*>(side), sideStride, blocks);
    return (int)hipGetLastError();
}
