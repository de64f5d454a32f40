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

// Launch tail (integrator.inc.hip PARTS, capi.cpp partTheTail): a block rendered in `parts` workgroups has part 0's sum -- the pixel's total
// so far plus the launch's first group -- in `tiles` and the later groups' sums in the compact side buffers (sideStride slots each, the
// j-th parted block's at j * blockDim.x); the total is their sum in group order. One workgroup per such block, `threads` = the render
// kernel's workgroup size (a block's slots are block * threads ...). Plain additions, no products: the result does not depend on the
// contraction setting of this file.
extern "C" __global__ void __launch_bounds__(256) kajo_fold_parts(float4* tiles, const float4* side, uint32_t sideStride, const uint32_t* blocks, int parts)
{
    const uint32_t slot = (blocks[blockIdx.x] & 0x0fffffffu) * blockDim.x + threadIdx.x;
    const uint32_t sideSlot = blockIdx.x * blockDim.x + threadIdx.x;
    float4 t = tiles[slot];
    for (int k = 1; k < parts; k++) {
        const float4 s = side[(size_t)(k - 1) * sideStride + sideSlot];
        t.x += s.x;
        t.y += s.y;
        t.z += s.z;
    }
    tiles[slot] = t;
}

extern "C" int kajo_fold_parts_launch(void* tiles, const void* side, uint32_t sideStride, const uint32_t* blocks, unsigned count, unsigned threads, int parts,
                                      void* stream)
{
    hipLaunchKernelGGL(kajo_fold_parts, dim3(count), dim3(threads), 0, static_cast<hipStream_t>(stream), static_cast<float4*>(tiles),
                       static_cast<const float4*>(side), sideStride, blocks, parts);
    return (int)hipGetLastError();
}
