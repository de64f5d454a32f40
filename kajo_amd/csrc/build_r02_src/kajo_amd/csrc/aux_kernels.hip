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
