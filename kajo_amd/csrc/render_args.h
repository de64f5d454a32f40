// render_args.h -- by-value argument block of the render / compose / resolve kernels.
#ifndef KAJO_RENDER_ARGS_H
#define KAJO_RENDER_ARGS_H

#include <stdint.h>

#include "device_scene.h"

struct RenderArgs
{
    DSceneView scene;
    void* tiles;          // this handle's compact tile buffer, float4 per slot
    int32_t W, H;         // whole image
    int32_t n;            // strata per axis = (int)sqrt(S), Renderer.cpp:38
    float S;              // (float)S, the divisor of Renderer.cpp:71
    float pixelWidth, pixelHeight, sampleWidth, sampleHeight; // Renderer.cpp:39-42
    int32_t firstPass, nPasses;  // passes [firstPass, firstPass + nPasses), 1-based
    int32_t depthLimit;
    uint64_t seed;
    int32_t tileW, tileH, tilesX, tilesY;
    int32_t tileIndex, tileCount, nTilesOwned;
    unsigned long long* counters; // [0] traversals, [1] vertices, [2] lane slots; may be null
    uint32_t mailboxOffset;       // byte offset of what follows the scene copy in dynamic LDS (16-byte aligned): the SPLIT kernels' term table
    int32_t sampleChunks;         // SPLIT kernels: > 1 = every pass's n*n samples are divided over this many waves of the block (integrator.inc.hip)
    int32_t stealWindow;          // passes at the end of a launch an idle lane may take over (1..KAJO_STEAL_WINDOW_MAX): sizes the mailboxes
    // Per-wave LDS (integrator.inc.hip renderBody): wave w of the workgroup owns perWaveBytes at perWaveOffset + w * perWaveBytes:
    // the mailbox of taken-over passes, 64 lanes x stealWindow x float4 (absent in the SPLIT kernels).
    uint32_t perWaveOffset, perWaveBytes;
    int32_t thrL;                 // MODE_HOLD: the light / BSDF blocks run in a trip when this many lanes want them, or when they
    int32_t holdTrips;            // have been put off this many trips in a row (1: no vertex waits twice)
    // Launch-order feedback: blocks are dispatched in blockIdx order; the host sorts them by the cost the
    // previous launch measured (longest first) so that the launch does not end on its most expensive
    // workgroups. Pure scheduling: the buffer slot of a pixel does not depend on it.
    const uint32_t* blockOrder;   // [grid] logical block run by physical block i; null = identity
    uint32_t* waveTrips;          // [grid * 4] loop trips of every wave of the launch; null = not recorded
    // Launch tail: bits 30-31 of an order word = log2 of the parts its block is rendered in (0: one workgroup renders all nPasses
    // passes), bits 28-29 = which part this workgroup is, bits 0-27 the block. Part k > 0 writes its sum to slot + k * sideStride: three
    // side buffers follow the tile buffer in the same allocation.
    uint32_t sideStride;          // slots per buffer; 0 unless the order holds parts
    // FAST / EXACT kernels of small scenes (integrator.inc.hip GROUPS): the passes of the launch enter a pixel's total in groups of
    // groupMask + 1 passes -- nPasses / 4 when nPasses is 8, 16, 32 ... and firstPass - 1 a multiple of it: a group ends before pass p
    // when ((p - 1) & groupMask) == 0 -- or in one group (groupMask 0x7fffffff: no pass number ends one). STRICT adds pass by pass and
    // does not read it.
    int32_t groupMask;
    // known-answer mode (kajo_hip_kat_shade): lane i runs ONE path from a given ray and RNG state
    const float* katRays;         // [katCount][6] origin, direction
    const uint64_t* katStates;    // [katCount][2]
    float* katRgb;                // [katCount][4]
    uint64_t* katFinal;           // [katCount][2]
    int32_t katCount;
};

struct KatTraceArgs
{
    DSceneView scene;
    const float* rays; // [count][6]
    int32_t count;
    int32_t* idx;      // [count]
    float* out;        // [count][13]: t, position, normal, tangent, binormal
};

// tile-buffer slot of pixel (x, y): tiles are dealt round-robin to `tileCount` owners; inside a
// tile pixels are grouped in 8x8 blocks (one wave each) so that a wave's 64 float4 are
// contiguous (1 KiB per store instruction).
struct TileMap
{
    int32_t W, H, tileW, tileH, tilesX, tileCount;
    int32_t slotsPerOwner; // padded tile count per owner * tileW * tileH
};

#if defined(__HIPCC__)
__host__ __device__
#endif
static inline void kajoTileSlot(const TileMap& m, int x, int y, int* owner, uint32_t* slot)
{
    const int tx = x / m.tileW, ty = y / m.tileH;
    const int tile = ty * m.tilesX + tx;
    const int ix = x - tx * m.tileW, iy = y - ty * m.tileH;
    const int wavesPerTile = (m.tileW >> 3) * (m.tileH >> 3);
    const int wb = (iy >> 3) * (m.tileW >> 3) + (ix >> 3);
    const int lane = ((iy & 7) << 3) | (ix & 7);
    *owner = tile % m.tileCount;
    *slot = (uint32_t)(((tile / m.tileCount) * wavesPerTile + wb) * 64 + lane);
}

#endif
