// render_args.h -- by-value argument block of the render / compose / resolve kernels.
#ifndef KAJO_RENDER_ARGS_H
#define KAJO_RENDER_ARGS_H

#include <stdint.h>

#include "device_scene.h"

#define KAJO_GROUP_PASSES 4 // passes per group of the FAST / EXACT totals (a power of two)

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
    // FAST / EXACT kernels of small scenes (integrator.inc.hip GROUPS): a pixel's total takes the passes in GROUPS of four by their ABSOLUTE
    // numbers -- passes 1-4, 5-8, ... (KAJO_GROUP_PASSES): every group is summed from zero in pass order, the group sums are added to the
    // total in group order. The frame after P passes is therefore a function of (scene, parameters, P) alone, however the passes were cut
    // into launches, waves, workgroups or GPUs. A launch that ends inside a group leaves the visible sum (complete groups + the partial
    // group) in `tiles` and the two summands in `carry`, where the launch that continues the group picks them up. STRICT adds pass by
    // pass (Renderer.cpp:70-71) and reads none of this.
    void* carry;                  // float4 [2][carrySlots]: the total of the complete groups, the sum of the group in progress; null unless
    uint32_t carrySlots;          // carryIn or carryOut
    int32_t carryIn;              // the launch starts inside a group whose first passes are in `carry`
    int32_t carryOut;             // the launch ends inside a group: write `carry`
    // Launch tail (capi.cpp partTheTail): the cheapest blocks of a launch of G = 2 .. 8 whole groups, last in the order, are rendered as G
    // workgroups of one group each. Order word: bit 31 = the block is parted, bits 28-30 = which group of the launch this workgroup
    // renders, bits 0-27 the block. Part 0 adds its group to the total in `tiles`; part k > 0 leaves its group's sum in side buffer
    // k - 1, and a fold kernel adds the side buffers in order after the launch. Side buffers are compact: sideStride slots each, the
    // slots of the j-th parted block of the order (physical blocks partedFirst + j * G ...) at j * blockDim.x.
    void* side;                   // float4 [G - 1][sideStride]
    uint32_t sideStride;          // slots per side buffer; 0 unless the order holds parts
    uint32_t partedFirst;         // physical index of the first workgroup of a parted block
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

// Launch tail: slot of thread `tid` of physical workgroup `physical` -- part `part` > 0 of a parted block -- in the compact side buffers
// (RenderArgs::side): buffer part - 1, behind the slots of the parted blocks before it in the order. The fold kernel reads the same
// place as (part - 1) * sideStride + j * threads + tid for the j-th parted block.
#if defined(__HIPCC__)
__host__ __device__
#endif
static inline uint32_t kajoSideSlot(uint32_t physical, uint32_t partedFirst, uint32_t nParts, uint32_t part, uint32_t sideStride, uint32_t threads, uint32_t tid)
{
    return (part - 1u) * sideStride + ((physical - partedFirst) / nParts) * threads + tid;
}

#endif
