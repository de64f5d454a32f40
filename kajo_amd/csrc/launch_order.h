// launch_order.h -- the order a handle dispatches its workgroups in, as plain host arithmetic (capi.cpp updateBlockOrder / partTheTail;
// kajo_hip_launch_order exports it for the host-only tests, which also run it under the sanitizers).
#ifndef KAJO_LAUNCH_ORDER_H
#define KAJO_LAUNCH_ORDER_H

#include <stdint.h>

#include <algorithm>
#include <vector>

// An order word (render_args.h): bits 0-27 the block, bits 28-30 which group of the launch the workgroup renders, bit 31 the block is parted.
#define KAJO_ORDER_BLOCK_MASK 0x0fffffffu
#define KAJO_ORDER_PART_SHIFT 28
#define KAJO_ORDER_PARTED 0x80000000u

// A block runs as long as its slowest wave: the cost of block b is the most loop trips any of its waves made.
inline void kajoBlockCosts(const uint32_t* waveTrips, size_t nBlocks, unsigned wavesPerBlock, std::vector<uint32_t>& cost)
{
    cost.assign(nBlocks, 0u);
    for (size_t b = 0; b < nBlocks; b++)
        for (unsigned k = 0; k < wavesPerBlock; k++)
            cost[b] = std::max(cost[b], waveTrips[(size_t)wavesPerBlock * b + k]);
}

// Longest-processing-time-first: blocks by the cost the first launch measured, the most expensive first; equal costs stay in image order.
inline void kajoCostOrder(const std::vector<uint32_t>& cost, std::vector<uint32_t>& order)
{
    order.resize(cost.size());
    for (size_t b = 0; b < cost.size(); b++)
        order[b] = (uint32_t)b;
    std::stable_sort(order.begin(), order.end(), [&](uint32_t x, uint32_t y) { return cost[x] > cost[y]; });
}

// How many of the order's last (cheapest) blocks are rendered in parts: half the wave slots' worth (measured: tools/tail_sweep.sh, q4 = 4
// eighths of the slots), at most half the frame; none for frames of one or two rounds of the slots (the SPLIT kernels' business) or
// beyond what an order word can number.
inline unsigned kajoTailBlocks(unsigned nBlocks, unsigned slots, int eighths = 4)
{
    if (nBlocks >= (1u << 28) || nBlocks < 2 * slots)
        return 0;
    return std::min<unsigned>(nBlocks / 2, (unsigned)((unsigned long long)slots * (unsigned)eighths / 8));
}

// The order of a launch of `parts` groups: every block once, the last nParted of them as `parts` consecutive workgroups.
inline void kajoPartedOrder(const std::vector<uint32_t>& order, unsigned nParted, int parts, std::vector<uint32_t>& out)
{
    const size_t n = order.size();
    out.clear();
    out.reserve(n + (size_t)nParted * (size_t)(parts - 1));
    for (size_t i = 0; i < n; i++) {
        if (i + nParted < n) {
            out.push_back(order[i]);
            continue;
        }
        for (uint32_t k = 0; k < (uint32_t)parts; k++)
            out.push_back(order[i] | (k << KAJO_ORDER_PART_SHIFT) | KAJO_ORDER_PARTED);
    }
}

#endif
