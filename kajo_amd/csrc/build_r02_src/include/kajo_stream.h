/*
 * kajo_stream.h -- the per-sample RNG stream protocol (part of the interface contract).
 *
 * The reference draws every pixel of a row slice from ONE serial cpu::Random stream
 * (renderer/cpu/Renderer.cpp:27), which makes its output depend on the host's core count
 * and on every earlier branch. For a result that is independent of
 * slicing, tiling, lane scheduling and GPU count, every camera path gets its own stream:
 * the 128-bit state of cpu::Random (renderer/cpu/Random.h:63-68, one __m128i = lo64, hi64)
 * is set to a 128-bit hash of the key
 * (seed, pass, sample index within the pixel, global pixel index), immediately before the
 * jitter draw of that path (the draw at Renderer.cpp:55). From there on the path consumes
 * the reference generator (Random.cpp:27-53) exactly as the reference does.
 *
 * Plain C, integer only; usable from host C/C++ and from HIP device code. The key hash is
 * add-rotate-xor only (three ChaCha quarter-rounds over the four 32-bit key words, D. J.
 * Bernstein's public-domain construction): every operation is a full-rate 32-bit VALU
 * instruction on gfx950, where 64-bit integer multiplies (splitmix64, PCG, ...) run at a quarter
 * of that rate. Avalanche over (pixel, sample, pass) bits is complete after three rounds
 * (|p - 1/2| < 0.014 for every input/output bit pair over 20 000 keys).
 */
#ifndef KAJO_STREAM_H
#define KAJO_STREAM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KAJO_HD __host__ __device__ static inline
#else
#define KAJO_HD static inline
#endif

KAJO_HD uint32_t kajo_rotl32(uint32_t x, int r)
{
    return (x << r) | (x >> (32 - r));
}

#define KAJO_QUARTER_ROUND(a, b, c, d)                                                                                 \
    do {                                                                                                               \
        a += b; d ^= a; d = kajo_rotl32(d, 16);                                                                        \
        c += d; b ^= c; b = kajo_rotl32(b, 12);                                                                        \
        a += b; d ^= a; d = kajo_rotl32(d, 8);                                                                         \
        c += d; b ^= c; b = kajo_rotl32(b, 7);                                                                         \
    } while (0)

/*
 * pass:   1-based pass number, as the reference counts them (Renderer.cpp:44: `for (pass = 1;; pass++)`, unbounded);
 *         its low 16 bits share key word b with the sample index, the bits above them enter key word c
 *         (zero for the first 65535 passes, so those streams are the ones of the 16-bit protocol)
 * sample: sampleY * n + sampleX, n = (int)sqrt(S) (Renderer.cpp:38,51-53), < 2^16
 * pixel:  y * W + x in whole-image coordinates, row 0 = top, < 2^32
 * state:  [0] = low 64 bits, [1] = high 64 bits of the __m128i
 */
KAJO_HD void kajo_stream_state(uint64_t seed, uint32_t pass, uint32_t sample, uint32_t pixel,
                               uint64_t state[2])
{
    uint32_t a = pixel ^ 0x61707865u;
    uint32_t b = (sample | (pass << 16)) ^ 0x3320646eu;
    uint32_t c = (uint32_t)seed ^ (pass >> 16) ^ 0x79622d32u;
    uint32_t d = (uint32_t)(seed >> 32) ^ 0x6b206574u;
    KAJO_QUARTER_ROUND(a, b, c, d);
    KAJO_QUARTER_ROUND(a, b, c, d);
    KAJO_QUARTER_ROUND(a, b, c, d);
    state[0] = (uint64_t)a | ((uint64_t)b << 32);
    state[1] = (uint64_t)c | ((uint64_t)d << 32);
}

#endif /* KAJO_STREAM_H */
