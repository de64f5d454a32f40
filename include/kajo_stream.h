/*
 * kajo_stream.h -- the per-sample RNG stream protocol (part of the interface contract).
 *
 * The reference draws every pixel of a row slice from ONE serial cpu::Random stream
 * (renderer/cpu/Renderer.cpp:27), which makes its output depend on the host's core count
 * and on every earlier branch. For a result that is independent of
 * slicing, tiling, lane scheduling and GPU count, every camera path gets its own stream:
 * the 128-bit state of cpu::Random (renderer/cpu/Random.h:63-68, one __m128i = lo64, hi64)
 * is set to two consecutive splitmix64 outputs of a key built from
 * (seed, pass, sample index within the pixel, global pixel index), immediately before the
 * jitter draw of that path (the draw at Renderer.cpp:55). From there on the path consumes
 * the reference generator (Random.cpp:27-53) exactly as the reference does.
 *
 * Plain C, integer only; usable from host C/C++ and from HIP device code.
 */
#ifndef KAJO_STREAM_H
#define KAJO_STREAM_H

#include <stdint.h>

#if defined(__HIPCC__)
#define KAJO_HD __host__ __device__ static inline
#else
#define KAJO_HD static inline
#endif

/* One splitmix64 step (Steele, Lea, Flood 2014; public-domain constants). */
KAJO_HD uint64_t kajo_splitmix64(uint64_t* z)
{
    uint64_t x = (*z += 0x9E3779B97F4A7C15ull);
    x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
    x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
    return x ^ (x >> 31);
}

/*
 * pass:   1-based pass number, as the reference counts them (Renderer.cpp:44), < 2^16
 * sample: sampleY * n + sampleX, n = (int)sqrt(S) (Renderer.cpp:38,51-53), < 2^16
 * pixel:  y * W + x in whole-image coordinates, row 0 = top, < 2^32
 * state:  [0] = low 64 bits, [1] = high 64 bits of the __m128i
 */
KAJO_HD void kajo_stream_state(uint64_t seed, uint32_t pass, uint32_t sample, uint32_t pixel,
                               uint64_t state[2])
{
    uint64_t z = (seed * 0x9E3779B97F4A7C15ull) ^ ((uint64_t)pass << 48) ^
                 ((uint64_t)sample << 32) ^ (uint64_t)pixel;
    state[0] = kajo_splitmix64(&z);
    state[1] = kajo_splitmix64(&z);
}

#endif /* KAJO_STREAM_H */
