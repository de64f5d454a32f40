// STRICT numerics: bit-identical to the CPU oracle. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#define KAJO_STRICT 1
#ifndef KAJO_INLINE_SHADOW
#define KAJO_INLINE_SHADOW 1 // small scenes answer shadow rays inside the light loop (integrator.inc.hip)
#endif
#define KAJO_KERNEL_NAME kajo_render_strict
#ifndef KAJO_STRICT_PRESAMPLE
#define KAJO_STRICT_PRESAMPLE 1 // small scenes of ONE light: their own instance, shadow ray in a trip of its own with the BSDF sampled in the light's visit
#endif
#if KAJO_STRICT_PRESAMPLE
#define KAJO_KERNEL_NAME_LIGHTS kajo_render_strict_lights // small scenes with several lights (or none): shadow walks inside the light loop
#endif
#define KAJO_KERNEL_NAME_BIG kajo_render_strict_big
#define KAJO_KERNEL_NAME_BIGLIST kajo_render_strict_biglist
// Large scenes, STRICT: an instance per home of the grid's cell lists (LDS: _lg; global memory: the plain names), each with typed loads and
// ONE walk -- a walk over a pointer of either home loads FLAT with a full wait behind every cell record, two walks in one kernel spill 20
// registers more. 1000 spheres / 16 lights: 3.25 -> 3.31 G paths/s. (FAST carries both walks in one kernel and measures 0.7 % faster so.)
#define KAJO_KERNEL_NAME_BIG_LG kajo_render_strict_big_lg
#define KAJO_KERNEL_NAME_BIGLIST_LG kajo_render_strict_biglist_lg
#define KAJO_KERNEL_NAME_SPLIT kajo_render_strict_split
#define KAJO_KAT_SHADE_NAME kajo_kat_shade_strict
#define KAJO_KAT_TRACE_NAME kajo_kat_trace_strict
#define KAJO_RESOLVE_NAME kajo_resolve_strict
#define KAJO_RESOLVE_TILES_NAME kajo_resolve_tiles_strict
#include "integrator.inc.hip"
#include "launch.inc.hip"

// include/kajo_strictmath.h element-wise on the device (kajo_hip_kat_strictmath): the claim that these
// functions give identical bits on x86-64 and gfx950 is checked directly.
// fn: 0 sin, 1 cos, 2 asin, 3 acos, 4 pow(x, y); 5 x / y and 6 sqrt(x) as the STRICT and EXACT kernels form them (integrator.inc.hip kdiv, ksqrt)
extern "C" __global__ void __launch_bounds__(256) kajo_kat_math(int fn, int n, const float* x, const float* y, float* out)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    float r;
    switch (fn) {
    case 0: r = kajo_sinf(x[i]); break;
    case 1: r = kajo_cosf(x[i]); break;
    case 2: r = kajo_asinf(x[i]); break;
    case 3: r = kajo_acosf(x[i]); break;
    case 5: r = kdiv(x[i], y[i]); break;
    case 6: r = ksqrt(x[i]); break;
    default: r = kajo_powf(x[i], y[i]); break;
    }
    out[i] = r;
}

extern "C" int kajo_kat_math_launch(int fn, int n, const void* x, const void* y, void* out, void* stream)
{
    hipLaunchKernelGGL(kajo_kat_math, dim3((n + 255) / 256), dim3(256), 0, static_cast<hipStream_t>(stream), fn, n,
                       static_cast<const float*>(x), static_cast<const float*>(y), static_cast<float*>(out));
    return (int)hipGetLastError();
}
