// EXACT numerics ("decision-exact", round 5): the STRICT build's arithmetic wherever a value can reach a decision -- every path meets
// the oracle's objects, draws the oracle's random numbers and ends in the oracle's generator state -- and the FAST build's forms for
// what only scales radiance (integrator.inc.hip, KAJO_RSTRICT). Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#define KAJO_STRICT 1
#define KAJO_EXACT 1
#ifndef KAJO_WAVES_PER_SIMD
#define KAJO_WAVES_PER_SIMD 5 // the EXACT loop of small scenes fits 96 VGPRs without vector spills (tools/kernel_resources.sh): +8 % over four waves
#endif
#ifndef KAJO_INLINE_SHADOW
#define KAJO_INLINE_SHADOW 1 // small scenes of several lights answer shadow rays inside the light loop, as the STRICT build does
#endif
#define KAJO_KERNEL_NAME kajo_render_exact
#define KAJO_KERNEL_NAME_LIGHTS kajo_render_exact_lights
#define KAJO_KERNEL_NAME_BIG kajo_render_exact_big
#define KAJO_KERNEL_NAME_BIGLIST kajo_render_exact_biglist
#define KAJO_KERNEL_NAME_BIG_LG kajo_render_exact_big_lg
#define KAJO_KERNEL_NAME_BIGLIST_LG kajo_render_exact_biglist_lg
#define KAJO_KERNEL_NAME_SPLIT kajo_render_exact_split
#define KAJO_KAT_SHADE_NAME kajo_kat_shade_exact
// (no known-answer walk and no resolve of its own: the walk is the STRICT build's instruction for instruction, and the image is
// resolved by the STRICT build's kernels)
#include "integrator.inc.hip"
#include "launch.inc.hip"
