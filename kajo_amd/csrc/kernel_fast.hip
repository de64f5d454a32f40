// FAST numerics: hardware transcendentals and FMA contraction everywhere -- the fastest of the three builds (bench.py fast_mode; the plugin runs
// EXACT by default since round 5, kernel_exact.hip). Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=fast
#define KAJO_STRICT 0
#ifndef KAJO_WAVES_PER_SIMD
#define KAJO_WAVES_PER_SIMD 5 // the FAST loop fits 96 VGPRs without spills (tools/vgpr_check.sh)
#endif
#ifndef KAJO_ANY_LANE_GIVES
#define KAJO_ANY_LANE_GIVES 0 // (integrator.inc.hip: measured -0.3 % on configs[1], -2 % on configs[4] in this build, +0.4 ... +0.9 % in the other two: profiles/r06_notes.txt)
#endif
#define KAJO_KERNEL_NAME kajo_render_fast
#ifndef KAJO_PRESAMPLE
#define KAJO_PRESAMPLE 1
#endif
#if KAJO_PRESAMPLE
#define KAJO_KERNEL_NAME_LIGHTS kajo_render_fast_lights // small scenes with several lights (or none); kajo_render_fast: exactly one
#endif
#define KAJO_KERNEL_NAME_BIG kajo_render_fast_big
#define KAJO_KERNEL_NAME_BIGLIST kajo_render_fast_biglist
// (an instance per home of the grid's cell lists -- LDS: _lg, global memory: the plain names -- as the STRICT / EXACT builds have since round 3:
// one typed walk per kernel instead of both, round 6)
#define KAJO_KERNEL_NAME_BIG_LG kajo_render_fast_big_lg
#define KAJO_KERNEL_NAME_BIGLIST_LG kajo_render_fast_biglist_lg
#define KAJO_KERNEL_NAME_SPLIT kajo_render_fast_split
#define KAJO_KAT_SHADE_NAME kajo_kat_shade_fast
#define KAJO_KAT_TRACE_NAME kajo_kat_trace_fast
#define KAJO_RESOLVE_NAME kajo_resolve_fast
#define KAJO_RESOLVE_TILES_NAME kajo_resolve_tiles_fast
#include "integrator.inc.hip"
#include "launch.inc.hip"
