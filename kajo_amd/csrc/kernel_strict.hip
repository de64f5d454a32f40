// STRICT numerics: bit-identical to the CPU oracle. Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off
#define KAJO_STRICT 1
#define KAJO_KERNEL_NAME kajo_render_strict
#define KAJO_KERNEL_NAME_BIG kajo_render_strict_big
#define KAJO_KAT_SHADE_NAME kajo_kat_shade_strict
#define KAJO_KAT_TRACE_NAME kajo_kat_trace_strict
#define KAJO_RESOLVE_NAME kajo_resolve_strict
#include "integrator.inc.hip"
#include "launch.inc.hip"
