/*
 * kajo_hip.h -- C ABI of the MI355X rendering backend for Kajo (libkajo_hip.so).
 *
 * This is the drop-in boundary for ONE path of the reference: the per-pixel Monte-Carlo
 * integrator that cpu::Scheduler::run() drives through cpu::Renderer::render()
 * (renderer/cpu/Scheduler.cpp:60-85, renderer/cpu/Renderer.cpp:25-81). A backend in the
 * reference is a class with the constructor (const scene::Scene&, Image*, Preview*) and one
 * method run() (renderer/Scheduler.h:12-16, selected by name in renderer/Main.cpp:135-142);
 * kajo_amd/host/HipScheduler.{h,cpp} is that class for "-r hip", and everything it needs from
 * the GPU goes through the entry points below: plain pointers and sizes, int error codes, no
 * C++ / HIP / torch types in any signature.
 *
 * What each entry point replaces in the reference:
 *   kajo_hip_create          cpu::Renderer::Renderer + cpu::Scene::Scene (Renderer.cpp:17-23,
 *                            cpu/Scene.cpp:9-38): copy the scene, stage inverse matrices and
 *                            determinants, camera basis of Renderer.cpp:29-34
 *   kajo_hip_render          the pass loop of cpu::Renderer::render (Renderer.cpp:44-72) for
 *                            every pixel this handle owns, `passes` more passes, asynchronous
 *   kajo_hip_wait            joinTasks (cpu/Scheduler.cpp:44-51)
 *   kajo_hip_resolve_argb8   Renderer.cpp:73-75 + Image::linearToSRGB/colorToRGBA8
 *                            (renderer/Image.cpp:14-27) into Image::pixels (Image.h:18-20)
 *   kajo_hip_read_radiance   the radianceMap local of Renderer.cpp:36 (not observable in the
 *                            reference; exported here for parity tests)
 *   kajo_hip_counters        the samples/s bookkeeping of Preview::update (Preview.cpp:79-98)
 *   kajo_hip_destroy         the unique_ptr members of cpu::Scheduler (cpu/Scheduler.h:29-31)
 *
 * Pixels are dealt to GPUs as fixed-size tiles: a handle created with
 * (tileIndex, tileCount) accumulates the tiles t with t % tileCount == tileIndex in a compact
 * device buffer; kajo_hip_tile_buffer exposes it for the one RCCL gather per frame that the
 * host performs (torch.distributed or rccl directly), kajo_hip_compose places the gathered
 * buffers into the whole frame on the root handle. Every camera path draws from its own RNG
 * stream (include/kajo_stream.h), so the frame is bit-identical for any tiling / GPU count.
 *
 * Threading: a handle is not re-entrant; distinct handles are independent (section 8b).
 * All calls return 0 on success or a negative KAJO_E_* code; kajo_hip_last_error() gives the
 * message of the calling thread's most recent failure. No exception crosses this boundary.
 */
#ifndef KAJO_HIP_H
#define KAJO_HIP_H

#include <stddef.h>
#include <stdint.h>

#include "kajo_scene.h"

#ifdef __cplusplus
extern "C" {
#endif

#define KAJO_OK 0
#define KAJO_E_INVALID (-1)   /* bad argument */
#define KAJO_E_HIP (-2)       /* a HIP runtime call failed (message has the HIP error) */
#define KAJO_E_NO_DEVICE (-3) /* no usable GPU: the backend never falls back to the CPU */
#define KAJO_E_STATE (-4)     /* call not valid in the handle's current state */

/* KajoParams.flags */
#define KAJO_FLAG_FAST 0u     /* (the absence of KAJO_FLAG_STRICT and KAJO_FLAG_EXACT) fast numerics: hardware transcendentals, contracted
                                   multiply-adds. The fastest build; it does NOT meet BASELINE's per-pixel RMSE < 1e-4 on spheres.json at
                                   1920x1080 x 16 passes (6e-4: a few paths per million flip a hit / miss decision). Opt in by clearing
                                   the flags kajo_hip_default_params sets. */
#define KAJO_FLAG_STRICT 1u   /* strict numerics: bit-identical to the CPU oracle (slower). Like KAJO_FLAG_EXACT it forms the IEEE quotient
                                   and square root of the closest-hit walk by hand, which is exact for operands of ordinary size only:
                                   kajo_hip_create refuses a scene with a non-zero coordinate (object and camera transforms, radii) outside
                                   2^-40 .. 2^40 in magnitude with KAJO_E_INVALID under either flag */
#define KAJO_FLAG_COUNTERS 2u /* maintain device-side work counters */
#define KAJO_FLAG_NO_GRID 4u  /* always walk every sphere (no uniform grid for large scenes) */
#define KAJO_FLAG_NO_REORDER 8u /* dispatch workgroups in image order (no cost-sorted launch order) */
#define KAJO_FLAG_COOP 32u      /* EXPERIMENT, round 2 (measured slower, profiles/HISTORY.md section 8; only in kajo_amd/libkajo_hip_r02.so of `make
                                   experiments`, refused by the product library): the 8 waves of a workgroup pool their rays in LDS every
                                   trip, counting-sort them by kind and octant into a compact queue and walk that */
#define KAJO_FLAG_DEFERRED 64u  /* EXPERIMENT, round 3 (measured slower, profiles/HISTORY.md section 8; only in kajo_amd/libkajo_hip_exp.so of `make
                                   experiments`, refused by the product library): surviving vertices are parked in LDS and the light / BSDF
                                   sampling blocks run only in trips where enough lanes have one (deferred.inc.hip) */
#define KAJO_FLAG_NO_SPLIT 16u  /* small frames: do not let several waves share a pixel block and divide the passes; large frames
                                   (FAST / EXACT): do not render the cheapest blocks of a launch in parts (KajoCounters.tailGroups).
                                   Scheduling only: the frame is the same bit for bit with and without */
#define KAJO_FLAG_NO_SHADOW_LISTS 128u /* large scenes: shadow rays walk the uniform grid as extension rays do, instead of being answered
                                   from the lights' visibility lists inside the light loop (same results; for A/B runs and tests) */
#define KAJO_FLAG_EXACT 512u  /* decision-exact numerics (the default of kajo_hip_default_params): the oracle's arithmetic (KAJO_FLAG_STRICT's)
                                   wherever a value can reach a decision -- the closest-hit walk, hit points, normals, sampled directions,
                                   coins -- so every path meets the oracle's objects, draws its random numbers and ends in its generator
                                   state; the fast forms where a value only scales radiance (BSDF values and pdfs, the light pdf, MIS
                                   weights, throughput products). Tolerance against KAJO_FLAG_STRICT's (= the oracle's) buffer, asserted
                                   by the whole-frame tests: the same pixels are not-a-number; every other pixel's sum differs by at most
                                   KAJO_EXACT_REL_TOL of max(|oracle|, KAJO_EXACT_ABS_FLOOR * passes) per channel (measured: 9.2e-4 at
                                   worst, where the oracle's light pdf 1 - cos(asin(r / d)) cancels); clamped per-pixel RMSE of the
                                   estimate 6.5e-7 on spheres.json at 1920x1080 x 16 passes. Not with KAJO_FLAG_STRICT. */
#define KAJO_EXACT_REL_TOL 1.5e-3f
#define KAJO_EXACT_ABS_FLOOR 1e-3f
#define KAJO_FLAG_NO_ONE_LIGHT 256u /* small scenes with exactly one light, every numerics build: run the kernel instance of any number of lights
                                   (kajo_render_*_lights) instead of the one that samples the BSDF in the light's visit (same results in the
                                   FAST and EXACT builds bit for bit and in the STRICT build, which stays the oracle; the hold thresholds
                                   stay those of the scene's own instance; for A/B runs and tests) */

typedef struct KajoParams {
    int32_t samplesPerPass; /* S: nominal samples per pixel per pass (reference: 32, Renderer.cpp:21);
                               n = floor(sqrt(S)) strata per axis are traced, the sum is divided by S */
    int32_t depthLimit;     /* reference: 8 (Shader.cpp:24); 0 .. 1000 */
    uint64_t seed;          /* stream seed (reference constant 0715517 = 236367, Random.h:43) */
    uint32_t flags;         /* KAJO_FLAG_* */
    int32_t device;         /* HIP device ordinal */
    int32_t tileW, tileH;   /* tile size in pixels: multiples of 8 with tileW * tileH a multiple of 256; 0 => 64 x 16 */
    int32_t tileIndex;      /* this handle renders tiles t with t % tileCount == tileIndex */
    int32_t tileCount;      /* number of handles sharing the frame (GPUs); 0 => 1 */
    int32_t passesPerLaunch; /* passes fused into one kernel launch; 0 => library default */
} KajoParams;

typedef struct KajoCounters {
    uint64_t passes;         /* passes rendered so far */
    uint64_t paths;          /* camera paths traced = pixels * n^2 * passes */
    uint64_t traversals;     /* closest-hit scene walks (device counter; 0 without KAJO_FLAG_COUNTERS) */
    uint64_t vertices;       /* shaded path vertices (device counter) */
    uint64_t primitiveTests; /* traversals * (nPlanes + nSpheres): what walking every object costs; with
                                the uniform grid of large scenes the spheres actually tested are fewer */
    uint64_t laneSlots;      /* 64 * wave-iterations of the trace loop: traversals / laneSlots = lane efficiency */
    double kernelMs;         /* summed device time of the render kernels (HIP events on the handle's stream) */
    uint64_t launches;       /* render kernel launches */
    uint64_t shadowQueries;  /* large scenes: shadow rays answered from the lights' visibility lists inside the light loop (they
                                are not among `traversals`, which then counts camera and extension rays only) */
    uint64_t tailGroups;     /* of the last render launch: workgroups beyond one per pixel block -- the cheapest blocks of a large frame
                                of a small scene are rendered as one workgroup per group of four passes of the launch, so that the launch
                                ends on short jobs (FAST / EXACT, launches of 2 .. 8 whole groups; the frame is the same bit for bit;
                                0 = not parted) */
} KajoCounters;

typedef struct KajoHip* kajo_hip_t;

/* Fills *p with the reference's constants -- S = 32, depth 8, seed 236367, one tile set -- and flags = KAJO_FLAG_EXACT: the fastest
   numerics build that meets BASELINE's per-pixel RMSE < 1e-4 against the reference. */
void kajo_hip_default_params(KajoParams* p);

int kajo_hip_create(const KajoScene* scene, int width, int height, const KajoParams* params, kajo_hip_t* out);
int kajo_hip_destroy(kajo_hip_t h); /* NULL is accepted */

/* Enqueue `passes` more passes (pass numbers continue from the handle's count, first = 1; at most 2^31 - 1 in all).
   The frame after P passes is a function of the scene, the parameters and P alone -- not of how the passes were cut into render()
   calls, launches (passesPerLaunch), workgroups or GPUs. STRICT adds the passes' terms radiance / S to a pixel's total one by one, as
   the reference does (Renderer.cpp:70-71); FAST and EXACT (small scenes) add them in groups of four passes by absolute number
   (1-4, 5-8, ...): a group is summed from zero in pass order, then added to the total; a group in progress is added last. */
int kajo_hip_render(kajo_hip_t h, int passes);
int kajo_hip_wait(kajo_hip_t h);
/* Zero the accumulation and restart the pass numbering at 1. */
int kajo_hip_reset(kajo_hip_t h);
/* Continue a progressive session from pass `passesDone`: the next pass rendered is passesDone + 1 (the reference's loop
   `for (pass = 1;; pass++)`, Renderer.cpp:44, has no end). The call does not touch the accumulation buffer and DECLARES
   that it holds the sum of `passesDone` passes: kajo_hip_resolve_* divide by the pass count and kajo_hip_counters reports
   it, so the caller restores the buffer of the session being continued through kajo_hip_tile_buffer() first (or calls
   kajo_hip_reset() and set_pass_count(0)). FAST / EXACT: a passesDone inside a group of four continues from the buffer as
   one sum (the passes of the group so far are not known apart). Pass numbers run to 2^31 - 2. */
int kajo_hip_set_pass_count(kajo_hip_t h, int passesDone);

/* Whole-frame outputs; valid when tileCount == 1, or on a handle that has been composed.
   dst are HOST pointers: width*height uint32 ARGB8 (row 0 = top) / width*height*4 floats
   (sum over passes of radiance / S; divide by the pass count for the estimate). They wait for
   outstanding passes. */
int kajo_hip_resolve_argb8(kajo_hip_t h, uint32_t* dst);
int kajo_hip_read_radiance(kajo_hip_t h, float* dst);
/* Same, into DEVICE memory (e.g. a mapped preview buffer), asynchronous on the handle's stream. */
int kajo_hip_resolve_argb8_device(kajo_hip_t h, void* dst);

/* Multi-GPU plumbing. The tile buffer is ceil(T / tileCount) tiles of tileW*tileH float4 each
   (T = tiles in the frame), identical size on every rank so that one gather moves it. */
int kajo_hip_tile_buffer(kajo_hip_t h, void** devicePtr, size_t* bytes);
/* gathered = DEVICE pointer to tileCount consecutive tile buffers in rank order (what a gather
   to this rank produced); composes them into this handle's whole-frame buffer. With
   tileCount == 1 composition happens implicitly. */
int kajo_hip_compose(kajo_hip_t h, const void* gathered);

/* The ARGB8 image straight from gathered tile buffers (DEVICE pointer, tileCount consecutive buffers in rank order; NULL = this
   handle's own buffer when tileCount == 1), into DEVICE memory, asynchronous on the handle's stream: compose + resolve in
   one pass over the data, without the whole-frame float buffer (renderer/cpu/Renderer.cpp:70-75 per pixel, as
   kajo_hip_resolve_argb8_device). kajo_hip_compose stays for kajo_hip_read_radiance. */
int kajo_hip_resolve_gathered_argb8_device(kajo_hip_t h, const void* gathered, void* dst);

/* Use an existing HIP stream (hipStream_t passed as void*) instead of the handle's own. */
int kajo_hip_set_stream(kajo_hip_t h, void* stream);

int kajo_hip_counters(kajo_hip_t h, KajoCounters* out);

/* Host-only helper (no GPU needed): what create() stages from a scene -- inverse(16) +
   determinant per object, planes first then spheres, 17 floats each (cpu/Scene.cpp:9-13) and
   the camera basis p1, p2, p3, origin (Renderer.cpp:30-34), 12 floats. For tests. */
int kajo_hip_stage_scene(const KajoScene* scene, float* invDet17, float* basis12);

/* Host-only helper (no GPU needed), for tests: the per-light visibility lists create() stages for large scenes whose spheres
   are all world-space balls (kajo_amd/csrc/device_scene.h DShadowLists) -- what a shadow query tests instead of walking the grid.
   Returns the number of list items (0 = the scene gets no lists; negative = error). *binsPerAxis = n of the 6 x n x n cube-map
   bins per light, *nLights = emissive spheres; lightSphere[nLights] their sphere indices; start[nLights * 6 n^2 + 1],
   key[items], index[items] are filled when non-null (call once with nulls for the sizes). */
int kajo_hip_stage_shadow_lists(const KajoScene* scene, int32_t* binsPerAxis, int32_t* nLights, int32_t* lightSphere, uint32_t* start,
                                size_t startCapacity, float* key, uint32_t* index, size_t itemCapacity);

/* Host-only helper (no GPU needed), for tests: what create() decides about a scene's culling structures (kajo_amd/csrc/stage.cpp).
   closedRoom: the planes, all opaque, leave a BOUNDED convex region around the camera -- every ray of every path starts inside it;
   room[6] its bounding box widened to hold every sphere (min xyz, max xyz). grid: the uniform grid is built (>= 48 spheres, all of
   determinant 1, rigid planes); gridReach: rays starting farther than this from gridCenter walk every sphere instead (the grid's
   margins are sized for nearer ones); 0 = no limit (closed room). shadowLists: per-light visibility lists are built (they need the
   closed room: their margins are sized from its extent). */
typedef struct KajoStageInfo {
    int32_t closedRoom, grid, shadowLists, reserved;
    float room[6];
    float gridCenter[3], gridReach;
} KajoStageInfo;
int kajo_hip_stage_info(const KajoScene* scene, KajoStageInfo* out);

/* Host-only helper (no GPU needed), for tests: the order a handle dispatches its workgroups in (kajo_amd/csrc/launch_order.h), from the
   loop trips its first launch measured per wave -- waveTrips[nBlocks * wavesPerBlock] -- on a chip of `waveSlots` resident waves:
   blocks by cost, the most expensive first, and -- parts = 2 .. 8, the groups of four passes of a launch of FAST / EXACT kernels of a
   small scene -- the last *nParted (cheapest) blocks as `parts` consecutive workgroups of one group each. An order word is
   block | part << 28 | parted << 31. Returns the number of words (order may be NULL to ask for it); negative = error. */
int kajo_hip_launch_order(const uint32_t* waveTrips, uint32_t nBlocks, uint32_t wavesPerBlock, uint32_t waveSlots, int32_t parts, uint32_t* order,
                          size_t capacity, uint32_t* nParted);

/* Known-answer hooks: run the kernels' OWN device functions on caller-supplied rays, so that the
   vectors captured from the compiled reference (tests/golden/kat_trace.npz, kat_shade.npz) can be
   checked on the GPU function by function. All pointers are HOST memory; scenes whose hot records
   exceed 48 KiB of LDS are refused.
     kat_trace: closest hit of Raytracer::trace (renderer/cpu/Raytracer.cpp:126-138) per ray: object index
                (0 = miss, planes 1.., then spheres), ray.maxDistance, position, normal, tangent, binormal.
     kat_shade: trace + Shader::shade (renderer/cpu/Shader.cpp:113-178) of one path per ray from the given
                128-bit RNG state (no jitter draw), with the handle's depth limit: RGB and the RNG state
                afterwards (pins the number of draws). */
int kajo_hip_kat_trace(kajo_hip_t h, int n, const float* origins, const float* dirs, int32_t* objIndex, float* t,
                       float* position, float* normal, float* tangent, float* binormal);
int kajo_hip_kat_shade(kajo_hip_t h, int n, const float* origins, const float* dirs, const uint64_t* states, float* rgb,
                       uint64_t* finalStates);
/* include/kajo_strictmath.h evaluated on the device, element-wise: fn 0 sin, 1 cos, 2 asin, 3 acos, 4 pow(x, y); and the two IEEE
   operations the STRICT / EXACT kernels form by hand (the reference's `/` and glm's sqrt, renderer/cpu/Raytracer.cpp:30-44):
   fn 5 x / y, fn 6 sqrt(x). */
int kajo_hip_kat_strictmath(kajo_hip_t h, int fn, int n, const float* x, const float* y, float* out);

const char* kajo_hip_last_error(void);
const char* kajo_hip_version(void);

#ifdef __cplusplus
}
#endif

#endif /* KAJO_HIP_H */
