// capi.cpp -- libkajo_hip.so: the C ABI of include/kajo_hip.h over the HIP runtime.
//
// One KajoHip handle = one GPU's share of a frame: the staged scene in device memory, the
// compact tile accumulation buffer, a stream, and (on demand) the composed whole frame and
// its ARGB8 image. There is NO CPU rendering path in this library: without a usable HIP
// device kajo_hip_create fails with KAJO_E_NO_DEVICE.
#include "kajo_hip.h"

#include <hip/hip_runtime_api.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <string>
#include <utility>
#include <vector>

#include "render_args.h"
#include "stage.h"
#include "launch_order.h"
#include "tuning.h"

constexpr int kMaxParts = 8; // workgroups a block of the launch tail is rendered in: one per group of a launch of 2 .. 8 groups

// launchers defined next to their kernels (kernel_fast.hip, kernel_strict.hip, kernel_exact.hip, aux_kernels.hip)
extern "C" {
int kajo_render_exact_launch(const RenderArgs*, int coldInLds, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_exact_split_launch(const RenderArgs*, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_exact_set_lds(int coldInLds, size_t lds);
int kajo_kat_shade_exact_launch(const RenderArgs*, unsigned grid, size_t lds, void* stream);
int kajo_render_fast_launch(const RenderArgs*, int coldInLds, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_strict_launch(const RenderArgs*, int coldInLds, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_fast_split_launch(const RenderArgs*, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_strict_split_launch(const RenderArgs*, unsigned grid, unsigned block, size_t lds, void* stream);
int kajo_render_fast_set_lds(int coldInLds, size_t lds);
int kajo_render_strict_set_lds(int coldInLds, size_t lds);
int kajo_resolve_fast_launch(const void* frame, int count, float passes, void* dst, void* stream);
int kajo_resolve_strict_launch(const void* frame, int count, float passes, void* dst, void* stream);
int kajo_resolve_tiles_fast_launch(const void* gathered, const TileMap* map, float passes, void* dst, void* stream);
int kajo_resolve_tiles_strict_launch(const void* gathered, const TileMap* map, float passes, void* dst, void* stream);
int kajo_compose_launch(const void* gathered, const TileMap* map, void* frame, void* stream);
int kajo_fold_parts_launch(void* tiles, const void* side, uint32_t sideStride, const uint32_t* blocks, unsigned count, unsigned threads, int parts, void* stream);
int kajo_kat_shade_fast_launch(const RenderArgs*, unsigned grid, size_t lds, void* stream);
int kajo_kat_shade_strict_launch(const RenderArgs*, unsigned grid, size_t lds, void* stream);
int kajo_kat_trace_fast_launch(const KatTraceArgs*, unsigned grid, size_t lds, void* stream);
int kajo_kat_trace_strict_launch(const KatTraceArgs*, unsigned grid, size_t lds, void* stream);
int kajo_kat_math_launch(int fn, int n, const void* x, const void* y, void* out, void* stream);
}

namespace
{

thread_local std::string g_error;

int fail(int code, const std::string& what)
{
    g_error = what;
    return code;
}

int failHip(hipError_t e, const char* what)
{
    return fail(KAJO_E_HIP, std::string(what) + ": " + hipGetErrorString(e));
}

#define HIP_TRY(expr)                                                                                                  \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess)                                                                                          \
            return failHip(e_, #expr);                                                                                 \
    } while (0)

template <class T>
hipError_t upload(const std::vector<T>& v, const T** out, std::vector<void*>& owned)
{
    *out = nullptr;
    const size_t bytes = (v.empty() ? 1 : v.size()) * sizeof(T);
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes);
    if (e != hipSuccess)
        return e;
    owned.push_back(p);
    if (!v.empty()) {
        e = hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice);
        if (e != hipSuccess)
            return e;
    }
    *out = static_cast<const T*>(p);
    return hipSuccess;
}

} // namespace

struct KajoHip
{
    int device = 0;
    hipStream_t stream = nullptr;
    bool ownStream = false;
    int W = 0, H = 0;
    KajoParams params{};
    kajo::StagedScene staged;
    DSceneView view{};
    std::vector<void*> sceneBuffers;
    TileMap map{};
    int tilesY = 0, nTiles = 0, nTilesOwned = 0, tilesPerOwner = 0;
    size_t tileBytes = 0;
    void* tiles = nullptr;   // float4[slotsPerOwner]
    void* frame = nullptr;   // float4[W*H], lazily
    void* argb = nullptr;    // uint32[W*H], lazily
    bool frameValid = false;
    unsigned long long* counters = nullptr; // device [4]
    // launch-order feedback (render_args.h): per-wave loop trips of the last launch, block order for the next
    uint32_t* waveTrips = nullptr;   // device [grid * 4]
    uint32_t* blockOrder = nullptr;  // device [grid]
    bool orderValid = false, tripsPending = false;
    unsigned gridBlocks = 0;
    // launch tail (updateBlockOrder / partTheTail): the cost-sorted order on the host, how many of its last (cheapest) blocks are rendered in
    // parts, those blocks, and per number of parts G = 2 .. 8 the order with each of them expanded into G workgroups (built on first use)
    std::vector<uint32_t> hostOrder;
    uint32_t* partedOrder[kMaxParts + 1] = {}; // device [gridBlocks + (G - 1) * nParted]
    uint32_t* partedBlocks = nullptr;               // device [nParted]
    unsigned nParted = 0;
    void* side = nullptr;        // float4 [kMaxParts - 1][nParted * block threads]: the later parts' group sums of one launch
    bool partsAllowed = false;   // FAST / EXACT handle of a small scene that orders its launches and may divide them
    // FAST / EXACT, small scenes (render_args.h): a launch that ends inside a group of four passes leaves the group so far and the total of
    // the complete groups here
    void* carry = nullptr;       // float4 [2][slotsPerOwner], on first need
    bool carryValid = false;     // ... and they are those of passesDone
    int waveSlots = 0;           // waves the chip holds at once with this handle's kernel (updateBlockOrder)
    unsigned lastTailGroups = 0; // KajoCounters.tailGroups
    unsigned wavesPerBlock = 1; // workgroup = 64 * wavesPerBlock threads: single-wave groups dispatch and retire
                                // independently (measured +2.3 % over 4-wave groups)
    int passesDone = 0;
    size_t ldsBytes = 0, hotBytes = 0;
    int stealWindow = 4; // render_args.h; 1 when a large scene needs the LDS for its grid
    int thrL = 1, holdTrips = 1; // integrator.inc.hip MODE_HOLD
    int ldsExtra = 0;    // (KAJO_TUNING builds only) unused bytes per wave, to study a launch at a lower occupancy
    int helpBytes = 0;   // list scenes: [64] owner lanes + [64] blocker flags of the cooperative list walk (integrator.inc.hip), behind the mailbox
    int accBytes = 0;    // FAST / EXACT, small scenes: [64] float4, the lanes' running totals behind the mailbox (integrator.inc.hip GROUPS)
    size_t perWaveBytes(bool withMailbox) const
    {
        return (size_t)ldsExtra + (size_t)helpBytes + (withMailbox ? (size_t)64 * stealWindow * 16 + (size_t)accBytes : 0);
    }
    void fillWaveLds(RenderArgs& a, size_t perWaveOffset, bool withMailbox) const
    {
        a.perWaveOffset = (uint32_t)perWaveOffset;
        a.perWaveBytes = (uint32_t)perWaveBytes(withMailbox);
        a.thrL = thrL;
        a.holdTrips = holdTrips;
    }
    int coldInLds = 1;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> pending; // kernel timing
    std::vector<hipEvent_t> eventPool;
    double kernelMs = 0.0;
    uint64_t launches = 0;

    // numerics build the handle runs (include/kajo_hip.h): 0 FAST, 1 STRICT, 2 EXACT
    int numerics() const { return (params.flags & KAJO_FLAG_STRICT) ? 1 : ((params.flags & KAJO_FLAG_EXACT) ? 2 : 0); }
    // the oracle's arithmetic in everything that decides (STRICT and EXACT): which walk, which hold policy, whose resolve
    bool strict() const { return numerics() != 0; }
    int launchRender(const RenderArgs* a, int home, unsigned grid, unsigned block, size_t lds) const
    {
        switch (numerics()) {
        case 1: return kajo_render_strict_launch(a, home, grid, block, lds, stream);
        case 2: return kajo_render_exact_launch(a, home, grid, block, lds, stream);
        default: return kajo_render_fast_launch(a, home, grid, block, lds, stream);
        }
    }
    int launchSplit(const RenderArgs* a, unsigned grid, unsigned block, size_t lds) const
    {
        switch (numerics()) {
        case 1: return kajo_render_strict_split_launch(a, grid, block, lds, stream);
        case 2: return kajo_render_exact_split_launch(a, grid, block, lds, stream);
        default: return kajo_render_fast_split_launch(a, grid, block, lds, stream);
        }
    }
    int setLds(size_t lds) const
    {
        switch (numerics()) {
        case 1: return kajo_render_strict_set_lds(coldInLds, lds);
        case 2: return kajo_render_exact_set_lds(coldInLds, lds);
        default: return kajo_render_fast_set_lds(coldInLds, lds);
        }
    }
};

namespace
{

int bind(KajoHip* h)
{
    HIP_TRY(hipSetDevice(h->device));
    return KAJO_OK;
}

int drainEvents(KajoHip* h)
{
    for (auto& pr : h->pending) {
        float ms = 0.f;
        HIP_TRY(hipEventElapsedTime(&ms, pr.first, pr.second));
        h->kernelMs += ms;
        h->eventPool.push_back(pr.first);
        h->eventPool.push_back(pr.second);
    }
    h->pending.clear();
    return KAJO_OK;
}

int getEvent(KajoHip* h, hipEvent_t* ev)
{
    if (!h->eventPool.empty()) {
        *ev = h->eventPool.back();
        h->eventPool.pop_back();
        return KAJO_OK;
    }
    HIP_TRY(hipEventCreate(ev));
    return KAJO_OK;
}

int ensureFrame(KajoHip* h)
{
    if (!h->frame)
        HIP_TRY(hipMalloc(&h->frame, (size_t)h->W * h->H * 16));
    return KAJO_OK;
}

// whole frame from this handle's own tiles (single-owner case)
int composeOwn(KajoHip* h)
{
    if (h->frameValid)
        return KAJO_OK;
    if (h->map.tileCount != 1)
        return fail(KAJO_E_STATE, "whole-frame output needs kajo_hip_compose() when tileCount > 1");
    int rc = ensureFrame(h);
    if (rc)
        return rc;
    HIP_TRY((hipError_t)kajo_compose_launch(h->tiles, &h->map, h->frame, h->stream));
    h->frameValid = true;
    return KAJO_OK;
}

void destroy(KajoHip* h)
{
    if (!h)
        return;
    (void)hipSetDevice(h->device);
    if (h->stream)
        (void)hipStreamSynchronize(h->stream);
    for (auto& pr : h->pending) {
        (void)hipEventDestroy(pr.first);
        (void)hipEventDestroy(pr.second);
    }
    for (hipEvent_t e : h->eventPool)
        (void)hipEventDestroy(e);
    for (void* p : h->sceneBuffers)
        (void)hipFree(p);
    if (h->tiles)
        (void)hipFree(h->tiles);
    if (h->frame)
        (void)hipFree(h->frame);
    if (h->argb)
        (void)hipFree(h->argb);
    if (h->counters)
        (void)hipFree(h->counters);
    if (h->waveTrips)
        (void)hipFree(h->waveTrips);
    if (h->blockOrder)
        (void)hipFree(h->blockOrder);
    for (uint32_t* o : h->partedOrder)
        if (o)
            (void)hipFree(o);
    if (h->partedBlocks)
        (void)hipFree(h->partedBlocks);
    if (h->side)
        (void)hipFree(h->side);
    if (h->carry)
        (void)hipFree(h->carry);
    if (h->ownStream && h->stream)
        (void)hipStreamDestroy(h->stream);
    delete h;
}

// The launch tail. Workgroups are dispatched in order as wave slots come free, so a launch ends while its last `waveSlots` jobs run out:
// on average half such a job per slot stands idle -- 3 % of a 1920x1080 launch (six rounds of the slots), 1 % at 3840x2160. The cheapest
// blocks, last in the order, are therefore rendered as one workgroup per GROUP of the launch's passes (integrator.inc.hip PARTS; a launch of
// 16 passes: four workgroups of four passes): the launch ends on short jobs. Short waves are the less efficient ones (a lane that has run
// out of passes can only take over whole ones: 16 -> 4 passes per wave costs 10 %, tools/ppl_sweep.py), so only the tail is parted. FAST
// and EXACT kernels of small scenes, whose totals take the passes in groups of four whoever renders them (integrator.inc.hip GROUPS):
// the frame does not change by a bit.
// Which blocks: decided once, when the order is known. How they are expanded depends on the number of groups of a launch: partedOrderFor.
int partTheTail(KajoHip* h)
{
    h->nParted = 0;
    const unsigned n = (unsigned)h->hostOrder.size();
    if (!h->partsAllowed || n >= (1u << 28))
        return KAJO_OK;
    if (!h->waveSlots) {
        int cus = 0;
        HIP_TRY(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
        h->waveSlots = cus * 4 * (h->coldInLds ? 5 : 4); // (launch bounds of the small-scene / large-scene kernels)
    }
    int q4 = 4; // how many blocks, in eighths of the slots (measured: tools/tail_sweep.sh)
    KAJO_TUNE_INT("KAJO_TAIL_Q4", 0, 64, q4);
    const unsigned nParted = kajoTailBlocks(n, (unsigned)h->waveSlots / h->wavesPerBlock, q4);
    if (nParted == 0)
        return KAJO_OK;
    const unsigned block = 64 * h->wavesPerBlock;
    for (uint32_t*& o : h->partedOrder) {
        if (o)
            (void)hipFree(o);
        o = nullptr;
    }
    if (h->partedBlocks)
        (void)hipFree(h->partedBlocks);
    if (h->side)
        (void)hipFree(h->side);
    h->partedBlocks = nullptr;
    h->side = nullptr;
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->partedBlocks), (size_t)nParted * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(h->partedBlocks, h->hostOrder.data() + (n - nParted), (size_t)nParted * sizeof(uint32_t), hipMemcpyHostToDevice));
    // the later parts' group sums of one launch: compact, a workgroup's worth of slots per parted block and part (18 MB at 1920x1080)
    HIP_TRY(hipMalloc(&h->side, (size_t)(kMaxParts - 1) * nParted * block * 16));
    h->nParted = nParted;
    return KAJO_OK;
}

// The order of a launch of `parts` groups: every block once, in cost order, the last nParted of them as `parts` consecutive workgroups.
int partedOrderFor(KajoHip* h, int parts)
{
    if (h->partedOrder[parts])
        return KAJO_OK;
    std::vector<uint32_t> parted;
    kajoPartedOrder(h->hostOrder, h->nParted, parts, parted);
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&h->partedOrder[parts]), parted.size() * sizeof(uint32_t)));
    HIP_TRY(hipMemcpy(h->partedOrder[parts], parted.data(), parted.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    return KAJO_OK;
}

// Longest-processing-time-first order of the workgroups from the trips the first launch recorded
// (a block runs as long as its slowest wave).
int updateBlockOrder(KajoHip* h)
{
    h->tripsPending = false;
    const unsigned n = h->gridBlocks;
    const unsigned w = h->wavesPerBlock;
    std::vector<uint32_t> trips((size_t)n * w);
    HIP_TRY(hipMemcpy(trips.data(), h->waveTrips, trips.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::vector<uint32_t> cost, order;
    kajoBlockCosts(trips.data(), n, w, cost);
    kajoCostOrder(cost, order);
    HIP_TRY(hipMemcpy(h->blockOrder, order.data(), n * sizeof(uint32_t), hipMemcpyHostToDevice));
    h->orderValid = true;
    h->hostOrder.swap(order);
    return partTheTail(h);
}

} // namespace

extern "C" {

const char* kajo_hip_last_error(void)
{
    return g_error.c_str();
}

const char* kajo_hip_version(void)
{
    return "kajo-hip 0.1 (gfx950)";
}

void kajo_hip_default_params(KajoParams* p)
{
    std::memset(p, 0, sizeof *p);
    p->samplesPerPass = 32;  // Renderer.cpp:21
    p->depthLimit = 8;       // Shader.cpp:24
    p->seed = 0715517;       // Random.h:43
    p->tileW = 64;
    p->tileH = 16;
    p->tileIndex = 0;
    p->tileCount = 1;
    p->passesPerLaunch = 0;
    p->flags = KAJO_FLAG_EXACT; // the fastest build that meets BASELINE's RMSE < 1e-4 against the reference (include/kajo_hip.h)
}

int kajo_hip_stage_scene(const KajoScene* scene, float* invDet17, float* basis12)
{
    if (!scene)
        return fail(KAJO_E_INVALID, "scene is null");
    kajo::StagedScene st;
    kajo::stageScene(*scene, st);
    if (invDet17)
        std::memcpy(invDet17, st.invDet.data(), st.invDet.size() * sizeof(float));
    if (basis12) {
        std::memcpy(basis12 + 0, st.p1, 12);
        std::memcpy(basis12 + 3, st.p2, 12);
        std::memcpy(basis12 + 6, st.p3, 12);
        std::memcpy(basis12 + 9, st.origin, 12);
    }
    return KAJO_OK;
}

int kajo_hip_stage_shadow_lists(const KajoScene* scene, int32_t* binsPerAxis, int32_t* nLights, int32_t* lightSphere, uint32_t* start,
                                size_t startCapacity, float* key, uint32_t* index, size_t itemCapacity)
{
    if (!scene || !binsPerAxis || !nLights)
        return fail(KAJO_E_INVALID, "null argument");
    kajo::StagedScene st;
    kajo::stageScene(*scene, st);
    *binsPerAxis = st.shadowEnabled ? st.shadowN : 0;
    *nLights = (int32_t)st.light.size();
    if (!st.shadowEnabled)
        return 0;
    if (lightSphere)
        std::memcpy(lightSphere, st.light.data(), st.light.size() * sizeof(int32_t));
    if (start) {
        if (startCapacity < st.shadowStart.size())
            return fail(KAJO_E_INVALID, "start array too small");
        std::memcpy(start, st.shadowStart.data(), st.shadowStart.size() * sizeof(uint32_t));
    }
    if (key || index) {
        if (itemCapacity < st.shadowItems.size())
            return fail(KAJO_E_INVALID, "item arrays too small");
        for (size_t i = 0; i < st.shadowItems.size(); i++) {
            if (key)
                key[i] = st.shadowItems[i].key;
            if (index)
                index[i] = st.shadowItems[i].index;
        }
    }
    if (st.shadowItems.size() > 0x7fffffffu)
        return fail(KAJO_E_INVALID, "too many items");
    return (int)st.shadowItems.size();
}

int kajo_hip_stage_info(const KajoScene* scene, KajoStageInfo* info)
{
    if (!scene || !info)
        return fail(KAJO_E_INVALID, "null argument");
    kajo::StagedScene st;
    kajo::stageScene(*scene, st);
    std::memset(info, 0, sizeof *info);
    info->closedRoom = st.roomClosed ? 1 : 0;
    info->grid = st.gridEnabled ? 1 : 0;
    info->shadowLists = st.shadowEnabled ? 1 : 0;
    for (int k = 0; k < 3; k++) {
        info->room[k] = (float)st.roomLo[k];
        info->room[3 + k] = (float)st.roomHi[k];
        info->gridCenter[k] = st.gridCenter[k];
    }
    info->gridReach = (st.gridEnabled && !st.roomClosed) ? std::sqrt(st.gridReach2) : 0.f;
    return KAJO_OK;
}

int kajo_hip_launch_order(const uint32_t* waveTrips, uint32_t nBlocks, uint32_t wavesPerBlock, uint32_t waveSlots, int32_t parts, uint32_t* order,
                          size_t capacity, uint32_t* nPartedOut)
{
    if (!waveTrips || wavesPerBlock < 1 || wavesPerBlock > 4 || parts < 1 || parts > kMaxParts || nBlocks >= (1u << 28))
        return fail(KAJO_E_INVALID, "invalid argument");
    std::vector<uint32_t> cost, plain, out;
    kajoBlockCosts(waveTrips, nBlocks, wavesPerBlock, cost);
    kajoCostOrder(cost, plain);
    const unsigned nParted = kajoTailBlocks(nBlocks, waveSlots / wavesPerBlock);
    if (nPartedOut)
        *nPartedOut = nParted;
    if (parts >= 2 && nParted)
        kajoPartedOrder(plain, nParted, parts, out);
    else
        out.swap(plain);
    if (out.size() > 0x7fffffffu)
        return fail(KAJO_E_INVALID, "order too long");
    if (order) {
        if (capacity < out.size())
            return fail(KAJO_E_INVALID, "order array too small");
        std::memcpy(order, out.data(), out.size() * sizeof(uint32_t));
    }
    return (int)out.size();
}

int kajo_hip_create(const KajoScene* scene, int width, int height, const KajoParams* params, kajo_hip_t* out)
{
    if (!scene || !params || !out)
        return fail(KAJO_E_INVALID, "null argument");
    *out = nullptr;
    if (width <= 0 || height <= 0 || (long long)width * height > (1ll << 31) - 1)
        return fail(KAJO_E_INVALID, "image size out of range");
    if (scene->nPlanes < 0 || scene->nSpheres < 0 || (scene->nPlanes && !scene->planes) || (scene->nSpheres && !scene->spheres))
        return fail(KAJO_E_INVALID, "scene arrays inconsistent");
    KajoParams p = *params;
    if (p.tileW == 0)
        p.tileW = 64;
    if (p.tileH == 0)
        p.tileH = 16;
    if (p.tileCount == 0)
        p.tileCount = 1;
    if (p.samplesPerPass < 1 || p.samplesPerPass > 65535)
        return fail(KAJO_E_INVALID, "samplesPerPass must be in [1, 65535]");
    if ((p.flags & KAJO_FLAG_STRICT) && (p.flags & KAJO_FLAG_EXACT))
        return fail(KAJO_E_INVALID, "the strict and the exact flag name two different numerics builds: set one");
    if (p.flags & KAJO_FLAG_COOP)
        return fail(KAJO_E_INVALID, "KAJO_FLAG_COOP: the cooperative-traversal experiment is not built into this library (make -C kajo_amd/csrc experiments)");
    if (p.flags & KAJO_FLAG_DEFERRED)
        return fail(KAJO_E_INVALID, "KAJO_FLAG_DEFERRED: the deferred-shading experiment is not built into this library (make -C kajo_amd/csrc experiments)");
    if (p.depthLimit < 0 || p.depthLimit > 1000) // (the reference's limit is 8)
        return fail(KAJO_E_INVALID, "depthLimit must be in [0, 1000]");
    if (p.tileW < 8 || p.tileH < 8 || (p.tileW & 7) || (p.tileH & 7) || (p.tileW * p.tileH) % 256)
        return fail(KAJO_E_INVALID, "tile size must be multiples of 8 with tileW*tileH a multiple of 256");
    if (p.tileCount < 1 || p.tileIndex < 0 || p.tileIndex >= p.tileCount)
        return fail(KAJO_E_INVALID, "tileIndex/tileCount out of range");

    if (p.flags & (KAJO_FLAG_STRICT | KAJO_FLAG_EXACT)) {
        // integrator.inc.hip kdiv / ksqrt: the IEEE quotient and root without the compiler's range scaling are exact while operands stay
        // dozens of binades inside the float range, which a scene of ordinary coordinates guarantees; outside it STRICT would silently
        // stop being the oracle
        float lo = 0.f, hi = 0.f;
        kajo::coordinateRange(*scene, &lo, &hi);
        if (!(hi == hi) || (hi > 0.f && (lo < 0x1p-40f || hi > 0x1p40f))) {
            char msg[256];
            std::snprintf(msg, sizeof msg, "strict / exact numerics need the scene's non-zero coordinates within 2^-40 .. 2^40 in magnitude "
                                           "(found %g .. %g): use the fast build or rescale the scene", (double)lo, (double)hi);
            return fail(KAJO_E_INVALID, msg);
        }
    }

    int nDev = 0;
    hipError_t e = hipGetDeviceCount(&nDev);
    if (e != hipSuccess || nDev <= 0)
        return fail(KAJO_E_NO_DEVICE, "no HIP device available; this backend has no CPU path");
    if (p.device < 0 || p.device >= nDev)
        return fail(KAJO_E_INVALID, "device ordinal out of range");

    KajoHip* h = new (std::nothrow) KajoHip;
    if (!h)
        return fail(KAJO_E_INVALID, "out of host memory");
    h->device = p.device;
    h->params = p;
    h->W = width;
    h->H = height;
#define CREATE_TRY(expr)                                                                                               \
    do {                                                                                                               \
        hipError_t e_ = (expr);                                                                                        \
        if (e_ != hipSuccess) {                                                                                        \
            destroy(h);                                                                                                \
            return failHip(e_, #expr);                                                                                 \
        }                                                                                                              \
    } while (0)
    CREATE_TRY(hipSetDevice(h->device));
    CREATE_TRY(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    h->ownStream = true;

    // ---- scene -----------------------------------------------------------------------------
    kajo::stageScene(*scene, h->staged, (p.flags & KAJO_FLAG_NO_GRID) ? 0 : 48, !(p.flags & KAJO_FLAG_NO_SHADOW_LISTS));
    const kajo::StagedScene& st = h->staged;
    DSceneView& v = h->view;
    CREATE_TRY(upload(st.planeRow, &v.planeRow, h->sceneBuffers));
    CREATE_TRY(upload(st.planeDet, &v.planeDet, h->sceneBuffers));
    CREATE_TRY(upload(st.planeFrame, &v.planeFrame, h->sceneBuffers));
    CREATE_TRY(upload(st.sphereHot, &v.sphereHot, h->sceneBuffers));
    CREATE_TRY(upload(st.sphereHotOffset, &v.sphereHotOffset, h->sceneBuffers));
    CREATE_TRY(upload(st.sphereCold, &v.sphereCold, h->sceneBuffers));
    CREATE_TRY(upload(st.material, &v.material, h->sceneBuffers));
    CREATE_TRY(upload(st.light, &v.light, h->sceneBuffers));
    {
        const uint32_t* cellStart = nullptr;
        const uint16_t* items = nullptr;
        CREATE_TRY(upload(st.gridCellStart, &cellStart, h->sceneBuffers));
        CREATE_TRY(upload(st.gridItems, &items, h->sceneBuffers));
        v.grid.enabled = st.gridEnabled;
        v.grid.nCells = st.gridEnabled ? (int32_t)st.gridCellStart.size() - 1 : 0;
        v.grid.nItems = (int32_t)st.gridItems.size();
        v.grid.inLds = 0;
        v.grid.cellStart = cellStart;
        v.grid.items = items;
        for (int k = 0; k < 3; k++) {
            v.grid.dim[k] = st.gridDim[k];
            v.grid.bmin[k] = st.gridEnabled ? st.gridMin[k] : 0.f;
            v.grid.bmax[k] = st.gridEnabled ? st.gridMax[k] : 0.f;
            v.grid.cell[k] = st.gridEnabled ? st.gridCell[k] : 1.f;
            v.grid.invCell[k] = st.gridEnabled ? 1.f / st.gridCell[k] : 1.f;
            v.grid.center[k] = st.gridCenter[k];
        }
        v.grid.reach2 = st.gridReach2;
    }
    {
        const uint32_t* rowBase = nullptr;
        const uint16_t* off16 = nullptr;
        const uint32_t* items = nullptr;
        const float* invKeyScale = nullptr;
        CREATE_TRY(upload(st.shadowRowBase, &rowBase, h->sceneBuffers));
        CREATE_TRY(upload(st.shadowOff16, &off16, h->sceneBuffers));
        CREATE_TRY(upload(st.shadowPacked, &items, h->sceneBuffers));
        CREATE_TRY(upload(st.shadowInvKeyScale, &invKeyScale, h->sceneBuffers));
        v.shadow.enabled = st.shadowEnabled ? 1 : 0;
        v.shadow.n = st.shadowN;
        v.shadow.rowBase = rowBase;
        v.shadow.off16 = off16;
        v.shadow.items = items;
        v.shadow.invKeyScale = invKeyScale;
    }
    v.nPlanes = st.nPlanes;
    v.nSpheres = st.nSpheres;
    v.nSphereHot = (int)st.sphereHot.size();
    v.nLights = (int)st.light.size();
    v.allTranslated = st.allTranslated;
    v.planesRigid = st.planesRigid;
    for (int i = 0; i < 3; i++) {
        v.background[i] = st.background[i];
        v.p1[i] = st.p1[i];
        v.dp2[i] = st.p2[i] - st.p1[i]; // (p2 - p1), (p3 - p1) of Renderer.cpp:58
        v.dp3[i] = st.p3[i] - st.p1[i];
        v.origin[i] = st.origin[i];
    }
    // LDS budget per workgroup (device_scene.h / integrator.inc.hip renderBody): hot records always,
    // cold records too while the total stays small enough for four workgroups per CU (160 KiB / 4).
    // (integrator.inc.hip stageToLds: the 4-byte arrays are padded to a 16-byte boundary before the light records)
    const size_t hotBytes = (size_t)v.nPlanes * 16 + (size_t)v.nSphereHot * 16 +
                            ((((size_t)v.nPlanes + (v.allTranslated ? 0 : v.nSpheres) + v.nLights) * 4 + 15) & ~(size_t)15) + (size_t)v.nLights * (64 + 16) +
                            ((((size_t)v.nLights * v.nPlanes) * 4 + 15) & ~(size_t)15) + 8 * 16;
    const size_t coldBytes = (size_t)v.nPlanes * 48 + (size_t)v.nSpheres * 64 + (size_t)(v.nPlanes + v.nSpheres) * sizeof(DMaterial);
    // What every wave adds to the scene copy (render_args.h): the mailbox of taken-over passes.
    // The sizes below are constants of the product library. A -DKAJO_TUNING build (libkajo_hip_tune.so, tools/ only) reads
    // overrides from the environment (tuning.h); libkajo_hip.so contains no getenv.
    const bool big = st.gridEnabled || hotBytes + coldBytes > 40 * 1024;
    h->stealWindow = 4;
    h->helpBytes = st.shadowEnabled ? 512 : 0;
    KAJO_TUNE_INT("KAJO_STEAL_WINDOW", 1, 16, h->stealWindow);
    KAJO_TUNE_INT("KAJO_LDS_EXTRA", 0, 64 * 1024, h->ldsExtra);
    h->ldsExtra &= ~15;
    // integrator.inc.hip MODE_HOLD: lanes that must want the light / BSDF blocks before they run without any lane having
    // waited a trip; 1 = every trip. Large scenes run them every trip (16 lights: most lanes are in them anyway).
    h->thrL = big ? 1 : (h->strict() ? 28 : 20);
    h->holdTrips = 1;
    if (!big && h->strict()) {
        // the STRICT loop of small scenes with several lights walks its shadow rays inside the light loop (KAJO_INLINE_SHADOW): a heavier
        // block, worth waiting longer for (three lights: 11.6 -> 13.4 G paths/s at 48 lanes / three trips). One light (its own instance:
        // one visit per vertex, the shadow ray in a trip of its own): 20.5-20.7 at 32-44 lanes / two trips (profiles/r04_presample.txt).
        h->thrL = v.nLights > 1 ? 48 : 36;
        h->holdTrips = v.nLights > 1 ? 3 : 2;
    }
    if (st.shadowEnabled) {
        // Large scenes with visibility lists: the light loop runs to its end inside one trip (16 lights: ~10 rounds of light
        // sample + shadow query) and is the expensive block of a trip, with a third of the lanes in it. It runs when 60 lanes
        // have a vertex waiting or it has been put off six trips in a row; the walk loses lanes to the waiting (lane
        // efficiency 0.975 -> 0.64) -- lanes without a ray skip the grid walk, so that costs the walk nothing but the slots -- and
        // the launch gains: FAST 2.35 -> 4.34 G paths/s on the 1000-sphere scene at 4K x 32 passes with 48 lanes / three trips,
        // 5.38 -> 5.54 from there to 60 / six once the idle lanes stopped walking stale rays (profiles/r04_c5_notes.txt).
        h->thrL = 60;
        h->holdTrips = 6;
    }
    KAJO_TUNE_INT("KAJO_THR_L", 1, 65, h->thrL);
    KAJO_TUNE_INT("KAJO_HOLD_TRIPS", 1, 16, h->holdTrips);
    size_t gridBytes = 0;
    const size_t gridHeaderBytes = st.gridEnabled ? 5 * 16 : 0; // always in LDS (integrator.inc.hip gridWalk)
    if (st.gridEnabled) {
        gridBytes = ((st.gridCellStart.size() * sizeof(uint32_t) + st.gridItems.size() * sizeof(uint16_t)) + 15) & ~(size_t)15;
        // The DDA reads a cell record and an item per step, each a dependent load: ~64 cycles from LDS, ~500 from L2. But the
        // walk is latency-bound and wants its workgroups per CU (measured on the 1000-sphere scene in round 2: the grid in LDS
        // at three workgroups per CU is 12 % SLOWER than the grid in L2 at four), so the grid moves into LDS only while hot
        // records + grid + the four waves' areas stay within the limit.
        int gridLimit = 40 * 1024;
        KAJO_TUNE_INT("KAJO_GRID_LDS_LIMIT", 0, 160 * 1024, gridLimit); // bytes
        // ... with the mailboxes shrunk to a one-pass steal window if need be
        const int wanted = h->stealWindow;
        for (int window : {4, 2, 1}) {
            if (window > wanted)
                continue;
            h->stealWindow = window;
            if (hotBytes + gridHeaderBytes + gridBytes + 4 * h->perWaveBytes(true) <= (size_t)gridLimit) {
                v.grid.inLds = 1;
                break;
            }
        }
        if (!v.grid.inLds) {
            h->stealWindow = wanted;
            gridBytes = 0;
        }
    }
    h->hotBytes = hotBytes + gridHeaderBytes + gridBytes; // what the big-scene staging (and the known-answer kernels) put in LDS
    h->coldInLds = !big;
    if (h->coldInLds && h->numerics() != 1) {
        // (the lanes' running totals take the room of one pass of the mailbox: three passes to take over instead of four costs nothing,
        // tools/steal_window_sweep.sh, and the scene copy + a wave's area of BASELINE's scenes stays within a fifth wave per SIMD's share)
        h->accBytes = 64 * 16;
        h->stealWindow = 3;
        KAJO_TUNE_INT("KAJO_STEAL_WINDOW", 1, 16, h->stealWindow);
    }
    h->ldsBytes = hotBytes + (h->coldInLds ? coldBytes : 0) + gridHeaderBytes + gridBytes;
    // every workgroup stages its own LDS copy of the scene: single-wave groups only while that copy is small
    h->wavesPerBlock = h->ldsBytes <= 6 * 1024 ? 1 : 4;
    {
        int w = 0;
        KAJO_TUNE_INT("KAJO_WAVES_PER_BLOCK", 1, 4, w); // 1, 2 or 4
        if (w == 1 || w == 2 || w == 4)
            h->wavesPerBlock = (unsigned)w;
    }
    // the one check, with the final values: scene copy + the waves' areas must fit a CU
    if (((h->ldsBytes + 15) & ~(size_t)15) + (size_t)h->wavesPerBlock * h->perWaveBytes(true) > 160 * 1024) {
        destroy(h);
        return fail(KAJO_E_INVALID, "scene exceeds the LDS staging limit: scene records + the waves' mailboxes must fit 160 KiB");
    }
    // ---- tiles -----------------------------------------------------------------------------
    TileMap& m = h->map;
    m.W = width;
    m.H = height;
    m.tileW = p.tileW;
    m.tileH = p.tileH;
    m.tilesX = (width + p.tileW - 1) / p.tileW;
    m.tileCount = p.tileCount;
    h->tilesY = (height + p.tileH - 1) / p.tileH;
    h->nTiles = m.tilesX * h->tilesY;
    h->tilesPerOwner = (h->nTiles + p.tileCount - 1) / p.tileCount;
    h->nTilesOwned = (h->nTiles - p.tileIndex + p.tileCount - 1) / p.tileCount;
    if (h->nTilesOwned < 0)
        h->nTilesOwned = 0;
    m.slotsPerOwner = h->tilesPerOwner * p.tileW * p.tileH;
    h->tileBytes = (size_t)m.slotsPerOwner * 16;
    // (FAST / EXACT handles of small scenes that order their launches render the tail of a launch in parts: partTheTail)
    h->partsAllowed = h->coldInLds && h->numerics() != 1 && !(p.flags & (KAJO_FLAG_NO_SPLIT | KAJO_FLAG_NO_REORDER));
    CREATE_TRY(hipMalloc(&h->tiles, h->tileBytes));
    CREATE_TRY(hipMemsetAsync(h->tiles, 0, h->tileBytes, h->stream));
    if (p.flags & KAJO_FLAG_COUNTERS) {
        CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->counters), 32 * sizeof(unsigned long long)));
        CREATE_TRY(hipMemsetAsync(h->counters, 0, 32 * sizeof(unsigned long long), h->stream));
    }
    {
        const int wavesPerTile = (p.tileW / 8) * (p.tileH / 8);
        h->gridBlocks = (unsigned)((long long)h->nTilesOwned * wavesPerTile / h->wavesPerBlock);
        if ((long long)h->nTilesOwned * wavesPerTile / h->wavesPerBlock >= (1ll << 28)) { // (render_args.h: an order word has 28 bits for the block)
            destroy(h);
            return fail(KAJO_E_INVALID, "frame too large: 2^28 pixel blocks per handle at most");
        }
        if (h->gridBlocks && !(p.flags & KAJO_FLAG_NO_REORDER)) {
            CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->waveTrips), (size_t)h->gridBlocks * h->wavesPerBlock * sizeof(uint32_t)));
            CREATE_TRY(hipMalloc(reinterpret_cast<void**>(&h->blockOrder), (size_t)h->gridBlocks * sizeof(uint32_t)));
        }
    }
    {
        const size_t ldsTotal = ((h->ldsBytes + 15) & ~(size_t)15) + (size_t)h->wavesPerBlock * h->perWaveBytes(true);
        if (ldsTotal > 48 * 1024) {
            CREATE_TRY((hipError_t)h->setLds(ldsTotal));
        }
    }
    CREATE_TRY(hipStreamSynchronize(h->stream));
#undef CREATE_TRY
    *out = h;
    return KAJO_OK;
}

int kajo_hip_destroy(kajo_hip_t h)
{
    destroy(h);
    return KAJO_OK;
}

int kajo_hip_set_stream(kajo_hip_t h, void* stream)
{
    if (!h)
        return fail(KAJO_E_INVALID, "null handle");
    int rc = bind(h);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->ownStream)
        HIP_TRY(hipStreamDestroy(h->stream));
    h->stream = static_cast<hipStream_t>(stream);
    h->ownStream = false;
    return KAJO_OK;
}

int kajo_hip_render(kajo_hip_t h, int passes)
{
    if (!h)
        return fail(KAJO_E_INVALID, "null handle");
    // (the kernels form the exclusive end of a launch's passes, firstPass + nPasses, in 32-bit integers: the last pass number a
    // handle can render is 2^31 - 2)
    if (passes < 0 || h->passesDone + (long long)passes > 0x7ffffffell)
        return fail(KAJO_E_INVALID, "pass count out of range (pass numbers run to 2^31 - 2)");
    int rc = bind(h);
    if (rc)
        return rc;
    if (passes == 0 || h->nTilesOwned == 0) {
        h->passesDone += passes;
        return KAJO_OK;
    }
    const KajoParams& p = h->params;
    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    a.scene = h->view;
    a.tiles = h->tiles;
    a.W = h->W;
    a.H = h->H;
    a.n = (int)std::sqrt((double)(unsigned)p.samplesPerPass); // Renderer.cpp:38
    a.S = (float)(unsigned)p.samplesPerPass;
    a.pixelWidth = 1.f / h->W;   // Renderer.cpp:39-42
    a.pixelHeight = 1.f / h->H;
    a.sampleWidth = a.pixelWidth / a.n;
    a.sampleHeight = a.pixelHeight / a.n;
    a.depthLimit = p.depthLimit;
    a.seed = p.seed;
    a.tileW = p.tileW;
    a.tileH = p.tileH;
    a.tilesX = h->map.tilesX;
    a.tilesY = h->tilesY;
    a.tileIndex = p.tileIndex;
    a.tileCount = p.tileCount;
    a.nTilesOwned = h->nTilesOwned;
    a.counters = h->counters;
    a.mailboxOffset = (uint32_t)((h->ldsBytes + 15) & ~(size_t)15);
    a.stealWindow = h->stealWindow;

    const unsigned block = 64 * h->wavesPerBlock;
    const unsigned grid = h->gridBlocks;
    a.blockOrder = h->orderValid ? h->blockOrder : nullptr;
    a.waveTrips = (h->waveTrips && !h->orderValid) ? h->waveTrips : nullptr; // measure once, on the first launch
    // scene copy + every wave's mailbox
    h->fillWaveLds(a, a.mailboxOffset, true);
    const size_t ldsTotal = a.mailboxOffset + (size_t)h->wavesPerBlock * a.perWaveBytes;
    const int perLaunch = p.passesPerLaunch > 0 ? p.passesPerLaunch : 16;
    int left = passes;
    while (left > 0) {
        const int now = left < perLaunch ? left : perLaunch;
        a.firstPass = h->passesDone + 1;
        a.nPasses = now;
        // (integrator.inc.hip GROUPS, FAST / EXACT kernels of small scenes: the total takes the passes in groups of four by their absolute
        // numbers. A launch that begins or ends inside a group hands the group over through `carry`: render_args.h)
        const bool grouped = h->coldInLds && h->numerics() != 1;
        const bool startsInside = grouped && h->passesDone % KAJO_GROUP_PASSES != 0, endsInside = grouped && (h->passesDone + now) % KAJO_GROUP_PASSES != 0;
        if ((startsInside || endsInside) && !h->carry)
            HIP_TRY(hipMalloc(&h->carry, 2 * h->tileBytes));
        a.carry = h->carry;
        a.carrySlots = (uint32_t)h->map.slotsPerOwner;
        // (a group whose first passes this handle did not render -- kajo_hip_set_pass_count to the middle of one -- continues from the
        // buffer as it stands: the restored sum counts as complete groups)
        a.carryIn = startsInside && h->carryValid;
        a.carryOut = endsInside;
        const int launchGroups = (!startsInside && !endsInside) ? now / KAJO_GROUP_PASSES : 0; // whole groups, or 0
        hipEvent_t e0, e1;
        if ((rc = getEvent(h, &e0)) || (rc = getEvent(h, &e1)))
            return rc;
        HIP_TRY(hipEventRecord(e0, h->stream));
        // Small frames: fewer pixel blocks than a few rounds of the chip's 4096 wave slots. 2 or 4 waves then share a
        // block and divide the passes of the launch (when they divide evenly); the per-pass terms meet in LDS.
        unsigned split = 1;
        const unsigned long long pixelBlocks = (unsigned long long)grid * h->wavesPerBlock;
        if (h->coldInLds && !(p.flags & KAJO_FLAG_NO_SPLIT)) {
            // measured (tools/size_sweep.py with KAJO_SPLIT=1..16, 256x144 ... 1920x1080): frames of fewer than three
            // rounds of the 4096 wave slots run best with the largest power of two -- up to 16 waves per block, as far
            // as the passes divide -- that keeps the launch within 8 rounds: many short waves pack the tail of the
            // launch better than few long ones. From 1280x720 on the unsplit kernel is 3-8 % faster.
            while (pixelBlocks < 3 * 4096 && split < 16 && now % (int)(split * 2) == 0 && pixelBlocks * split * 2 <= 8 * 4096)
                split *= 2;
            int v = 0;
            KAJO_TUNE_INT("KAJO_SPLIT", 1, 16, v);
            if (v >= 1 && (v & (v - 1)) == 0 && now % v == 0)
                split = (unsigned)v;
        }
        while (split > 1 && a.mailboxOffset + (size_t)now * 64 * 16 + split * h->perWaveBytes(false) > 48 * 1024)
            split /= 2; // the table and the waves' areas would need the large-LDS opt-in: not worth it
        // Launches of FEW passes (BASELINE configs[0] is one pass of 16 samples on 1024 pixel blocks: a quarter of the chip's SIMDs,
        // one wave each): the waves of a block divide the SAMPLES of every pass instead -- `chunks` per pass, now * chunks waves
        // per block -- and the paths' radiances meet in the table [pass][sample][pixel]. Chosen when it puts more waves on a
        // block than dividing the passes does.
        unsigned chunks = 1;
        if (h->coldInLds && !(p.flags & KAJO_FLAG_NO_SPLIT) && pixelBlocks < 3 * 4096) {
            const unsigned nn = (unsigned)(a.n * a.n);
            // the smallest division that gives the launch one round of the chip's wave slots (measured on configs[0], 1024 blocks:
            // 4 chunks 18.0, 8 chunks 17.4, 16 chunks 15.5 G paths/s against 8.3 undivided; profiles/r03_configs.txt)
            for (unsigned q = 2; q <= nn && (unsigned)now * q <= 16; q++)
                if (nn % q == 0 && pixelBlocks * now * q <= 8 * 4096 && a.mailboxOffset + (size_t)now * nn * 64 * 16 + (size_t)now * q * h->perWaveBytes(false) <= 48 * 1024) {
                    chunks = q;
                    if (pixelBlocks * now * q >= 4096)
                        break;
                }
            int v = 0;
            KAJO_TUNE_INT("KAJO_SAMPLE_CHUNKS", 1, 16, v); // (held to the same 48 KiB bound as the automatic choice, the waves' areas included)
            if (v >= 1 && nn % (unsigned)v == 0 && (unsigned)now * v <= 16 &&
                a.mailboxOffset + (size_t)now * nn * 64 * 16 + (size_t)now * v * h->perWaveBytes(false) <= 48 * 1024)
                chunks = (unsigned)v;
            if ((unsigned)now * chunks <= split)
                chunks = 1;
        }
        hipError_t le;
        h->lastTailGroups = 0;
        if (chunks > 1) {
            RenderArgs b = a;
            b.blockOrder = nullptr;
            b.waveTrips = nullptr;
            b.sampleChunks = (int32_t)chunks;
            const unsigned waves = (unsigned)now * chunks;
            h->fillWaveLds(b, a.mailboxOffset + (size_t)now * a.n * a.n * 64 * 16, false); // behind the [pass][sample][pixel] table
            b.thrL = 1; // (short waves: holding a vertex only lengthens their tail -- configs[0] 18.9 against 16.5 G paths/s)
            const size_t ldsSplit = b.perWaveOffset + (size_t)waves * b.perWaveBytes;
            le = (hipError_t)h->launchSplit(&b, (unsigned)pixelBlocks, 64 * waves, ldsSplit);
        } else if (split > 1) {
            RenderArgs b = a;
            b.blockOrder = nullptr; // one round or two: the launch order does not matter
            b.waveTrips = nullptr;
            h->fillWaveLds(b, a.mailboxOffset + (size_t)now * 64 * 16, false); // behind the [pass][pixel] term table
            b.thrL = 1; // (as above: 1-3 % on frames below 720p)
            const size_t ldsSplit = b.perWaveOffset + (size_t)split * b.perWaveBytes;
            le = (hipError_t)h->launchSplit(&b, (unsigned)pixelBlocks, 64 * split, ldsSplit);
        } else {
            // (coldInLds 2: the small-scene instance of any number of lights although the scene has one, KAJO_FLAG_NO_ONE_LIGHT)
            const int home = (h->coldInLds && (h->params.flags & KAJO_FLAG_NO_ONE_LIGHT)) ? 2 : h->coldInLds;
            // the launch tail: the cheapest blocks as one workgroup per group of the launch (partTheTail)
            const bool parted = grouped && h->orderValid && h->nParted && launchGroups >= 2 && launchGroups <= kMaxParts;
            if (parted) {
                if ((rc = partedOrderFor(h, launchGroups))) {
                    h->eventPool.push_back(e0);
                    h->eventPool.push_back(e1);
                    return rc;
                }
                h->lastTailGroups = h->nParted * (unsigned)(launchGroups - 1);
                RenderArgs b = a;
                b.blockOrder = h->partedOrder[launchGroups];
                b.side = h->side;
                b.sideStride = h->nParted * block;
                b.partedFirst = grid - h->nParted;
                le = (hipError_t)h->launchRender(&b, home, grid + h->lastTailGroups, block, ldsTotal);
                if (le == hipSuccess)
                    le = (hipError_t)kajo_fold_parts_launch(h->tiles, h->side, b.sideStride, h->partedBlocks, h->nParted, block, launchGroups, h->stream);
            } else {
                le = (hipError_t)h->launchRender(&a, home, grid, block, ldsTotal);
            }
        }
        if (le != hipSuccess) {
            h->eventPool.push_back(e0);
            h->eventPool.push_back(e1);
            return failHip(le, "render kernel launch");
        }
        HIP_TRY(hipEventRecord(e1, h->stream));
        h->pending.emplace_back(e0, e1);
        if (a.waveTrips && split == 1 && chunks == 1) {
            h->tripsPending = true;
            a.waveTrips = nullptr; // later launches of this call keep the first measurement
        }
        h->launches++;
        h->passesDone += now;
        h->carryValid = endsInside;
        left -= now;
    }
    h->frameValid = false;
    return KAJO_OK;
}

int kajo_hip_wait(kajo_hip_t h)
{
    if (!h)
        return fail(KAJO_E_INVALID, "null handle");
    int rc = bind(h);
    if (rc)
        return rc;
    HIP_TRY(hipStreamSynchronize(h->stream));
    if (h->tripsPending && (rc = updateBlockOrder(h)))
        return rc;
    return drainEvents(h);
}

int kajo_hip_reset(kajo_hip_t h)
{
    if (!h)
        return fail(KAJO_E_INVALID, "null handle");
    int rc = kajo_hip_wait(h);
    if (rc)
        return rc;
    HIP_TRY(hipMemsetAsync(h->tiles, 0, h->tileBytes, h->stream));
    if (h->counters)
        HIP_TRY(hipMemsetAsync(h->counters, 0, 32 * sizeof(unsigned long long), h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    h->passesDone = 0;
    h->carryValid = false;
    h->frameValid = false;
    h->kernelMs = 0.0;
    h->launches = 0;
    return KAJO_OK;
}

int kajo_hip_set_pass_count(kajo_hip_t h, int passesDone)
{
    if (!h || passesDone < 0 || passesDone > 0x7ffffffe)
        return fail(KAJO_E_INVALID, "invalid argument");
    int rc = kajo_hip_wait(h);
    if (rc)
        return rc;
    h->passesDone = passesDone;
    h->carryValid = false; // (a group in progress is not known apart from the buffer the caller declares: include/kajo_hip.h)
    h->frameValid = false;
    return KAJO_OK;
}

int kajo_hip_tile_buffer(kajo_hip_t h, void** devicePtr, size_t* bytes)
{
    if (!h || !devicePtr || !bytes)
        return fail(KAJO_E_INVALID, "null argument");
    *devicePtr = h->tiles;
    *bytes = h->tileBytes;
    return KAJO_OK;
}

int kajo_hip_compose(kajo_hip_t h, const void* gathered)
{
    if (!h || !gathered)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = ensureFrame(h)))
        return rc;
    HIP_TRY((hipError_t)kajo_compose_launch(gathered, &h->map, h->frame, h->stream));
    h->frameValid = true;
    return KAJO_OK;
}

int kajo_hip_resolve_gathered_argb8_device(kajo_hip_t h, const void* gathered, void* dst)
{
    if (!h || !dst)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc)
        return rc;
    if (h->passesDone < 1)
        return fail(KAJO_E_STATE, "nothing rendered yet");
    if (!gathered) {
        if (h->map.tileCount != 1)
            return fail(KAJO_E_STATE, "a handle that owns part of the frame needs the gathered tile buffers");
        gathered = h->tiles;
    }
    hipError_t le = (hipError_t)(h->strict() ? kajo_resolve_tiles_strict_launch(gathered, &h->map, (float)h->passesDone, dst, h->stream)
                                             : kajo_resolve_tiles_fast_launch(gathered, &h->map, (float)h->passesDone, dst, h->stream));
    if (le != hipSuccess)
        return failHip(le, "resolve kernel launch");
    return KAJO_OK;
}

int kajo_hip_resolve_argb8_device(kajo_hip_t h, void* dst)
{
    if (!h || !dst)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc)
        return rc;
    if (h->passesDone < 1)
        return fail(KAJO_E_STATE, "nothing rendered yet");
    // one owner and no composed frame at hand: resolve straight from the tile buffer (the frame is composed when somebody
    // asks for the float radiance)
    if (!h->frameValid && h->map.tileCount == 1)
        return kajo_hip_resolve_gathered_argb8_device(h, nullptr, dst);
    if ((rc = composeOwn(h)))
        return rc;
    const int count = h->W * h->H;
    hipError_t le = (hipError_t)(h->strict() ? kajo_resolve_strict_launch(h->frame, count, (float)h->passesDone, dst, h->stream)
                                             : kajo_resolve_fast_launch(h->frame, count, (float)h->passesDone, dst, h->stream));
    if (le != hipSuccess)
        return failHip(le, "resolve kernel launch");
    return KAJO_OK;
}

int kajo_hip_resolve_argb8(kajo_hip_t h, uint32_t* dst)
{
    if (!h || !dst)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc)
        return rc;
    const size_t bytes = (size_t)h->W * h->H * 4;
    if (!h->argb)
        HIP_TRY(hipMalloc(&h->argb, bytes));
    if ((rc = kajo_hip_resolve_argb8_device(h, h->argb)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dst, h->argb, bytes, hipMemcpyDeviceToHost, h->stream));
    return kajo_hip_wait(h);
}

int kajo_hip_read_radiance(kajo_hip_t h, float* dst)
{
    if (!h || !dst)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc)
        return rc;
    if ((rc = composeOwn(h)))
        return rc;
    HIP_TRY(hipMemcpyAsync(dst, h->frame, (size_t)h->W * h->H * 16, hipMemcpyDeviceToHost, h->stream));
    return kajo_hip_wait(h);
}

namespace
{

struct DeviceBuffer // scratch for the known-answer entry points
{
    void* p = nullptr;
    ~DeviceBuffer()
    {
        if (p)
            (void)hipFree(p);
    }
    hipError_t alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1); }
};

} // namespace

int kajo_hip_kat_trace(kajo_hip_t h, int n, const float* origins, const float* dirs, int32_t* objIndex, float* t,
                       float* position, float* normal, float* tangent, float* binormal)
{
    if (!h || n < 0 || !origins || !dirs || !objIndex || !t || !position || !normal || !tangent || !binormal)
        return fail(KAJO_E_INVALID, "null argument");
    if (h->hotBytes > 48 * 1024)
        return fail(KAJO_E_INVALID, "known-answer entry points are limited to scenes whose hot records fit 48 KiB of LDS");
    int rc = bind(h);
    if (rc || n == 0)
        return rc;
    std::vector<float> rays(6 * (size_t)n);
    for (int i = 0; i < n; i++) {
        std::memcpy(&rays[6 * i], origins + 3 * i, 12);
        std::memcpy(&rays[6 * i + 3], dirs + 3 * i, 12);
    }
    DeviceBuffer dRays, dIdx, dOut;
    HIP_TRY(dRays.alloc(rays.size() * 4));
    HIP_TRY(dIdx.alloc((size_t)n * 4));
    HIP_TRY(dOut.alloc((size_t)n * 13 * 4));
    HIP_TRY(hipMemcpyAsync(dRays.p, rays.data(), rays.size() * 4, hipMemcpyHostToDevice, h->stream));
    KatTraceArgs a;
    a.scene = h->view;
    a.rays = static_cast<const float*>(dRays.p);
    a.count = n;
    a.idx = static_cast<int32_t*>(dIdx.p);
    a.out = static_cast<float*>(dOut.p);
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipError_t le = (hipError_t)(h->strict() ? kajo_kat_trace_strict_launch(&a, grid, h->hotBytes, h->stream)
                                             : kajo_kat_trace_fast_launch(&a, grid, h->hotBytes, h->stream));
    if (le != hipSuccess)
        return failHip(le, "kat trace launch");
    std::vector<float> out((size_t)n * 13);
    HIP_TRY(hipMemcpyAsync(objIndex, dIdx.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(out.data(), dOut.p, out.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++) {
        t[i] = out[13 * (size_t)i];
        std::memcpy(position + 3 * i, &out[13 * (size_t)i + 1], 12);
        std::memcpy(normal + 3 * i, &out[13 * (size_t)i + 4], 12);
        std::memcpy(tangent + 3 * i, &out[13 * (size_t)i + 7], 12);
        std::memcpy(binormal + 3 * i, &out[13 * (size_t)i + 10], 12);
    }
    return KAJO_OK;
}

int kajo_hip_kat_shade(kajo_hip_t h, int n, const float* origins, const float* dirs, const uint64_t* states, float* rgb,
                       uint64_t* finalStates)
{
    if (!h || n < 0 || !origins || !dirs || !states || !rgb || !finalStates)
        return fail(KAJO_E_INVALID, "null argument");
    if (h->hotBytes > 48 * 1024)
        return fail(KAJO_E_INVALID, "known-answer entry points are limited to scenes whose hot records fit 48 KiB of LDS");
    int rc = bind(h);
    if (rc || n == 0)
        return rc;
    std::vector<float> rays(6 * (size_t)n);
    for (int i = 0; i < n; i++) {
        std::memcpy(&rays[6 * i], origins + 3 * i, 12);
        std::memcpy(&rays[6 * i + 3], dirs + 3 * i, 12);
    }
    DeviceBuffer dRays, dStates, dRgb, dFinal;
    HIP_TRY(dRays.alloc(rays.size() * 4));
    HIP_TRY(dStates.alloc((size_t)n * 16));
    HIP_TRY(dRgb.alloc((size_t)n * 16));
    HIP_TRY(dFinal.alloc((size_t)n * 16));
    HIP_TRY(hipMemcpyAsync(dRays.p, rays.data(), rays.size() * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dStates.p, states, (size_t)n * 16, hipMemcpyHostToDevice, h->stream));
    RenderArgs a;
    std::memset(&a, 0, sizeof a);
    a.scene = h->view;
    a.W = a.H = 1;
    a.n = 1;
    a.S = 1.f;
    a.depthLimit = h->params.depthLimit;
    a.tileW = 64;
    a.tileH = 16;
    a.tilesX = a.tilesY = 1;
    a.tileCount = 1;
    a.katRays = static_cast<const float*>(dRays.p);
    a.katStates = static_cast<const uint64_t*>(dStates.p);
    a.katRgb = static_cast<float*>(dRgb.p);
    a.katFinal = static_cast<uint64_t*>(dFinal.p);
    a.katCount = n;
    a.stealWindow = 1;
    a.mailboxOffset = (uint32_t)((h->hotBytes + 15) & ~(size_t)15);
    h->fillWaveLds(a, a.mailboxOffset, false);
    const size_t ldsKat = a.mailboxOffset + 4 * (size_t)a.perWaveBytes;
    if (ldsKat > 64 * 1024)
        return fail(KAJO_E_INVALID, "known-answer entry points are limited to scenes whose hot records and wave areas fit 64 KiB of LDS");
    const unsigned grid = (unsigned)((n + 255) / 256);
    hipError_t le = (hipError_t)(h->numerics() == 1 ? kajo_kat_shade_strict_launch(&a, grid, ldsKat, h->stream)
                                 : h->numerics() == 2 ? kajo_kat_shade_exact_launch(&a, grid, ldsKat, h->stream)
                                                      : kajo_kat_shade_fast_launch(&a, grid, ldsKat, h->stream));
    if (le != hipSuccess)
        return failHip(le, "kat shade launch");
    std::vector<float> out4((size_t)n * 4);
    HIP_TRY(hipMemcpyAsync(out4.data(), dRgb.p, out4.size() * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipMemcpyAsync(finalStates, dFinal.p, (size_t)n * 16, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    for (int i = 0; i < n; i++)
        std::memcpy(rgb + 3 * i, &out4[4 * (size_t)i], 12);
    return KAJO_OK;
}

int kajo_hip_kat_strictmath(kajo_hip_t h, int fn, int n, const float* x, const float* y, float* out)
{
    if (!h || n < 0 || !x || !y || !out)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = bind(h);
    if (rc || n == 0)
        return rc;
    DeviceBuffer dx, dy, dout;
    HIP_TRY(dx.alloc((size_t)n * 4));
    HIP_TRY(dy.alloc((size_t)n * 4));
    HIP_TRY(dout.alloc((size_t)n * 4));
    HIP_TRY(hipMemcpyAsync(dx.p, x, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    HIP_TRY(hipMemcpyAsync(dy.p, y, (size_t)n * 4, hipMemcpyHostToDevice, h->stream));
    hipError_t le = (hipError_t)kajo_kat_math_launch(fn, n, dx.p, dy.p, dout.p, h->stream);
    if (le != hipSuccess)
        return failHip(le, "kat math launch");
    HIP_TRY(hipMemcpyAsync(out, dout.p, (size_t)n * 4, hipMemcpyDeviceToHost, h->stream));
    HIP_TRY(hipStreamSynchronize(h->stream));
    return KAJO_OK;
}

// Diagnostic builds only (-DKAJO_PROFILE): 28 raw block-profile words (16 block counts, 5 stamp sums, spare) behind the work counters.
extern "C" int kajo_hip_debug_profile(kajo_hip_t h, unsigned long long* out28)
{
    if (!h || !out28 || !h->counters)
        return fail(KAJO_E_INVALID, "no counters");
    int rc = kajo_hip_wait(h);
    if (rc)
        return rc;
    HIP_TRY(hipMemcpy(out28, h->counters + 4, 28 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return KAJO_OK;
}

int kajo_hip_counters(kajo_hip_t h, KajoCounters* out)
{
    if (!h || !out)
        return fail(KAJO_E_INVALID, "null argument");
    int rc = kajo_hip_wait(h);
    if (rc)
        return rc;
    std::memset(out, 0, sizeof *out);
    const int n = (int)std::sqrt((double)(unsigned)h->params.samplesPerPass);
    // pixels this handle owns
    unsigned long long pixels = 0;
    for (int t = h->params.tileIndex; t < h->nTiles; t += h->params.tileCount) {
        const int tx = t % h->map.tilesX, ty = t / h->map.tilesX;
        const int w = std::min(h->map.tileW, h->W - tx * h->map.tileW);
        const int hh = std::min(h->map.tileH, h->H - ty * h->map.tileH);
        pixels += (unsigned long long)w * hh;
    }
    out->passes = (uint64_t)h->passesDone;
    out->paths = pixels * (unsigned long long)(n * n) * (unsigned long long)h->passesDone;
    out->kernelMs = h->kernelMs;
    out->launches = h->launches;
    out->tailGroups = h->lastTailGroups;
    if (h->counters) {
        unsigned long long c[4];
        HIP_TRY(hipMemcpy(c, h->counters, sizeof c, hipMemcpyDeviceToHost));
        out->traversals = c[0];
        out->vertices = c[1];
        out->laneSlots = c[2];
        out->shadowQueries = c[3];
        out->primitiveTests = c[0] * (unsigned long long)(h->view.nPlanes + h->view.nSpheres);
    }
    return KAJO_OK;
}

} // extern "C"
