// launch.inc.hip -- host-side launcher of the kernel defined by the including file.
#include <mutex>
#define KAJO_CAT2(a, b) a##b
#define KAJO_CAT(a, b) KAJO_CAT2(a, b)

// coldInLds: 1 = whole scene staged in LDS (KAJO_KERNEL_NAME; FAST: its one-light instance for scenes of one light, 2 = never that one),
// 0 = cold records stay global (the scene's shadow.enabled picks the kernel that answers shadow queries from the lights' visibility lists)
extern "C" int KAJO_CAT(KAJO_KERNEL_NAME, _launch)(const RenderArgs* args, int coldInLds, unsigned grid, unsigned block,
                                                size_t ldsBytes, void* stream)
{
#ifdef KAJO_KERNEL_NAME_LIGHTS
    if (coldInLds && (args->scene.nLights != 1 || coldInLds == 2))
        hipLaunchKernelGGL(KAJO_KERNEL_NAME_LIGHTS, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
    else
#endif
    if (coldInLds)
        hipLaunchKernelGGL(KAJO_KERNEL_NAME, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
#ifdef KAJO_KERNEL_NAME_BIG_LG
    else if (args->scene.grid.enabled && args->scene.grid.inLds && args->scene.shadow.enabled)
        hipLaunchKernelGGL(KAJO_KERNEL_NAME_BIGLIST_LG, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
    else if (args->scene.grid.enabled && args->scene.grid.inLds)
        hipLaunchKernelGGL(KAJO_KERNEL_NAME_BIG_LG, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
#endif
    else if (args->scene.shadow.enabled)
        hipLaunchKernelGGL(KAJO_KERNEL_NAME_BIGLIST, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
    else
        hipLaunchKernelGGL(KAJO_KERNEL_NAME_BIG, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
    return (int)hipGetLastError();
}

// small frames: `block` = 64 * (waves sharing one pixel block), grid = pixel blocks
extern "C" int KAJO_CAT(KAJO_KERNEL_NAME, _split_launch)(const RenderArgs* args, unsigned grid, unsigned block, size_t ldsBytes, void* stream)
{
    hipLaunchKernelGGL(KAJO_KERNEL_NAME_SPLIT, dim3(grid), dim3(block), ldsBytes, static_cast<hipStream_t>(stream), *args);
    return (int)hipGetLastError();
}

// Dynamic LDS above the 64 KiB default needs an explicit opt-in on the function. The attribute is state of the
// FUNCTION ON A DEVICE, shared by every handle of the process on that device: it is only ever raised (a later, smaller
// scene must not lower the limit under an earlier handle's launches), and remembered per device -- a process that drives
// several GPUs (hip::Scheduler) sets it on each. Called with the handle's device current (kajo_hip_create).
extern "C" int KAJO_CAT(KAJO_KERNEL_NAME, _set_lds)(int coldInLds, size_t ldsBytes)
{
    static size_t highWaterOfDevice[64][2] = {};
    static std::mutex guard; // handles are created from any thread; the table and the attribute calls below are one critical section
    int device = 0;
    hipError_t e = hipGetDevice(&device);
    if (e != hipSuccess)
        return (int)e;
    if (device < 0 || device >= 64)
        return (int)hipErrorInvalidDevice; // (the table has 64 rows; ordinals are never aliased)
    std::lock_guard<std::mutex> lock(guard);
    size_t* highWater = highWaterOfDevice[device];
    const int k = coldInLds ? 1 : 0;
    if (ldsBytes <= highWater[k])
        return (int)hipSuccess;
    if (coldInLds) {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
#ifdef KAJO_KERNEL_NAME_LIGHTS
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME_LIGHTS), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
#endif
    } else {
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME_BIG), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME_BIGLIST), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
#ifdef KAJO_KERNEL_NAME_BIG_LG
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME_BIG_LG), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
        if (e == hipSuccess)
            e = hipFuncSetAttribute(reinterpret_cast<const void*>(KAJO_KERNEL_NAME_BIGLIST_LG), hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsBytes);
#endif
    }
    if (e == hipSuccess)
        highWater[k] = ldsBytes;
    return (int)e;
}

#ifdef KAJO_RESOLVE_NAME
extern "C" int KAJO_CAT(KAJO_RESOLVE_NAME, _launch)(const void* frame, int count, float passes, void* dst, void* stream)
{
    const unsigned block = 256, grid = (unsigned)((count + 255) / 256);
    hipLaunchKernelGGL(KAJO_RESOLVE_NAME, dim3(grid), dim3(block), 0, static_cast<hipStream_t>(stream),
                       static_cast<const float4*>(frame), count, passes, static_cast<uint32_t*>(dst));
    return (int)hipGetLastError();
}

extern "C" int KAJO_CAT(KAJO_RESOLVE_TILES_NAME, _launch)(const void* gathered, const TileMap* map, float passes, void* dst, void* stream)
{
    dim3 grid((map->W + 63) / 64, (map->H + 3) / 4), block(256);
    hipLaunchKernelGGL(KAJO_RESOLVE_TILES_NAME, grid, block, 0, static_cast<hipStream_t>(stream), static_cast<const float4*>(gathered), *map, passes,
                       static_cast<uint32_t*>(dst));
    return (int)hipGetLastError();
}

#endif

extern "C" int KAJO_CAT(KAJO_KAT_SHADE_NAME, _launch)(const RenderArgs* args, unsigned grid, size_t ldsBytes, void* stream)
{
    hipLaunchKernelGGL(KAJO_KAT_SHADE_NAME, dim3(grid), dim3(256), ldsBytes, static_cast<hipStream_t>(stream), *args);
    return (int)hipGetLastError();
}

#ifdef KAJO_KAT_TRACE_NAME
extern "C" int KAJO_CAT(KAJO_KAT_TRACE_NAME, _launch)(const KatTraceArgs* args, unsigned grid, size_t ldsBytes, void* stream)
{
    hipLaunchKernelGGL(KAJO_KAT_TRACE_NAME, dim3(grid), dim3(256), ldsBytes, static_cast<hipStream_t>(stream), *args);
    return (int)hipGetLastError();
}
#endif
