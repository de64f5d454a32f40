#include "HipScheduler.h"

#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>

#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>
#include <thread>
#include <vector>

#include "Image.h"
#include "Preview.h"
#include "kajo_hip.h"
#include "scene/Scene.h"

namespace hip
{

namespace
{

// C ABI error codes become exceptions on this side of the boundary
void check(int rc, const char* what)
{
    if (rc != KAJO_OK)
        throw std::runtime_error(std::string(what) + ": " + kajo_hip_last_error());
}

void checkHip(hipError_t e, const char* what)
{
    if (e != hipSuccess)
        throw std::runtime_error(std::string(what) + ": " + hipGetErrorString(e));
}

void checkNccl(ncclResult_t r, const char* what)
{
    if (r != ncclSuccess)
        throw std::runtime_error(std::string(what) + ": " + ncclGetErrorString(r));
}

// The first float of a 4x4 matrix or 4-vector member of scene::Scene, whatever its C++ type: this repo's stand-in
// (float[16] / four floats) or the reference's glm::mat4 / glm::vec4 -- both are 16 resp. 4 consecutive floats, column
// major. `make -C kajo_amd/host refcheck` compiles this file against the reference's own scene/Scene.h, Image.h and
// Scheduler.h where /root/reference exists.
template <class T, size_t N>
const float* floatsOf(const T& v)
{
    static_assert(sizeof(T) == N * sizeof(float), "matrix / vector member is not a packed float array");
    return reinterpret_cast<const float*>(&v);
}

void copyMaterial(const scene::Material& m, KajoMaterial& k)
{
    static_assert(sizeof(scene::Material) == sizeof(KajoMaterial), "scene::Material and KajoMaterial share one layout");
    std::memcpy(&k, &m, sizeof k);
}

} // namespace

struct Scheduler::Impl
{
    Image* image = nullptr;
    Preview* preview = nullptr;
    Options opt;
    std::vector<kajo_hip_t> handles;
    std::vector<int> devices;
    std::vector<hipStream_t> streams; // one per owner, shared with its handle
    std::vector<ncclComm_t> comms;
    void* gathered = nullptr;         // on device 0: gpus consecutive tile buffers
    void* argbDevice = nullptr;       // on device 0: the resolved image of a gathered frame, before it goes to Image::pixels
    size_t tileBytes = 0;
    Statistics stats;

    ~Impl()
    {
        for (kajo_hip_t h : handles)
            kajo_hip_destroy(h);
        for (ncclComm_t c : comms)
            ncclCommDestroy(c);
        for (size_t i = 0; i < streams.size(); i++) {
            (void)hipSetDevice(devices[i]);
            (void)hipStreamDestroy(streams[i]);
        }
        if (gathered) {
            (void)hipSetDevice(devices.empty() ? 0 : devices[0]);
            (void)hipFree(gathered);
        }
        if (argbDevice)
            (void)hipFree(argbDevice);
    }

    void create(const scene::Scene& s)
    {
        // scene::Scene -> flat POD (valid only during this call, like the reference's const&)
        std::vector<KajoSphere> spheres(s.spheres.size());
        std::vector<KajoPlane> planes(s.planes.size());
        for (size_t i = 0; i < s.spheres.size(); i++) {
            std::memcpy(spheres[i].transform, floatsOf<decltype(s.spheres[i].transform), 16>(s.spheres[i].transform), 64);
            copyMaterial(s.spheres[i].material, spheres[i].material);
            spheres[i].radius = s.spheres[i].radius;
        }
        for (size_t i = 0; i < s.planes.size(); i++) {
            std::memcpy(planes[i].transform, floatsOf<decltype(s.planes[i].transform), 16>(s.planes[i].transform), 64);
            copyMaterial(s.planes[i].material, planes[i].material);
        }
        KajoScene pod;
        std::memcpy(pod.backgroundColor, floatsOf<decltype(s.backgroundColor), 4>(s.backgroundColor), 16);
        std::memcpy(pod.camera.transform, floatsOf<decltype(s.camera.transform), 16>(s.camera.transform), 64);
        std::memcpy(pod.camera.projection, floatsOf<decltype(s.camera.projection), 16>(s.camera.projection), 64);
        pod.nSpheres = (int32_t)spheres.size();
        pod.nPlanes = (int32_t)planes.size();
        pod.spheres = spheres.data();
        pod.planes = planes.data();

        if (opt.gpus == 0) {
            // The reference's driver has no flag for it (renderer/Main.cpp:104-120): the backend takes the node as it finds it,
            // as cpu::Scheduler takes every core (renderer/cpu/Scheduler.cpp:17-24). KAJO_HIP_GPUS caps it.
            int visible = 0;
            checkHip(hipGetDeviceCount(&visible), "hipGetDeviceCount");
            opt.gpus = visible;
            if (const char* e = std::getenv("KAJO_HIP_GPUS")) {
                const int want = std::atoi(e);
                if (want < 1 || want > visible)
                    throw std::runtime_error("hip::Scheduler: KAJO_HIP_GPUS must be between 1 and the number of visible GPUs");
                opt.gpus = want;
            }
        }
        if (opt.gpus < 1)
            throw std::runtime_error("hip::Scheduler: no GPU visible (this backend has no CPU path)");
        if (opt.sameDevice && opt.gather != Options::Copy)
            throw std::runtime_error("hip::Scheduler: sameDevice needs gather = Copy (RCCL wants one rank per device)");
        Options::Numerics numerics = opt.strict ? Options::Strict : opt.numerics;
        if (opt.numericsFromEnvironment && std::getenv("KAJO_HIP_FORCE_GATHER"))
            opt.forceGather = true; // (testing: the three-argument form's N > 1 code -- communicator, gather, resolve from the gathered buffers -- on a one-GPU box)
        if (opt.numericsFromEnvironment) {
            if (const char* e = std::getenv("KAJO_HIP_NUMERICS")) {
                const std::string v(e);
                if (v == "exact") numerics = Options::Exact;
                else if (v == "fast") numerics = Options::Fast;
                else if (v == "strict") numerics = Options::Strict;
                else throw std::runtime_error("hip::Scheduler: KAJO_HIP_NUMERICS must be exact, fast or strict");
            }
        }
        for (int g = 0; g < opt.gpus; g++) {
            KajoParams p;
            kajo_hip_default_params(&p);
            p.samplesPerPass = opt.samplesPerPass;
            p.depthLimit = opt.depthLimit;
            p.seed = opt.seed;
            p.flags = (numerics == Options::Strict ? KAJO_FLAG_STRICT : numerics == Options::Exact ? KAJO_FLAG_EXACT : 0u) | (opt.counters ? KAJO_FLAG_COUNTERS : 0u);
            p.device = opt.sameDevice ? 0 : g;
            p.tileIndex = g;
            p.tileCount = opt.gpus;
            p.passesPerLaunch = opt.passesPerUpdate > 0 ? opt.passesPerUpdate : 16;
            kajo_hip_t h = nullptr;
            check(kajo_hip_create(&pod, image->width, image->height, &p, &h), "kajo_hip_create");
            handles.push_back(h);
            devices.push_back(p.device);
        }
        if (opt.gpus > 1 || opt.forceGather) {
            void* ptr = nullptr;
            check(kajo_hip_tile_buffer(handles[0], &ptr, &tileBytes), "kajo_hip_tile_buffer");
            checkHip(hipSetDevice(devices[0]), "hipSetDevice");
            checkHip(hipMalloc(&gathered, tileBytes * opt.gpus), "hipMalloc(gather buffer)");
            checkHip(hipMalloc(&argbDevice, (size_t)image->width * image->height * 4), "hipMalloc(image)");
            // one stream per owner carries both its render kernels and its share of the gather
            for (int g = 0; g < opt.gpus; g++) {
                checkHip(hipSetDevice(devices[g]), "hipSetDevice");
                hipStream_t st;
                checkHip(hipStreamCreateWithFlags(&st, hipStreamNonBlocking), "hipStreamCreate");
                streams.push_back(st);
                check(kajo_hip_set_stream(handles[g], st), "kajo_hip_set_stream");
            }
            if (opt.gather == Options::Rccl) {
                comms.resize(opt.gpus);
                checkNccl(ncclCommInitAll(comms.data(), opt.gpus, devices.data()), "ncclCommInitAll");
            }
        }
    }

    // One exchange per displayed frame: every owner's tile buffer -> GPU 0; then the image straight
    // from the gathered buffers into Image::pixels (the float frame is composed only when readRadiance() asks for it).
    void gatherAndResolve()
    {
        if (opt.gpus == 1 && !opt.forceGather) {
            // single owner: the library resolves from its own tile buffer
            check(kajo_hip_resolve_argb8(handles[0], image->pixels.get()), "kajo_hip_resolve_argb8");
            return;
        }
        const size_t count = tileBytes / sizeof(float);
        if (opt.gather == Options::Rccl) {
            checkNccl(ncclGroupStart(), "ncclGroupStart");
            for (int g = 0; g < opt.gpus; g++) {
                void* src = nullptr;
                size_t bytes = 0;
                check(kajo_hip_tile_buffer(handles[g], &src, &bytes), "kajo_hip_tile_buffer");
                checkNccl(ncclSend(src, count, ncclFloat, 0, comms[g], streams[g]), "ncclSend");
                checkNccl(ncclRecv(static_cast<char*>(gathered) + (size_t)g * tileBytes, count, ncclFloat, g, comms[0], streams[0]),
                          "ncclRecv");
            }
            checkNccl(ncclGroupEnd(), "ncclGroupEnd");
        } else {
            for (int g = 0; g < opt.gpus; g++) {
                void* src = nullptr;
                size_t bytes = 0;
                check(kajo_hip_tile_buffer(handles[g], &src, &bytes), "kajo_hip_tile_buffer");
                check(kajo_hip_wait(handles[g]), "kajo_hip_wait");
                checkHip(hipSetDevice(devices[0]), "hipSetDevice");
                checkHip(hipMemcpyAsync(static_cast<char*>(gathered) + (size_t)g * tileBytes, src, bytes, hipMemcpyDefault, streams[0]),
                         "hipMemcpyAsync(gather)");
            }
        }
        composed = false;
        check(kajo_hip_resolve_gathered_argb8_device(handles[0], gathered, argbDevice), "kajo_hip_resolve_gathered_argb8_device");
        checkHip(hipSetDevice(devices[0]), "hipSetDevice");
        checkHip(hipMemcpyAsync(image->pixels.get(), argbDevice, (size_t)image->width * image->height * 4, hipMemcpyDeviceToHost, streams[0]),
                 "hipMemcpyAsync(image)");
        checkHip(hipStreamSynchronize(streams[0]), "hipStreamSynchronize");
    }
    bool composed = false;
};

namespace
{
Options optionsOfTheThreeArgumentForm()
{
    Options o; // every visible GPU (gpus = 0), reference constants, EXACT numerics unless KAJO_HIP_NUMERICS says otherwise
    o.numericsFromEnvironment = true;
    return o;
}
} // namespace

Scheduler::Scheduler(const scene::Scene& scene, Image* image, Preview* preview): Scheduler(scene, image, preview, optionsOfTheThreeArgumentForm())
{
}

Scheduler::Scheduler(const scene::Scene& scene, Image* image, Preview* preview, const Options& options): m_impl(new Impl)
{
    m_impl->image = image;
    m_impl->preview = preview;
    m_impl->opt = options;
    m_impl->create(scene);
}

Scheduler::~Scheduler() {}

const Statistics& Scheduler::statistics() const
{
    return m_impl->stats;
}

void Scheduler::readRadiance(float* dst)
{
    Impl& d = *m_impl;
    if (d.gathered && !d.composed) { // several owners (or the forced gather): the whole float frame from the last gather
        check(kajo_hip_compose(d.handles[0], d.gathered), "kajo_hip_compose");
        d.composed = true;
    }
    check(kajo_hip_read_radiance(d.handles[0], dst), "kajo_hip_read_radiance");
}

void Scheduler::run()
{
    Impl& d = *m_impl;
    const Options& o = d.opt;
    const int budget = o.passes > 0 ? o.passes : (d.preview ? 0 : 16);
    const std::thread::id self = std::this_thread::get_id();
    const auto t0 = std::chrono::steady_clock::now();
    int done = 0;
    std::vector<double> batchMs;
    std::vector<int> batchPasses;
    // Passes between two refreshes. Fusing passes into one launch evens out the lanes' trip counts (38 G paths/s at
    // 16 per launch against 30 at one, profiles/HISTORY.md section 6), so headless runs take all that is left and a live preview
    // gets as many as fit a 30 Hz refresh, from the measured time per pass.
    // A launch cannot be interrupted (the reference's workers look at their stop flag once per row, cpu/Renderer.cpp:77-78): the
    // bound on how long run() can overshoot a closed window -- or a caller waits for the first image -- is the length of one
    // launch. With a preview: a 30 Hz refresh. Headless: half a second (the 1000-sphere scene at 4K takes 37 ms per pass: 13
    // passes per launch instead of 16, where round 4 launched 16 regardless -- 0.6 s -- and 32 in the tools: 1.2 s).
    double msPerPass = 0.0;
    int samples = 0; // batches measured so far
    auto autoBatch = [&]() {
        if (o.passesPerUpdate > 0)
            return o.passesPerUpdate;
        if (msPerPass <= 0.0)
            return d.preview ? 1 : 2; // nothing measured yet
        const int fit = (int)((d.preview ? 33.0 : 500.0) / msPerPass);
        if (fit >= 8) {
            // launches of 8 or 16 passes that begin with a group of four (include/kajo_hip.h kajo_hip_render): the FAST / EXACT kernels then
            // render the cheapest blocks of a large frame as one workgroup per group (KajoCounters.tailGroups: +2 % at 1920x1080); a shorter
            // batch first if the passes done so far end inside a group. Scheduling only: the frame does not depend on the batches.
            const int over = done % 4;
            return over ? 4 - over : (fit >= 16 ? 16 : 8);
        }
        return fit < 1 ? 1 : fit;
    };

    // The SDL calls of the preview stay on this (the main) thread, as the reference requires
    // (cpu/Scheduler.cpp:64-81 marshals worker progress through a queue for the same reason).
    while ((!d.preview || d.preview->processEvents()) && (budget == 0 || done < budget)) {
        const int batch = autoBatch();
        const int now = budget == 0 ? batch : (budget - done < batch ? budget - done : batch);
        const auto tb = std::chrono::steady_clock::now();
        for (kajo_hip_t h : d.handles)
            check(kajo_hip_render(h, now), "kajo_hip_render"); // asynchronous, one stream per GPU
        for (kajo_hip_t h : d.handles)
            check(kajo_hip_wait(h), "kajo_hip_wait");
        done += now;
        d.gatherAndResolve();
        const double batchWall = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - tb).count();
        batchMs.push_back(batchWall);
        batchPasses.push_back(now);
        // (the first batch pays for the cold start -- module load, the blocks' cost measurement and its order: it sizes the second batch and
        // is then forgotten)
        const double ms = batchWall / now;
        msPerPass = samples < 2 ? ms : 0.75 * msPerPass + 0.25 * ms;
        samples++;
        if (d.preview)
            for (int p = done - now + 1; p <= done; p++)
                d.preview->update(self, p, o.samplesPerPass, 0, 0, d.image->width, d.image->height);
    }

    d.stats = Statistics();
    d.stats.passes = done;
    d.stats.gpus = o.gpus;
    d.stats.batchMs = batchMs;
    d.stats.batchPasses = batchPasses;
    d.stats.wallSeconds = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    for (kajo_hip_t h : d.handles) {
        KajoCounters c;
        check(kajo_hip_counters(h, &c), "kajo_hip_counters");
        d.stats.paths += c.paths;
        d.stats.traversals += c.traversals;
        d.stats.vertices += c.vertices;
        d.stats.laneSlots += c.laneSlots;
        if (c.kernelMs > d.stats.kernelMs)
            d.stats.kernelMs = c.kernelMs;
    }
}

} // namespace hip
