// HipScheduler.h -- the "-r hip" backend behind Kajo's Scheduler plugin interface.
//
// Drop-in counterpart of cpu::Scheduler (renderer/cpu/Scheduler.h:22-34, .cpp:53-85): same
// constructor shape (const scene::Scene&, Image*, Preview*), same single blocking run(). Where the
// CPU backend cuts the image into one row slice per core and renders them with std::async
// (cpu/Scheduler.cpp:32-42), this one deals fixed-size tiles round-robin to the node's GPUs, runs
// one libkajo_hip handle per GPU (include/kajo_hip.h), gathers the tile buffers to GPU 0 once per
// displayed frame (RCCL over xGMI) and resolves into Image::pixels. The scene is only read inside
// the constructor, as in the reference (cpu/Scene.cpp:29-38 copies it).
#ifndef KAJO_HIP_SCHEDULER_H
#define KAJO_HIP_SCHEDULER_H

#include <cstdint>
#include <memory>
#include <vector>

#include "Scheduler.h"

class Image;
class Preview;

namespace scene
{
class Scene;
}

namespace hip
{

struct Options
{
    int samplesPerPass = 32;      // m_samples, renderer/cpu/Renderer.cpp:21
    int depthLimit = 8;           // g_depthLimit, renderer/cpu/Shader.cpp:24
    uint64_t seed = 0715517;      // renderer/cpu/Random.h:43
    int passes = 0;               // stop after this many passes; 0 = until the preview closes
                                  // (16 when there is no preview: the reference never stops by itself)
    int passesPerUpdate = 0;      // passes rendered between two image/preview refreshes; 0 = automatic: everything that is
                                  // left (up to 16) without a preview, as many as fit a 30 Hz refresh with one
    int gpus = 0;                 // devices 0 .. gpus-1; 0 = every GPU the HIP runtime shows (hipGetDeviceCount), or the
                                  // number in the environment variable KAJO_HIP_GPUS when that is set -- what the
                                  // three-argument constructor (the form renderer/Main.cpp:135-142 calls) uses, so that
                                  // `renderer -r hip scene.json` tiles the frame over the whole node with no flag the
                                  // reference does not have
    // Numerics build of the kernels (include/kajo_hip.h). Exact (the default, round 5): every path takes the decisions of
    // renderer/cpu -- same hits, same random draws -- and only the products that scale its radiance are formed in the GPU's fast
    // forms: the frame is the reference's to ~1e-6 (BASELINE.json asks for per-pixel RMSE < 1e-4). Fast: hardware
    // transcendentals and contraction everywhere, 1.6 x the rate, a few paths per million decide differently (RMSE ~6e-4 on
    // spheres.json at 512 spp). Strict: the CPU oracle bit for bit. The environment variable KAJO_HIP_NUMERICS (exact | fast |
    // strict) overrides it for the three-argument constructor, which has no options.
    enum Numerics { Exact, Fast, Strict } numerics = Exact;
    bool strict = false;          // (kept from rounds 1-4) true = Numerics::Strict
    bool numericsFromEnvironment = false; // set by the three-argument constructor: KAJO_HIP_NUMERICS may choose the build
    bool counters = false;
    enum Gather { Rccl, Copy } gather = Rccl; // Copy: hipMemcpyAsync instead of RCCL (also lets
                                              // several tile owners share ONE device, for tests)
    bool sameDevice = false;      // all tile owners on device 0 (needs gather = Copy)
    bool forceGather = false;     // run the gather + compose step with ONE owner too (a one-rank communicator whose
                                  // rank sends its tile buffer to itself): the multi-GPU call sequence on a one-GPU box
};

struct Statistics
{
    int passes = 0;
    unsigned long long paths = 0, traversals = 0, vertices = 0, laneSlots = 0;
    double kernelMs = 0;   // max over GPUs of the summed render-kernel time
    double wallSeconds = 0;
    int gpus = 0;          // tile owners the frame was dealt to
    std::vector<double> batchMs; // wall time of every refresh: render launch .. Image::pixels filled (host), in run() order
    std::vector<int> batchPasses;
};

class Scheduler : public ::Scheduler
{
public:
    Scheduler(const scene::Scene&, Image*, Preview*);
    Scheduler(const scene::Scene&, Image*, Preview*, const Options&);
    // not marked `override`: the reference's ::Scheduler (renderer/Scheduler.h:12-16) has no virtual destructor -- its
    // Main.cpp deletes backends through the base pointer at exit, so there this destructor would not run (the process
    // ends right after; GPU memory goes with it). With this repo's stand-in base it is virtual.
    ~Scheduler();

    void run() override;

    const Statistics& statistics() const;
    // whole-frame float accumulation (W*H*4, sum over passes of radiance / S) after run()
    void readRadiance(float* dst);

private:
    struct Impl;
    std::unique_ptr<Impl> m_impl;
};

} // namespace hip

#endif
