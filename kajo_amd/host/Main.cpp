// Main.cpp -- headless driver with the reference's command line (renderer/Main.cpp:97-146):
//   kajo_render [-w SIZE] [-h SIZE] [-r hip] [options] SCENE.json     (no SCENE: the built-in test scene)
// plus what a window-less run needs: a pass budget, an output name and the backend's options.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <memory>
#include <string>
#include <vector>

#include "HipScheduler.h"
#include "Image.h"
#include "Preview.h"
#include "scene/Scene.h"

int main(int argc, char** argv)
{
    std::vector<std::string> args(argv, argv + argc);
    std::string rendererName = "hip", out = "out.png", rawOut, scenePath;
    int width = 640, height = 480;
    hip::Options opt;
    opt.passes = 16;
    opt.gpus = 1;
    std::string podPath;
    bool verbose = false, json = false, threeArg = false;
    int closeAtEvent = 0;
    bool noPreview = false;
    for (size_t i = 1; i < args.size(); i++) {
        bool more = i + 1 < args.size();
        const std::string& a = args[i];
        if (a == "--help") {
            std::printf("Usage: %s OPTIONS [SCENE]\n\n"
                        "    -w SIZE         image width (640)\n"
                        "    -h SIZE         image height (480)\n"
                        "    -r NAME         renderer (hip)\n"
                        "    --spp N         nominal samples per pixel per pass (32)\n"
                        "    --passes N      passes to render (16)\n"
                        "    --bounces N     depth limit (8)\n"
                        "    --seed N        stream seed (236367)\n"
                        "    --gpus N        GPUs to tile the frame over (1; 0 = every visible GPU, or KAJO_HIP_GPUS)\n"
                        "    --three-arg     construct the backend exactly as a Kajo checkout does, hip::Scheduler(scene, image, preview):\n"
                        "                    reference constants, every visible GPU (or KAJO_HIP_GPUS), until the preview closes (--passes)\n"
                        "    --scene-pod F   read the scene as a flat binary image of scene::Scene (int32 nSpheres, nPlanes; background 4 f32;\n"
                        "                    view 16; projection 16; spheres 39 f32 each; planes 38 f32 each) instead of a JSON file\n"
                        "    --batch N       passes per image refresh (0 = automatic: 16 headless, a 30 Hz refresh with a preview)\n"
                        "    --exact         numerics: the reference's decisions on every path, fast arithmetic for radiance only (default)\n"
                        "    --fast          numerics: hardware transcendentals and contraction everywhere (1.6 x the rate, RMSE ~6e-4)\n"
                        "    --strict        numerics: the CPU oracle bit for bit\n"
                        "    --gather MODE   rccl | copy (multi-GPU gather transport)\n"
                        "    --same-device   put every tile owner on GPU 0 (testing; implies --gather copy)\n"
                        "    --force-gather  run the gather + compose step with one GPU too (testing: the RCCL call sequence at N = 1)\n"
                        "    --close-at-event K  the headless preview's window closes at the K-th processEvents() call (testing: Esc mid-run)\n"
                        "    --no-preview    hand the backend a null Preview* (what renderer/Main.cpp:132 gets without a display): run() renders\n"
                        "                    --passes passes (16 when 0) in launches planned for half a second at most\n"
                        "    -o FILE         PNG output (out.png)\n"
                        "    --raw FILE      also dump the float4 accumulation (W*H*4 floats)\n"
                        "    --json          print run statistics as one JSON line\n"
                        "    -v              progress on stderr\n",
                        args[0].c_str());
            return 1;
        } else if (a == "-w" && more) width = std::atoi(args[++i].c_str());
        else if (a == "-h" && more) height = std::atoi(args[++i].c_str());
        else if (a == "-r" && more) rendererName = args[++i];
        else if (a == "--spp" && more) opt.samplesPerPass = std::atoi(args[++i].c_str());
        else if (a == "--passes" && more) opt.passes = std::atoi(args[++i].c_str());
        else if (a == "--bounces" && more) opt.depthLimit = std::atoi(args[++i].c_str());
        else if (a == "--seed" && more) opt.seed = std::strtoull(args[++i].c_str(), nullptr, 0);
        else if (a == "--gpus" && more) opt.gpus = std::atoi(args[++i].c_str());
        else if (a == "--batch" && more) opt.passesPerUpdate = std::atoi(args[++i].c_str());
        else if (a == "--strict") opt.numerics = hip::Options::Strict;
        else if (a == "--exact") opt.numerics = hip::Options::Exact;
        else if (a == "--fast") opt.numerics = hip::Options::Fast;
        else if (a == "--gather" && more) opt.gather = args[++i] == "copy" ? hip::Options::Copy : hip::Options::Rccl;
        else if (a == "--same-device") { opt.sameDevice = true; opt.gather = hip::Options::Copy; }
        else if (a == "--force-gather") opt.forceGather = true;
        else if (a == "--three-arg") threeArg = true;
        else if (a == "--close-at-event" && more) closeAtEvent = std::atoi(args[++i].c_str());
        else if (a == "--no-preview") noPreview = true;
        else if (a == "--scene-pod" && more) podPath = args[++i];
        else if (a == "-o" && more) out = args[++i];
        else if (a == "--raw" && more) rawOut = args[++i];
        else if (a == "--json") json = true;
        else if (a == "-v") verbose = true;
        else if (!a.empty() && a[0] != '-') scenePath = a;
    }
    if (width <= 0 || height <= 0) {
        std::cerr << "Bad image size" << std::endl;
        return 1;
    }

    scene::Scene scene;
    if (!podPath.empty()) {
        // a parsed scene as the tests' fixtures hold it (tests/golden/scenes.npz): the records have scene::Scene's own layout
        std::ifstream f(podPath, std::ios::binary);
        int32_t counts[2] = {0, 0};
        f.read(reinterpret_cast<char*>(counts), sizeof counts);
        if (!f || counts[0] < 0 || counts[1] < 0 || counts[0] > (1 << 20) || counts[1] > (1 << 20)) {
            std::cerr << "Failed to read scene image " << podPath << std::endl;
            return 1;
        }
        static_assert(sizeof(scene::Sphere) == 39 * sizeof(float) && sizeof(scene::Plane) == 38 * sizeof(float), "records are packed floats");
        f.read(reinterpret_cast<char*>(&scene.backgroundColor), 16);
        f.read(reinterpret_cast<char*>(&scene.camera.transform), 64);
        f.read(reinterpret_cast<char*>(&scene.camera.projection), 64);
        scene.spheres.resize(counts[0]);
        scene.planes.resize(counts[1]);
        f.read(reinterpret_cast<char*>(scene.spheres.data()), (std::streamsize)(sizeof(scene::Sphere) * scene.spheres.size()));
        f.read(reinterpret_cast<char*>(scene.planes.data()), (std::streamsize)(sizeof(scene::Plane) * scene.planes.size()));
        if (!f) {
            std::cerr << "Failed to read scene image " << podPath << std::endl;
            return 1;
        }
    } else if (scenePath.empty())
        scene::buildTestScene(scene);
    else if (!scene::Parser::load(scene, scenePath, static_cast<float>(width) / height)) {
        std::cerr << "Failed to parse scene from " << scenePath << std::endl;
        return 1;
    }

    std::unique_ptr<Image> image(new Image(width, height));
    std::unique_ptr<Preview> preview(Preview::create(image.get(), true)); // as renderer/Main.cpp:132
    preview->setPassBudget(opt.passes, verbose);
    preview->closeAtEvent(closeAtEvent);
    std::unique_ptr<Scheduler> scheduler;
    hip::Scheduler* hipScheduler = nullptr;
    opt.counters = json;
    try {
        if (rendererName == "hip") {
            // (--three-arg: the statement integration/apply_to_kajo.sh adds to renderer/Main.cpp:135-142, word for word)
            Preview* pv = noPreview ? nullptr : preview.get();
            hipScheduler = threeArg ? new hip::Scheduler(scene, image.get(), pv) : new hip::Scheduler(scene, image.get(), pv, opt);
            scheduler.reset(hipScheduler);
        } else {
            std::cerr << "Unknown renderer: " << rendererName << std::endl;
            return 1;
        }
        scheduler->run();
        if (!rawOut.empty()) {
            std::vector<float> acc((size_t)width * height * 4);
            hipScheduler->readRadiance(acc.data());
            std::ofstream f(rawOut, std::ios::binary);
            f.write(reinterpret_cast<const char*>(acc.data()), (std::streamsize)(acc.size() * sizeof(float)));
        }
    } catch (const std::exception& e) {
        std::cerr << "kajo_render: " << e.what() << std::endl;
        return 2;
    }
    if (!out.empty() && !image->save(out))
        return 3;
    if (json) {
        const hip::Statistics& s = hipScheduler->statistics();
        std::printf("{\"width\": %d, \"height\": %d, \"passes\": %d, \"gpus\": %d, \"paths\": %llu, \"traversals\": %llu, "
                    "\"vertices\": %llu, \"wall_s\": %.6f, \"kernel_ms\": %.3f, \"msamples_per_s\": %.2f, \"batch_ms\": [",
                    width, height, s.passes, s.gpus, s.paths, s.traversals, s.vertices, s.wallSeconds, s.kernelMs,
                    s.paths / s.wallSeconds / 1e6);
        for (size_t i = 0; i < s.batchMs.size(); i++)
            std::printf("%s%.4f", i ? ", " : "", s.batchMs[i]);
        std::printf("], \"batch_passes\": [");
        for (size_t i = 0; i < s.batchPasses.size(); i++)
            std::printf("%s%d", i ? ", " : "", s.batchPasses[i]);
        // what the preview saw (renderer/Preview.cpp:79-98,216-234): every update() in call order, whether each came on the thread that
        // owns the window, and how often it was asked for events
        std::printf("], \"preview_updates\": [");
        bool onOwner = true;
        for (size_t i = 0; i < preview->updates().size(); i++) {
            std::printf("%s%d", i ? ", " : "", preview->updates()[i].pass);
            onOwner = onOwner && preview->updates()[i].onCreatingThread;
        }
        std::printf("], \"preview_updates_on_owning_thread\": %s, \"preview_event_calls\": %d}\n", onOwner ? "true" : "false", preview->eventCalls());
    }
    return 0;
}
