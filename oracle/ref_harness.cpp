// ref_harness.cpp -- TEST INFRASTRUCTURE, not product code.
//
// A thin extern "C" driver around the *compiled reference* (skyostil/kajo, renderer/cpu).
// All arithmetic stays inside the reference's own objects (cpu::Raytracer::trace,
// cpu::Shader::shade, cpu::Random::generate, the BSDF / SphericalLight classes and
// scene::Parser); this file only feeds them inputs and copies their outputs into flat
// arrays so that tests/golden/make_golden.py can capture known-answer vectors, and so
// that bench.py can time the reference's own hot loop as the CPU baseline.
//
// Built by oracle/Makefile into oracle/_ref/libkajo_ref*.so from the reference sources
// where they lie under /root/reference. Never linked or loaded by the product. The built
// libraries are kept out of git history (.gitignore) but DO travel to the GPU box with the
// snapshot, like this repo's own built .so files, so that tests and bench.py there can check
// against, and time, the reference itself (cpu_baseline.kind "reference"); the reference's
// SOURCES never leave /root/reference. (profiles/HISTORY.md section 2, "What travels to the GPU box".)
//
// Per-sample stream protocol: the 128-bit cpu::Random state is
// overwritten (16 bytes at offset 0 of the object; Random derives from the empty
// NonCopyable, renderer/cpu/Random.h:40,63) immediately before the jitter draw of every
// camera path (the draw at renderer/cpu/Renderer.cpp:55).

#include "scene/Scene.h"
#include "scene/Parser.h"
#include "renderer/Image.h"
#include "renderer/cpu/BSDF.h"
#include "renderer/cpu/Light.h"
#include "renderer/cpu/Random.h"
#include "renderer/cpu/Ray.h"
#include "renderer/cpu/Raytracer.h"
#include "renderer/cpu/Renderer.h"
#include "renderer/cpu/Scene.h"
#include "renderer/cpu/Shader.h"
#include "renderer/cpu/SurfacePoint.h"

#include <glm/gtc/matrix_transform.hpp>

#include <atomic>
#include <chrono>
#include <cmath>
#include <cstring>
#include <future>
#include <memory>
#include <vector>

#include "kajo_scene.h"
#include "kajo_stream.h"

// The depth limit is a file-static const in renderer/cpu/Shader.cpp:24. The Makefile
// feeds that one translation unit through sed (stdin, no copy on disk) so that the name
// resolves to this variable instead; default = the reference's 8.
extern "C" int kajo_ref_depth_limit;
int kajo_ref_depth_limit = 8;

namespace
{

struct Ref
{
    scene::Scene ss;
    std::unique_ptr<cpu::Scene> cs;
    std::unique_ptr<cpu::Raytracer> rt;
    std::unique_ptr<cpu::Shader> sh;
};

glm::mat4 toMat4(const float* m)
{
    glm::mat4 r;
    std::memcpy(&r[0][0], m, 16 * sizeof(float));
    return r;
}

glm::vec4 toVec4(const float* v)
{
    return glm::vec4(v[0], v[1], v[2], v[3]);
}

scene::Material toMaterial(const KajoMaterial& m)
{
    scene::Material r;
    r.ambient = toVec4(m.ambient);
    r.diffuse = toVec4(m.diffuse);
    r.specular = toVec4(m.specular);
    r.emission = toVec4(m.emission);
    r.transparency = toVec4(m.transparency);
    r.specularExponent = m.specularExponent;
    r.refractiveIndex = m.refractiveIndex;
    return r;
}

void fromMaterial(const scene::Material& m, KajoMaterial& r)
{
    std::memcpy(r.ambient, &m.ambient[0], 16);
    std::memcpy(r.diffuse, &m.diffuse[0], 16);
    std::memcpy(r.specular, &m.specular[0], 16);
    std::memcpy(r.emission, &m.emission[0], 16);
    std::memcpy(r.transparency, &m.transparency[0], 16);
    r.specularExponent = m.specularExponent;
    r.refractiveIndex = m.refractiveIndex;
}

void finish(Ref* ref)
{
    ref->cs.reset(new cpu::Scene(ref->ss));
    ref->rt.reset(new cpu::Raytracer(ref->cs.get()));
    ref->sh.reset(new cpu::Shader(ref->cs.get(), ref->rt.get()));
}

// Object id -> 1-based index: planes first, then spheres (the traversal order of
// Raytracer.cpp:131-132); 0 = miss.
int objectIndex(const Ref* ref, intptr_t id)
{
    if (!id)
        return 0;
    const cpu::Scene& s = *ref->cs;
    if (!s.planes.empty()) {
        intptr_t b = reinterpret_cast<intptr_t>(&s.planes[0]);
        intptr_t e = b + static_cast<intptr_t>(s.planes.size() * sizeof(cpu::Plane));
        if (id >= b && id < e)
            return 1 + static_cast<int>((id - b) / sizeof(cpu::Plane));
    }
    if (!s.spheres.empty()) {
        intptr_t b = reinterpret_cast<intptr_t>(&s.spheres[0]);
        intptr_t e = b + static_cast<intptr_t>(s.spheres.size() * sizeof(cpu::Sphere));
        if (id >= b && id < e)
            return 1 + static_cast<int>(s.planes.size()) + static_cast<int>((id - b) / sizeof(cpu::Sphere));
    }
    return -1;
}

void setState(cpu::Random& rng, const uint64_t state[2])
{
    static_assert(sizeof(cpu::Random) == 16, "Random is one __m128i");
    std::memcpy(reinterpret_cast<void*>(&rng), state, 16);
}

void getState(const cpu::Random& rng, uint64_t state[2])
{
    std::memcpy(state, reinterpret_cast<const void*>(&rng), 16);
}

struct CameraBasis
{
    glm::vec3 p1, p2, p3, origin;
};

CameraBasis cameraBasis(const Ref* ref)
{
    // Same four calls as renderer/cpu/Renderer.cpp:29-34, made on glm itself.
    const scene::Camera& camera = ref->cs->camera;
    const glm::vec4 viewport(0, 0, 1, 1);
    CameraBasis b;
    b.p1 = glm::unProject(glm::vec3(0.f, 0.f, 0.f), camera.transform, camera.projection, viewport);
    b.p2 = glm::unProject(glm::vec3(1.f, 0.f, 0.f), camera.transform, camera.projection, viewport);
    b.p3 = glm::unProject(glm::vec3(0.f, 1.f, 0.f), camera.transform, camera.projection, viewport);
    b.origin = glm::vec3(glm::inverse(camera.transform) * glm::vec4(0.f, 0.f, 0.f, 1.f));
    return b;
}

void put3(float* dst, const glm::vec3& v)
{
    dst[0] = v.x;
    dst[1] = v.y;
    dst[2] = v.z;
}

} // namespace

extern "C" {

// ---------------------------------------------------------------------------------
// Scene: either parsed by the reference's own scene::Parser (container only) or built
// from the flat POD (anywhere).
// ---------------------------------------------------------------------------------

void* kref_create_from_file(const char* path, float aspect)
{
    std::unique_ptr<Ref> ref(new Ref);
    if (!scene::Parser::load(ref->ss, path, aspect))
        return nullptr;
    finish(ref.get());
    return ref.release();
}

void* kref_create(const KajoScene* pod)
{
    std::unique_ptr<Ref> ref(new Ref);
    ref->ss.backgroundColor = toVec4(pod->backgroundColor);
    ref->ss.camera.transform = toMat4(pod->camera.transform);
    ref->ss.camera.projection = toMat4(pod->camera.projection);
    for (int i = 0; i < pod->nSpheres; i++) {
        scene::Sphere s;
        s.transform = toMat4(pod->spheres[i].transform);
        s.material = toMaterial(pod->spheres[i].material);
        s.radius = pod->spheres[i].radius;
        ref->ss.spheres.push_back(s);
    }
    for (int i = 0; i < pod->nPlanes; i++) {
        scene::Plane p;
        p.transform = toMat4(pod->planes[i].transform);
        p.material = toMaterial(pod->planes[i].material);
        ref->ss.planes.push_back(p);
    }
    finish(ref.get());
    return ref.release();
}

void kref_destroy(void* h)
{
    delete static_cast<Ref*>(h);
}

void kref_counts(void* h, int* nSpheres, int* nPlanes)
{
    Ref* ref = static_cast<Ref*>(h);
    *nSpheres = static_cast<int>(ref->ss.spheres.size());
    *nPlanes = static_cast<int>(ref->ss.planes.size());
}

// Dump the scene held by the handle as POD. `spheres`/`planes` must have room for the
// counts kref_counts reported; header receives background + camera.
void kref_export(void* h, float background[4], float view[16], float proj[16],
                 KajoSphere* spheres, KajoPlane* planes)
{
    Ref* ref = static_cast<Ref*>(h);
    std::memcpy(background, &ref->ss.backgroundColor[0], 16);
    std::memcpy(view, &ref->ss.camera.transform[0][0], 64);
    std::memcpy(proj, &ref->ss.camera.projection[0][0], 64);
    for (size_t i = 0; i < ref->ss.spheres.size(); i++) {
        std::memcpy(spheres[i].transform, &ref->ss.spheres[i].transform[0][0], 64);
        fromMaterial(ref->ss.spheres[i].material, spheres[i].material);
        spheres[i].radius = ref->ss.spheres[i].radius;
    }
    for (size_t i = 0; i < ref->ss.planes.size(); i++) {
        std::memcpy(planes[i].transform, &ref->ss.planes[i].transform[0][0], 64);
        fromMaterial(ref->ss.planes[i].material, planes[i].material);
    }
}

// Staged per-object data of cpu::Scene (cpu/Scene.cpp:9-13): inverse (16) + determinant,
// planes first then spheres, 17 floats each.
void kref_staged(void* h, float* out)
{
    Ref* ref = static_cast<Ref*>(h);
    for (const cpu::Plane& p : ref->cs->planes) {
        std::memcpy(out, &p.transform.invMatrix[0][0], 64);
        out[16] = p.transform.determinant;
        out += 17;
    }
    for (const cpu::Sphere& s : ref->cs->spheres) {
        std::memcpy(out, &s.transform.invMatrix[0][0], 64);
        out[16] = s.transform.determinant;
        out += 17;
    }
}

void kref_set_depth_limit(int limit)
{
    kajo_ref_depth_limit = limit;
}

// p1, p2, p3, origin (Renderer.cpp:30-34) -> 12 floats.
void kref_camera_basis(void* h, float out[12])
{
    CameraBasis b = cameraBasis(static_cast<Ref*>(h));
    put3(out + 0, b.p1);
    put3(out + 3, b.p2);
    put3(out + 6, b.p3);
    put3(out + 9, b.origin);
}

// ---------------------------------------------------------------------------------
// RNG (cpu/Random.cpp:13-53)
// ---------------------------------------------------------------------------------

void kref_rng_from_seed(unsigned seed, int n, float* out /* n*4 */, uint64_t finalState[2])
{
    cpu::Random rng(seed);
    for (int i = 0; i < n; i++) {
        glm::vec4 v = rng.generate();
        std::memcpy(out + 4 * i, &v[0], 16);
    }
    getState(rng, finalState);
}

void kref_rng_from_state(const uint64_t state[2], int n, float* out, uint64_t finalState[2])
{
    cpu::Random rng;
    setState(rng, state);
    for (int i = 0; i < n; i++) {
        glm::vec4 v = rng.generate();
        std::memcpy(out + 4 * i, &v[0], 16);
    }
    getState(rng, finalState);
}

// flipCoin / russianRoulette (Random.cpp:104-117): out = (value, probability)
void kref_flip_coin(const uint64_t state[2], float p, int* value, float* probability)
{
    cpu::Random rng;
    setState(rng, state);
    cpu::RandomValue<bool> r = rng.flipCoin(p);
    *value = r.value;
    *probability = r.probability;
}

// ---------------------------------------------------------------------------------
// trace (cpu/Raytracer.cpp:126-138)
// ---------------------------------------------------------------------------------

void kref_trace(void* h, int n, const float* origins, const float* dirs,
                int* objIndex, float* t, float* position, float* normal, float* tangent,
                float* binormal)
{
    Ref* ref = static_cast<Ref*>(h);
    for (int i = 0; i < n; i++) {
        cpu::Ray ray;
        ray.origin = glm::vec3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        ray.direction = glm::vec3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        cpu::SurfacePoint sp = ref->rt->trace(ray);
        objIndex[i] = objectIndex(ref, sp.objectId);
        t[i] = ray.maxDistance;
        if (sp.valid()) {
            put3(position + 3 * i, sp.position);
            put3(normal + 3 * i, sp.normal);
            put3(tangent + 3 * i, sp.tangent);
            put3(binormal + 3 * i, sp.binormal);
        } else {
            for (int k = 0; k < 3; k++)
                position[3 * i + k] = normal[3 * i + k] = tangent[3 * i + k] = binormal[3 * i + k] = 0.f;
        }
    }
}

// ---------------------------------------------------------------------------------
// BSDFs (cpu/BSDF.cpp) and SphericalLight (cpu/Light.cpp) on the surface point hit by
// each ray. kind: 0 Lambert, 1 Phong, 2 IdealReflector, 3 IdealTransmission,
// 4 SphericalLight of sphere `lightSphere`. For each ray with a valid hit:
//   dir(3), pdf = generateSample(rng); f(3) = evaluateSample(dir).rgb;
//   pq = sampleProbability(dir)
// Misses are reported with hit = 0 and zeros.
// ---------------------------------------------------------------------------------

void kref_sample(void* h, int kind, int n, const float* origins, const float* dirs,
                 const uint64_t* states, const float color[4], float param, int lightSphere,
                 int* hit, float* outDir, float* outPdf, float* outF, float* outPq,
                 uint64_t* finalStates)
{
    Ref* ref = static_cast<Ref*>(h);
    glm::vec4 c = toVec4(color);
    for (int i = 0; i < n; i++) {
        cpu::Ray ray;
        ray.origin = glm::vec3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        ray.direction = glm::vec3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        cpu::SurfacePoint sp = ref->rt->trace(ray);
        cpu::Random rng;
        setState(rng, states + 2 * i);
        hit[i] = objectIndex(ref, sp.objectId);
        for (int k = 0; k < 3; k++)
            outDir[3 * i + k] = outF[3 * i + k] = 0.f;
        outPdf[i] = outPq[i] = 0.f;
        if (sp.valid()) {
            cpu::RandomValue<glm::vec3> d;
            glm::vec4 f;
            float pq;
            if (kind == 4) {
                const cpu::Sphere* sphere = &ref->cs->spheres[lightSphere];
                cpu::SphericalLight light(&sp, ref->rt.get(), sphere, sphere->material.emission);
                d = light.generateSample(rng);
                f = light.evaluateSample(d.value);
                pq = light.sampleProbability(d.value);
            } else {
                std::unique_ptr<cpu::BSDF> bsdf;
                if (kind == 0)
                    bsdf.reset(new cpu::LambertBSDF(&sp, c));
                else if (kind == 1)
                    bsdf.reset(new cpu::PhongBSDF(&sp, c, param));
                else if (kind == 2)
                    bsdf.reset(new cpu::IdealReflectorBSDF(&sp, c));
                else
                    bsdf.reset(new cpu::IdealTransmissionBSDF(&sp, c, param));
                d = bsdf->generateSample(rng);
                f = bsdf->evaluateSample(d.value);
                pq = bsdf->sampleProbability(d.value);
            }
            put3(outDir + 3 * i, d.value);
            outPdf[i] = d.probability;
            outF[3 * i] = f.x;
            outF[3 * i + 1] = f.y;
            outF[3 * i + 2] = f.z;
            outPq[i] = pq;
        }
        getState(rng, finalStates + 2 * i);
    }
}

// ---------------------------------------------------------------------------------
// shade (cpu/Shader.cpp:113-178) of the point hit by each camera ray, with an injected
// RNG state; returns rgb and the RNG state afterwards (pins the number of draws).
// ---------------------------------------------------------------------------------

void kref_shade(void* h, int n, const float* origins, const float* dirs, const uint64_t* states,
                float* rgb, uint64_t* finalStates)
{
    Ref* ref = static_cast<Ref*>(h);
    for (int i = 0; i < n; i++) {
        cpu::Ray ray;
        ray.origin = glm::vec3(origins[3 * i], origins[3 * i + 1], origins[3 * i + 2]);
        ray.direction = glm::vec3(dirs[3 * i], dirs[3 * i + 1], dirs[3 * i + 2]);
        cpu::Random rng;
        setState(rng, states + 2 * i);
        cpu::SurfacePoint sp = ref->rt->trace(ray);
        glm::vec4 c = ref->sh->shade(sp, rng);
        rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y;
        rgb[3 * i + 2] = c.z;
        getState(rng, finalStates + 2 * i);
    }
}

// ---------------------------------------------------------------------------------
// Frame with per-sample streams. The loop nest and the sample-position arithmetic are
// those of Renderer.cpp:38-71 (n = (int)sqrt(S) strata per axis, sum / S), driven on the
// reference's trace()/shade()/generate(); accum is W*H float4 (row 0 = top), it is
// ADDED to (so callers can continue a progressive render); passes are numbered from
// firstPass (the reference starts at 1).
// ---------------------------------------------------------------------------------

void kref_render(void* h, int W, int H, int S, int firstPass, int nPasses, uint64_t seed,
                 int x0, int y0, int w, int hgt, float* accum)
{
    Ref* ref = static_cast<Ref*>(h);
    CameraBasis b = cameraBasis(ref);
    int samplesPerAxis = sqrt((unsigned)S);
    float pixelWidth = 1.f / W;
    float pixelHeight = 1.f / H;
    float sampleWidth = pixelWidth / samplesPerAxis;
    float sampleHeight = pixelHeight / samplesPerAxis;
    unsigned samples = S;
    cpu::Random rng;

    for (int pass = firstPass; pass < firstPass + nPasses; pass++) {
        for (int y = y0; y < y0 + hgt; y++) {
            for (int x = x0; x < x0 + w; x++) {
                glm::vec4 radiance;
                for (int sampleY = 0; sampleY < samplesPerAxis; sampleY++) {
                    for (int sampleX = 0; sampleX < samplesPerAxis; sampleX++) {
                        uint64_t state[2];
                        kajo_stream_state(seed, (uint32_t)pass, (uint32_t)(sampleY * samplesPerAxis + sampleX),
                                          (uint32_t)(y * W + x), state);
                        setState(rng, state);
                        glm::vec4 offset = rng.generate() * .5f + glm::vec4(.5f);
                        float sx = x * pixelWidth + sampleX * sampleWidth + offset.x * sampleWidth;
                        float sy = (H - y) * pixelHeight + sampleY * sampleHeight + offset.y * sampleHeight;
                        glm::vec3 direction = b.p1 + (b.p2 - b.p1) * sx + (b.p3 - b.p1) * sy - b.origin;
                        direction = glm::normalize(direction);
                        cpu::Ray ray;
                        ray.origin = b.origin;
                        ray.direction = direction;
                        cpu::SurfacePoint sp = ref->rt->trace(ray);
                        radiance += ref->sh->shade(sp, rng);
                    }
                }
                glm::vec4 r = radiance / samples;
                float* dst = accum + 4 * ((size_t)y * W + x);
                dst[0] += r.x;
                dst[1] += r.y;
                dst[2] += r.z;
                dst[3] += r.w;
            }
        }
    }
}

// Resolve as Renderer.cpp:73-75 does, with the reference's own Image statics.
void kref_resolve(int n, const float* accum, int pass, uint32_t* pixels)
{
    for (int i = 0; i < n; i++) {
        glm::vec4 total(accum[4 * i], accum[4 * i + 1], accum[4 * i + 2], accum[4 * i + 3]);
        glm::vec4 pixel = Image::linearToSRGB(glm::clamp(total / pass, glm::vec4(0), glm::vec4(1)));
        pixel.a = 1;
        pixels[i] = Image::colorToRGBA8(pixel);
    }
}

// ---------------------------------------------------------------------------------
// The reference's own hot loop, unmodified: cpu::Renderer::render (Renderer.cpp:25-81)
// on row slices dealt as cpu/Scheduler.cpp:32-42 deals them (slice = (H+1)/nThreads, one
// std::async each), stopped by the observer after `passes` full passes. Serial stream
// per slice, m_samples = 32 (25 paths per pixel per pass). Returns wall seconds; pixels
// (W*H ARGB8) receives the image. The last slice is clamped to the image (the reference
// overruns it, Scheduler.cpp:34-38) so that no out-of-bounds rows are rendered or timed.
// ---------------------------------------------------------------------------------

double kref_render_native(void* h, int W, int H, int passes, int nThreads, uint32_t* pixels)
{
    Ref* ref = static_cast<Ref*>(h);
    Image image(W, H);
    cpu::Renderer renderer(ref->ss);
    // The observer is called once per finished row (Renderer.cpp:77): stop a slice when
    // its last row of pass `passes` has been reported.
    int slice = (H + 1) / nThreads;
    if (slice < 1)
        slice = 1;
    renderer.setObserver([passes, slice, H](int pass, int, int, int y, int, int) {
        int y0 = (y / slice) * slice;
        int last = std::min(y0 + slice, H) - 1;
        return !(pass >= passes && y >= last);
    });
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::future<void>> tasks;
    for (int y = 0; y < H; y += slice) {
        int hgt = std::min(slice, H - y);
        tasks.push_back(std::async(std::launch::async, [&renderer, &image, y, W, hgt] {
            renderer.render(image, 0, y, W, hgt);
        }));
    }
    for (auto& t : tasks)
        t.wait();
    auto t1 = std::chrono::steady_clock::now();
    if (pixels)
        std::memcpy(pixels, image.pixels.get(), sizeof(uint32_t) * W * H);
    return std::chrono::duration<double>(t1 - t0).count();
}

} // extern "C"
