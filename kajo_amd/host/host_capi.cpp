// host_capi.cpp -- C entry points over the host-side scene loader, for tests: parse a Kajo JSON
// scene (or build the reference's test scene) into the flat arrays of include/kajo_scene.h.
#include <cstring>
#include <string>

#include "kajo_scene.h"
#include "scene/Scene.h"

namespace
{
scene::Scene g_scene;

void exportScene(float background[4], float view[16], float proj[16], KajoSphere* spheres, KajoPlane* planes)
{
    static_assert(sizeof(scene::Material) == sizeof(KajoMaterial), "layout");
    std::memcpy(background, &g_scene.backgroundColor, 16);
    std::memcpy(view, g_scene.camera.transform.m, 64);
    std::memcpy(proj, g_scene.camera.projection.m, 64);
    for (size_t i = 0; i < g_scene.spheres.size(); i++) {
        std::memcpy(spheres[i].transform, g_scene.spheres[i].transform.m, 64);
        std::memcpy(&spheres[i].material, &g_scene.spheres[i].material, sizeof(KajoMaterial));
        spheres[i].radius = g_scene.spheres[i].radius;
    }
    for (size_t i = 0; i < g_scene.planes.size(); i++) {
        std::memcpy(planes[i].transform, g_scene.planes[i].transform.m, 64);
        std::memcpy(&planes[i].material, &g_scene.planes[i].material, sizeof(KajoMaterial));
    }
}
} // namespace

extern "C" {

// Returns 0 on success; counts receive the object counts. text == NULL builds the test scene.
int kajo_host_parse(const char* text, float aspect, int* nSpheres, int* nPlanes)
{
    g_scene = scene::Scene();
    if (!text)
        scene::buildTestScene(g_scene);
    else if (!scene::Parser::loadFromString(g_scene, text, aspect))
        return -1;
    *nSpheres = (int)g_scene.spheres.size();
    *nPlanes = (int)g_scene.planes.size();
    return 0;
}

void kajo_host_export(float background[4], float view[16], float proj[16], KajoSphere* spheres, KajoPlane* planes)
{
    exportScene(background, view, proj, spheres, planes);
}

} // extern "C"
