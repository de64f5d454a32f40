// stage.h -- host image of the render-ready scene (see device_scene.h for the layout).
#ifndef KAJO_STAGE_H
#define KAJO_STAGE_H

#include <vector>

#include "device_scene.h"
#include "kajo_scene.h"

namespace kajo
{

struct StagedScene
{
    int nPlanes = 0, nSpheres = 0, allTranslated = 1, planesRigid = 1;
    float background[3] = {0, 0, 0};
    float p1[3], p2[3], p3[3], origin[3]; // Renderer.cpp:30-34
    std::vector<DFloat4> planeRow;
    std::vector<float> planeDet;
    std::vector<DFloat4> planeFrame;
    std::vector<DFloat4> sphereHot;
    std::vector<uint32_t> sphereHotOffset;
    std::vector<DSphereCold> sphereCold;
    std::vector<DMaterial> material;
    std::vector<int32_t> light;
    std::vector<float> invDet; // 17 floats per object, planes first (debug / tests)
    // the bounded convex region the planes leave around the camera, if there is one: every ray of every path starts inside it
    bool roomClosed = false;
    double roomLo[3] = {0, 0, 0}, roomHi[3] = {0, 0, 0};
    // uniform grid over the spheres (built when there are at least `gridMinSpheres` of them)
    bool gridEnabled = false;
    float gridCenter[3] = {0, 0, 0}, gridReach2 = 3e38f; // rays starting farther than sqrt(gridReach2) from the centre walk every sphere
    int gridDim[3] = {0, 0, 0};
    float gridMin[3], gridMax[3], gridCell[3];
    std::vector<uint32_t> gridCellStart;
    std::vector<uint16_t> gridItems;
    // per-light visibility lists for shadow queries (device_scene.h DShadowLists; built with the grid when every sphere is a
    // world-space ball)
    bool shadowEnabled = false;
    int shadowN = 0;
    std::vector<uint32_t> shadowStart;     // [bins + 1], 32-bit, and
    std::vector<DShadowItem> shadowItems;  // float keys: the lists as built (host side: tests, kajo_hip_stage_shadow_lists)
    // ... and as the kernels read them (device_scene.h DShadowLists)
    std::vector<uint32_t> shadowPacked, shadowRowBase;
    std::vector<uint16_t> shadowOff16;
    std::vector<float> shadowInvKeyScale;
};

void stageScene(const KajoScene& scene, StagedScene& out, int gridMinSpheres = 48, bool shadowLists = true);

// Smallest and largest magnitude among the scene's non-zero, finite coordinates -- the elements of the object and camera (view) transforms and
// the sphere radii; (0, 0) if there is none; a NaN or infinite coordinate makes *hi NaN. What kajo_hip_create holds against the range the
// STRICT / EXACT kernels' hand-made IEEE quotient and square root are exact in (integrator.inc.hip kdiv, ksqrt).
void coordinateRange(const KajoScene& scene, float* lo, float* hi);

} // namespace kajo

#endif
