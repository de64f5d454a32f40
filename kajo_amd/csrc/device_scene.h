// device_scene.h -- the render-ready scene as the HIP kernels read it.
//
// Counterpart of cpu::Scene (renderer/cpu/Scene.h:16-59: per object `matrix`, `invMatrix`,
// `determinant`, material, radius), re-laid for wave64 execution: every lane of a wave tests
// the same primitive at the same time, so the records the closest-hit loop touches ("hot")
// are packed small and staged into LDS where one broadcast read serves 64 lanes; data that
// only the winning hit needs ("cold": shading frames, materials) is fetched per lane
// afterwards.
//
//   plane   hot  float4 row   = row y of inverse(M): (inv[0][1], inv[1][1], inv[2][1], inv[3][1])
//                               -- the only row Raytracer.cpp:74-98 uses (local plane y = 0)
//                float  det
//           cold float4 x 3   = world normal, tangent, binormal (constants of the plane)
//   sphere  hot  translated:  float4 (cx, cy, cz, r*r)          when M is a pure translation
//                general:     3 x float4 rows of inverse(M) + float4 (r*r, det, 0, 0)
//           cold DSphereCold  = centre M*(0,0,0,1), radius, mat3(M)
//   material     DMaterial    = scene::Material (scene/Scene.h:11-23) + the three
//                               per-material coin probabilities of Shader.cpp:124,130-134,153
//
// A pure translation makes inverse(M) = translate(-c) and det = 1 exactly, so the translated
// record gives bit-identical t, normal and position to the general formula.
#ifndef KAJO_DEVICE_SCENE_H
#define KAJO_DEVICE_SCENE_H

#include <stdint.h>

struct DFloat4
{
    float x, y, z, w;
};

// 24 floats = 6 float4, grouped by WHEN the integrator needs them, so that each stage of a vertex is one or two 16-byte
// reads issued together (an LDS round trip is ~100 cycles of a wave that has nothing else to issue; round 2's layout cost the
// vertex block seven of them, one per field group in member order of scene::Material):
//   q0 the three coins of Shader.cpp:124,130-134,153 and the scale of a path the first coin stops
//   q1 emission (every vertex adds it, Shader.cpp:121) and two flags
//   q2 specular colour + index of refraction (the refraction branch, Shader.cpp:137-151)
//   q3 diffuse colour + Phong exponent, q4 the path-weight scales (light / BSDF sampling, Shader.cpp:160-177)
#define KAJO_MAT_IS_LIGHT 1u     /* emission != vec4(0), Shader.cpp:57 */
#define KAJO_MAT_HAS_EXPONENT 2u /* specularExponent != 0: Phong lobe, else the ideal reflector (Shader.cpp:155-158) */
struct DMaterial
{
    float pRR;      // max over rgb of max(diffuse, specular, transparency)   (Shader.cpp:124-125)
    float pT;       // sum(transparency) / (sum d + sum s + sum t)            (Shader.cpp:130-133)
    float pD;       // sum(diffuse) / (sum d + sum s)                         (Shader.cpp:153)
    float sStop;    // 1 / (1 - pRR): Russian roulette said stop              (Shader.cpp:126-127)
    float emission[3];
    uint32_t flags; // KAJO_MAT_*
    float specular[3];
    float ior;
    float diffuse[3];
    float exponent;
    // the path-weight scales of Shader.cpp:146-147,160-177, which depend only on the material and on which way the coins
    // fell; formed on the host in the reference's operation order (exact in both numerics modes)
    float sDiffuse;     // 1 / pRR * 1 / (1 - pT) * 1 / pD
    float sSpecular;    // 1 / pRR * 1 / (1 - pT) * 1 / (1 - pD)
    float sTransparent; // 1 / pRR * 1 / pT
    float sDepth;       // 1 / pRR: depth limit reached
    float transparency[3];
    float pad;
};

struct DSphereCold // 16 floats
{
    float cx, cy, cz, radius;
    float m[9];     // mat3(M), m[3*row + col]: world = m * object
    uint32_t general;
    float invRadius;    // 1 / radius                       (FAST numerics only)
    float invTwoPiR2;   // 1 / (2 pi radius^2): light pdf    (FAST numerics only)
};

// sphereHotOffset[i]: index of the sphere's first float4 in the hot array; bit 31 set = general
#define KAJO_SPHERE_GENERAL 0x80000000u

// Uniform grid over the spheres (large scenes): a conservative culling structure. A ray visits the
// cells it crosses front to back (3D-DDA) and runs the SAME per-sphere intersection arithmetic on
// the spheres registered in each cell; the closest hit, and the "later object wins" tie rule of
// Raytracer.cpp:108-124, are those of the brute-force walk. Planes are always tested one by one.
struct DGrid
{
    int32_t enabled;
    int32_t dim[3];
    float bmin[3], bmax[3];
    float cell[3], invCell[3];
    const uint32_t* cellStart; // [dim.x * dim.y * dim.z + 1]
    const uint16_t* items;     // sphere indices (16 bits: the hot records of 65536 spheres would not fit LDS anyway), ascending within a cell
    int32_t nCells, nItems;
    int32_t inLds;             // the two arrays are staged into LDS behind the hot records
    // The registration margins cover what binary32 rounding lets the reference's sphere test report as a hit (stage.cpp
    // floatHitSlack) for rays that start within sqrt(reach2) of `center`; a ray from farther out (a vertex far away on an open
    // floor) walks every sphere instead. 3e38 when the planes close the scene in (stage.cpp findRoom): no ray starts outside.
    float center[3], reach2;
};

// Per-light visibility lists (large scenes whose spheres are all world-space balls; stage.cpp buildShadowLists). A shadow ray
// asks one thing -- is the closest hit the light it was aimed at (Raytracer::canReach, Raytracer.cpp:140-144) -- and every point
// of it up to the light's surface lies in the convex hull of its origin O and the light's ball (C, r): within r of the segment
// [C, O]. So only spheres within (their radius + r) of that segment can be hit before the light. For every light the directions
// u = (O - C) / |O - C| as seen FROM the light are binned on a cube map (6 faces x n x n), and a bin lists the spheres that come
// that close to some ray C + s u of the bin, sorted by the distance of their near side from C. The query runs the SAME
// per-sphere arithmetic as the closest-hit walk on the light, the planes and the bin's spheres nearer than O, and applies the
// walk's acceptance rule in its order-independent form (closest wins; among equal distances the later object): the answer is
// the brute-force walk's, from a handful of tests instead of a grid walk.
struct DShadowItem // host side (stage.cpp builds and sorts these; tests read them through kajo_hip_stage_shadow_lists)
{
    float key;      // |c_i - C| - (radius_i + margin): no point of the ray nearer to C than this can touch sphere i
    uint32_t index; // sphere index
};

// What the kernels read (round 5: 3.4 MB for 1000 spheres / 16 lights where the 8-byte items and 32-bit starts of round 4 took
// 6.8 MB -- more than one XCD's 4 MB of L2, so every list read missed it):
//   items    one 32-bit word per item: the key in 16 bits -- in units of the light's `keyScale`, ROUNDED DOWN, so a quantised key
//            never exceeds the true bound and the walk can only test more items, never fewer -- over the sphere index in 16
//            (a scene with a grid has at most 65 535 spheres); ascending key within a bin
//   rowBase  [nLights * 6 * n] first item of every row of bins (32 bits)
//   off16    [nLights * 6 * n][n + 1] first item of every bin relative to its row's base, and the row's end (16 bits: a row of 64 bins
//            holds a few hundred items; staging gives up on the lists if one ever exceeded 65 535)
// 1 / keyScale of light k rides in the w of its emission record in LDS (LdsScene::lightEmission).
struct DShadowLists
{
    int32_t enabled;
    int32_t n;                 // bins per cube-face axis
    const uint32_t* rowBase;
    const uint16_t* off16;
    const uint32_t* items;
    const float* invKeyScale;  // [nLights]
};

struct DSceneView // device pointers + counts, passed to the kernels by value
{
    const DFloat4* planeRow;    // [nPlanes]
    const float* planeDet;      // [nPlanes]
    const DFloat4* planeFrame;  // [3 * nPlanes] normal, tangent, binormal
    const DFloat4* sphereHot;   // [nSphereHot]
    const uint32_t* sphereHotOffset; // [nSpheres]
    const DSphereCold* sphereCold;   // [nSpheres]
    const DMaterial* material;  // [nPlanes + nSpheres], index = object id - 1
    const int32_t* light;       // [nLights] sphere indices, scene order
    int32_t nPlanes, nSpheres, nSphereHot, nLights;
    int32_t allTranslated;      // every sphere uses the 1-float4 record
    int32_t planesRigid;        // every plane has |determinant - 1| <= 2^-20 (FAST numerics only)
    float background[3];
    DGrid grid;
    DShadowLists shadow;
    // camera (Renderer.cpp:29-34): p1, p2 - p1, p3 - p1, origin
    float p1[3], dp2[3], dp3[3], origin[3];
};

#endif
