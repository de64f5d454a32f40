/*
 * kajo_scene.h -- flat POD image of Kajo's scene model, as handed across the C ABI.
 *
 * Field for field this is the reference's scene::Scene (scene/Scene.h:11-62):
 *   scene::Material  (scene/Scene.h:11-23)  -> KajoMaterial
 *   scene::Sphere    (scene/Scene.h:25-31)  -> KajoSphere
 *   scene::Plane     (scene/Scene.h:33-38)  -> KajoPlane
 *   scene::Camera    (scene/Scene.h:40-45)  -> KajoCamera
 *   scene::Scene     (scene/Scene.h:50-62)  -> KajoScene
 * Matrices are 16 floats, column-major (glm::mat4 memory order: m[col][row]).
 * Colours are linear RGBA exactly as scene::Parser leaves them (Parser.cpp:70-92).
 * Only data lives here: no functions, no C++, no HIP or torch types.
 */
#ifndef KAJO_SCENE_H
#define KAJO_SCENE_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct KajoMaterial {
    float ambient[4];      /* unused by the cpu integrator, carried for layout parity */
    float diffuse[4];
    float specular[4];
    float emission[4];
    float transparency[4];
    float specularExponent; /* 0 => ideal reflector (Shader.cpp:158) */
    float refractiveIndex;  /* default 1 (scene/Scene.cpp:10-14) */
} KajoMaterial;             /* 22 floats */

typedef struct KajoSphere {
    float transform[16];
    KajoMaterial material;
    float radius;
} KajoSphere;               /* 39 floats */

typedef struct KajoPlane {
    float transform[16];    /* local plane y = 0, normal -Y (Raytracer.cpp:74-98) */
    KajoMaterial material;
} KajoPlane;                /* 38 floats */

typedef struct KajoCamera {
    float transform[16];    /* view matrix */
    float projection[16];
} KajoCamera;

typedef struct KajoScene {
    float backgroundColor[4];
    KajoCamera camera;
    int32_t nSpheres;
    int32_t nPlanes;
    const KajoSphere* spheres;
    const KajoPlane* planes;
} KajoScene;

#ifdef __cplusplus
}
#endif

#endif /* KAJO_SCENE_H */
