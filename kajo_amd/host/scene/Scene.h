// scene/Scene.h -- Kajo's scene data model (scene/Scene.h:11-62 in the reference), with plain
// float arrays where the reference uses glm types: Mat4 is a column-major glm::mat4 image, Vec4 a
// glm::vec4 image, so every record has the reference's memory layout and converts to the C ABI's
// KajoScene (include/kajo_scene.h) by memcpy.
#ifndef KAJO_HOST_SCENE_H
#define KAJO_HOST_SCENE_H

#include <string>
#include <vector>

namespace scene
{

struct Vec4
{
    float x = 0, y = 0, z = 0, w = 0;
};

struct Mat4
{
    float m[16] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1}; // glm::mat4() is the identity
};

class Material
{
public:
    Vec4 ambient;
    Vec4 diffuse;
    Vec4 specular;
    Vec4 emission;
    Vec4 transparency;
    float specularExponent = 0; // scene/Scene.cpp:10-14
    float refractiveIndex = 1;
};

class Sphere
{
public:
    Mat4 transform;
    Material material;
    float radius = 0;
};

class Plane
{
public:
    Mat4 transform;
    Material material;
};

class Camera
{
public:
    Mat4 transform;
    Mat4 projection;
};

typedef std::vector<Sphere> SphereList;
typedef std::vector<Plane> PlaneList;

class Scene
{
public:
    Vec4 backgroundColor;
    Camera camera;
    SphereList spheres;
    PlaneList planes;
};

// The test scene the reference builds when started without a scene file (renderer/Main.cpp:13-95).
void buildTestScene(Scene& scene);

// Loader for Kajo's JSON scene dialect (scene::Parser::load in the reference, scene/Parser.h:12-16);
// implemented in scene/SceneLoader.cpp.
class Parser
{
public:
    static bool load(Scene& scene, const std::string& fileName, float aspectRatio);
    static bool loadFromString(Scene& scene, const std::string& text, float aspectRatio);
};

} // namespace scene

#endif
