// scene/Parser.h -- loader for Kajo's JSON scene dialect (scene/Parser.h:12-16 in the reference).
#ifndef KAJO_HOST_PARSER_H
#define KAJO_HOST_PARSER_H

#include <string>

namespace scene
{

class Scene;

class Parser
{
public:
    static bool load(Scene& scene, const std::string& fileName, float aspectRatio);
    static bool loadFromString(Scene& scene, const std::string& text, float aspectRatio);
};

} // namespace scene

#endif
