// scene/SceneLoader.cpp -- loader for Kajo's JSON scene dialect (scene::Parser).
//
// Behaviour follows the reference's scene::Parser (scene/Parser.cpp:14-232) so that the same files
// give the same scene::Scene:
//   * lenient JSON: trailing commas are accepted (the reference uses SimpleJSON, and every file
//     under data/ relies on it);
//   * colours  "#rgb" (digits / 15), "#rrggbb" (bytes / 255), "rgb(r, g, b)", "rgba(r, g, b, a)";
//     alpha defaults to 1; the result is raised to 2.2 per component INCLUDING alpha (:70-92);
//     an unrecognised string gives (0,0,0,0);
//   * transforms: a sequence of lookat(9) / translate(3) / scale(3) / rotate(angle_deg, axis)
//     commands, each post-multiplied onto the running matrix (:101-148);
//   * camera: "perspective(fovy_deg, near, far)" with the aspect ratio supplied by the caller
//     (:150-166) -- only the first three numbers are read;
//   * objects need a "type" ("sphere" | "plane"); material keys are optional (:168-211).
// The matrix helpers restate glm 0.9.3.4's lookAt / translate / scale / rotate / perspective
// (third_party/glm/glm/gtc/matrix_transform.inl:32-89,223-245,383-409) in float.

#include <cmath>
#include <cstdlib>
#include <fstream>
#include <map>
#include <memory>
#include <sstream>
#include <string>
#include <vector>

#include "Scene.h"

namespace scene
{

namespace
{

// ---- a small JSON reader --------------------------------------------------------------------
struct Json
{
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    double number = 0;
    bool boolean = false;
    std::string string;
    std::vector<Json> array;
    std::vector<std::pair<std::string, Json>> object;

    const Json* get(const std::string& key) const
    {
        const Json* found = nullptr;
        for (const auto& kv : object)
            if (kv.first == key)
                found = &kv.second; // the last duplicate wins, as in a map assignment
        return found;
    }
};

class Reader
{
public:
    explicit Reader(const std::string& s): m_s(s) {}

    bool parse(Json& out)
    {
        skip();
        if (!value(out))
            return false;
        skip();
        return m_pos == m_s.size();
    }

private:
    const std::string& m_s;
    size_t m_pos = 0;

    void skip()
    {
        while (m_pos < m_s.size() && (m_s[m_pos] == ' ' || m_s[m_pos] == '\t' || m_s[m_pos] == '\n' || m_s[m_pos] == '\r'))
            m_pos++;
    }

    bool literal(const char* word)
    {
        size_t n = std::char_traits<char>::length(word);
        if (m_s.compare(m_pos, n, word) != 0)
            return false;
        m_pos += n;
        return true;
    }

    bool string(std::string& out)
    {
        if (m_pos >= m_s.size() || m_s[m_pos] != '"')
            return false;
        m_pos++;
        out.clear();
        while (m_pos < m_s.size() && m_s[m_pos] != '"') {
            char c = m_s[m_pos++];
            if (c == '\\' && m_pos < m_s.size()) {
                char e = m_s[m_pos++];
                switch (e) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u': // keep the low byte of a \uXXXX escape; scene files are ASCII
                    if (m_pos + 4 <= m_s.size()) {
                        out += (char)std::strtol(m_s.substr(m_pos, 4).c_str(), nullptr, 16);
                        m_pos += 4;
                    }
                    break;
                default: out += e; break;
                }
            } else {
                out += c;
            }
        }
        if (m_pos >= m_s.size())
            return false;
        m_pos++;
        return true;
    }

    bool value(Json& out)
    {
        skip();
        if (m_pos >= m_s.size())
            return false;
        char c = m_s[m_pos];
        if (c == '{') {
            out.kind = Json::Object;
            m_pos++;
            for (;;) {
                skip();
                if (m_pos < m_s.size() && m_s[m_pos] == '}') { // also after a trailing comma
                    m_pos++;
                    return true;
                }
                std::string key;
                if (!string(key))
                    return false;
                skip();
                if (m_pos >= m_s.size() || m_s[m_pos] != ':')
                    return false;
                m_pos++;
                Json v;
                if (!value(v))
                    return false;
                out.object.emplace_back(key, std::move(v));
                skip();
                if (m_pos < m_s.size() && m_s[m_pos] == ',')
                    m_pos++;
            }
        }
        if (c == '[') {
            out.kind = Json::Array;
            m_pos++;
            for (;;) {
                skip();
                if (m_pos < m_s.size() && m_s[m_pos] == ']') {
                    m_pos++;
                    return true;
                }
                Json v;
                if (!value(v))
                    return false;
                out.array.push_back(std::move(v));
                skip();
                if (m_pos < m_s.size() && m_s[m_pos] == ',')
                    m_pos++;
            }
        }
        if (c == '"') {
            out.kind = Json::String;
            return string(out.string);
        }
        if (literal("true")) {
            out.kind = Json::Bool;
            out.boolean = true;
            return true;
        }
        if (literal("false")) {
            out.kind = Json::Bool;
            return true;
        }
        if (literal("null"))
            return true;
        const char* begin = m_s.c_str() + m_pos;
        char* end = nullptr;
        double d = std::strtod(begin, &end);
        if (end == begin)
            return false;
        out.kind = Json::Number;
        out.number = d;
        m_pos += (size_t)(end - begin);
        return true;
    }
};

// ---- numbers inside the mini-languages ---------------------------------------------------------
// `count` floats separated by optional commas (scene/Parser.cpp:26-68); missing ones stay 0.
struct Cursor
{
    const std::string& s;
    size_t pos;
};

void readFloats(Cursor& c, float* out, int count)
{
    for (int i = 0; i < count; i++) {
        while (c.pos < c.s.size() && (c.s[c.pos] == ' ' || c.s[c.pos] == '\t'))
            c.pos++;
        const char* begin = c.s.c_str() + c.pos;
        char* end = nullptr;
        float v = std::strtof(begin, &end);
        if (end == begin)
            return;
        out[i] = v;
        c.pos += (size_t)(end - begin);
        if (c.pos < c.s.size() && c.s[c.pos] == ',')
            c.pos++;
    }
}

int hexToInt(char d)
{
    if (d >= '0' && d <= '9')
        return d - '0';
    if (d >= 'A' && d <= 'F')
        return d - 'A' + 10;
    if (d >= 'a' && d <= 'f')
        return d - 'a' + 10;
    return 0;
}

Vec4 parseColor(const std::string& v)
{
    float c[4] = {0, 0, 0, 0};
    if (v.size() == 4 && v[0] == '#') {
        c[0] = hexToInt(v[1]) / 15.f;
        c[1] = hexToInt(v[2]) / 15.f;
        c[2] = hexToInt(v[3]) / 15.f;
        c[3] = 1;
    } else if (v.size() == 7 && v[0] == '#') {
        c[0] = (hexToInt(v[1]) * 16 + hexToInt(v[2])) / 255.f;
        c[1] = (hexToInt(v[3]) * 16 + hexToInt(v[4])) / 255.f;
        c[2] = (hexToInt(v[5]) * 16 + hexToInt(v[6])) / 255.f;
        c[3] = 1;
    } else if (v.size() >= 6 && v.compare(0, 5, "rgba(") == 0) {
        Cursor cur{v, 5};
        readFloats(cur, c, 4);
    } else if (v.size() >= 5 && v.compare(0, 4, "rgb(") == 0) {
        Cursor cur{v, 4};
        readFloats(cur, c, 3);
        c[3] = 1;
    }
    Vec4 r;
    r.x = std::pow(c[0], 2.2f); // srgbToLinear, scene/Parser.cpp:70-73
    r.y = std::pow(c[1], 2.2f);
    r.z = std::pow(c[2], 2.2f);
    r.w = std::pow(c[3], 2.2f);
    return r;
}

// ---- matrices (column-major, e(c, r) = m[4c + r]) ---------------------------------------------
float& e(Mat4& a, int c, int r) { return a.m[4 * c + r]; }
float e(const Mat4& a, int c, int r) { return a.m[4 * c + r]; }

Mat4 multiply(const Mat4& a, const Mat4& b) // glm mat4 * mat4
{
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            e(r, c, i) = e(a, 0, i) * e(b, c, 0) + e(a, 1, i) * e(b, c, 1) + e(a, 2, i) * e(b, c, 2) + e(a, 3, i) * e(b, c, 3);
    return r;
}

void normalize3(float v[3])
{
    float sqr = v[0] * v[0] + v[1] * v[1] + v[2] * v[2];
    float inv = 1.0f / std::sqrt(sqr);
    v[0] *= inv;
    v[1] *= inv;
    v[2] *= inv;
}

void cross3(const float a[3], const float b[3], float out[3])
{
    out[0] = a[1] * b[2] - b[1] * a[2];
    out[1] = a[2] * b[0] - b[2] * a[0];
    out[2] = a[0] * b[1] - b[0] * a[1];
}

float dot3(const float a[3], const float b[3]) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }

Mat4 translate(const Mat4& m, const float v[3]) // matrix_transform.inl:32-42
{
    Mat4 r = m;
    for (int i = 0; i < 4; i++)
        e(r, 3, i) = e(m, 0, i) * v[0] + e(m, 1, i) * v[1] + e(m, 2, i) * v[2] + e(m, 3, i);
    return r;
}

Mat4 scale(const Mat4& m, const float v[3]) // :82-94
{
    Mat4 r;
    for (int i = 0; i < 4; i++) {
        e(r, 0, i) = e(m, 0, i) * v[0];
        e(r, 1, i) = e(m, 1, i) * v[1];
        e(r, 2, i) = e(m, 2, i) * v[2];
        e(r, 3, i) = e(m, 3, i);
    }
    return r;
}

float radians(float degrees) { return degrees * (float(3.1415926535897932384626433832795) / 180.f); }

Mat4 rotate(const Mat4& m, float angle, const float v[3]) // :44-80, angle in degrees
{
    float a = radians(angle);
    float c = std::cos(a);
    float s = std::sin(a);
    float axis[3] = {v[0], v[1], v[2]};
    normalize3(axis);
    float temp[3] = {(1.f - c) * axis[0], (1.f - c) * axis[1], (1.f - c) * axis[2]};
    float R[3][3];
    R[0][0] = c + temp[0] * axis[0];
    R[0][1] = 0 + temp[0] * axis[1] + s * axis[2];
    R[0][2] = 0 + temp[0] * axis[2] - s * axis[1];
    R[1][0] = 0 + temp[1] * axis[0] - s * axis[2];
    R[1][1] = c + temp[1] * axis[1];
    R[1][2] = 0 + temp[1] * axis[2] + s * axis[0];
    R[2][0] = 0 + temp[2] * axis[0] + s * axis[1];
    R[2][1] = 0 + temp[2] * axis[1] - s * axis[0];
    R[2][2] = c + temp[2] * axis[2];
    Mat4 r;
    for (int k = 0; k < 3; k++)
        for (int i = 0; i < 4; i++)
            e(r, k, i) = e(m, 0, i) * R[k][0] + e(m, 1, i) * R[k][1] + e(m, 2, i) * R[k][2];
    for (int i = 0; i < 4; i++)
        e(r, 3, i) = e(m, 3, i);
    return r;
}

Mat4 lookAt(const float eye[3], const float center[3], const float up[3]) // :383-409
{
    float f[3] = {center[0] - eye[0], center[1] - eye[1], center[2] - eye[2]};
    normalize3(f);
    float u[3] = {up[0], up[1], up[2]};
    normalize3(u);
    float s[3];
    cross3(f, u, s);
    normalize3(s);
    cross3(s, f, u);
    Mat4 r;
    e(r, 0, 0) = s[0];
    e(r, 1, 0) = s[1];
    e(r, 2, 0) = s[2];
    e(r, 0, 1) = u[0];
    e(r, 1, 1) = u[1];
    e(r, 2, 1) = u[2];
    e(r, 0, 2) = -f[0];
    e(r, 1, 2) = -f[1];
    e(r, 2, 2) = -f[2];
    e(r, 3, 0) = -dot3(s, eye);
    e(r, 3, 1) = -dot3(u, eye);
    e(r, 3, 2) = dot3(f, eye);
    return r;
}

Mat4 perspective(float fovy, float aspect, float zNear, float zFar) // :223-245, fovy in degrees
{
    float range = std::tan(radians(fovy / 2.f)) * zNear;
    float left = -range * aspect;
    float right = range * aspect;
    float bottom = -range;
    float top = range;
    Mat4 r;
    for (float& x : r.m)
        x = 0;
    e(r, 0, 0) = (2.f * zNear) / (right - left);
    e(r, 1, 1) = (2.f * zNear) / (top - bottom);
    e(r, 2, 2) = -(zFar + zNear) / (zFar - zNear);
    e(r, 2, 3) = -1.f;
    e(r, 3, 2) = -(2.f * zFar * zNear) / (zFar - zNear);
    return r;
}

// scene/Parser.cpp:101-148
Mat4 parseTransform(const std::string& v)
{
    Mat4 result;
    size_t pos = 0;
    while (pos < v.size()) {
        size_t open = v.find('(', pos);
        std::string command = v.substr(pos, open == std::string::npos ? std::string::npos : open - pos);
        if (open == std::string::npos)
            break;
        Cursor cur{v, open + 1};
        if (command == "lookat") {
            float p[9] = {0};
            readFloats(cur, p, 9);
            result = multiply(result, lookAt(p, p + 3, p + 6));
        } else if (command == "translate") {
            float p[3] = {0};
            readFloats(cur, p, 3);
            result = translate(result, p);
        } else if (command == "scale") {
            float p[3] = {0};
            readFloats(cur, p, 3);
            result = scale(result, p);
        } else if (command == "rotate") {
            float p[4] = {0};
            readFloats(cur, p, 4);
            result = rotate(result, p[0], p + 1);
        }
        // the reference then reads one whitespace-delimited word (the ")") and skips blanks
        pos = cur.pos;
        while (pos < v.size() && v[pos] != ' ' && v[pos] != '\t' && v[pos] != '\n')
            pos++;
        while (pos < v.size() && v[pos] == ' ')
            pos++;
    }
    return result;
}

bool asString(const Json* j, std::string& out)
{
    if (!j || j->kind != Json::String)
        return false;
    out = j->string;
    return true;
}

} // namespace

bool Parser::loadFromString(Scene& scene, const std::string& text, float aspectRatio)
{
    Json root;
    Reader reader(text);
    if (!reader.parse(root) || root.kind != Json::Object)
        return false;

    std::string s;
    if (asString(root.get("background"), s))
        scene.backgroundColor = parseColor(s);

    if (const Json* cam = root.get("camera")) {
        Camera camera;
        if (asString(cam->get("projection"), s) && s.compare(0, 12, "perspective(") == 0) {
            float p[3] = {0};
            Cursor cur{s, 12};
            readFloats(cur, p, 3);
            camera.projection = perspective(p[0], aspectRatio, p[1], p[2]);
        }
        if (asString(cam->get("transform"), s))
            camera.transform = parseTransform(s);
        scene.camera = camera;
    }

    if (const Json* objects = root.get("objects")) {
        for (const Json& o : objects->array) {
            std::string type;
            if (!asString(o.get("type"), type))
                continue;
            Material material;
            if (asString(o.get("diffuse"), s))
                material.diffuse = parseColor(s);
            if (asString(o.get("specular"), s))
                material.specular = parseColor(s);
            if (const Json* j = o.get("specularExponent"))
                material.specularExponent = (float)j->number;
            if (asString(o.get("emission"), s))
                material.emission = parseColor(s);
            if (asString(o.get("transparency"), s))
                material.transparency = parseColor(s);
            if (const Json* j = o.get("refractiveIndex"))
                material.refractiveIndex = (float)j->number;
            Mat4 transform;
            if (asString(o.get("transform"), s))
                transform = parseTransform(s);
            if (type == "sphere") {
                Sphere sphere;
                const Json* r = o.get("radius");
                sphere.radius = r ? (float)r->number : 0.f;
                sphere.material = material;
                sphere.transform = transform;
                scene.spheres.push_back(sphere);
            } else if (type == "plane") {
                Plane plane;
                plane.material = material;
                plane.transform = transform;
                scene.planes.push_back(plane);
            }
        }
    }
    return true;
}

bool Parser::load(Scene& scene, const std::string& fileName, float aspectRatio)
{
    std::ifstream in(fileName);
    if (!in)
        return false;
    std::stringstream source;
    source << in.rdbuf();
    return loadFromString(scene, source.str(), aspectRatio);
}

// renderer/Main.cpp:13-95: four unit spheres (white glass-ish, red with exponent 20, green, blue),
// a small emitter, a grey floor and five white walls, 4:3 camera.
void buildTestScene(Scene& scene)
{
    const float colors[4][4] = {{1, 1, 1, 1}, {.8f, .1f, .1f, 1}, {.1f, .8f, .1f, 1}, {.1f, .1f, .8f, 1}};
    auto vec = [](float x, float y, float z, float w) {
        Vec4 v;
        v.x = x;
        v.y = y;
        v.z = z;
        v.w = w;
        return v;
    };
    for (int i = 0; i < 4; i++) {
        Sphere sphere;
        sphere.radius = 1.f;
        const float* c = colors[i % 4];
        sphere.material.ambient = vec(c[0] * 0.1f, c[1] * 0.1f, c[2] * 0.1f, c[3] * 0.1f);
        sphere.material.diffuse = vec(c[0], c[1], c[2], c[3]);
        if (i % 4 == 1)
            sphere.material.specularExponent = 20;
        if (i % 4 == 0) {
            sphere.material.transparency = vec(0.9f, 0.9f, 0.9f, 0.9f);
            sphere.material.refractiveIndex = 1.5f;
        }
        const float off[3] = {(float)(i * 3 - 2), 0, i * .5f};
        sphere.transform = translate(sphere.transform, off);
        scene.spheres.push_back(sphere);
    }
    {
        Sphere sphere;
        sphere.radius = .3f;
        sphere.material.emission = vec(1 * 8.f, 1 * 8.f, 1 * 8.f, 0 * 8.f);
        const float off[3] = {0, -1.5f, 2};
        sphere.transform = translate(sphere.transform, off);
        scene.spheres.push_back(sphere);
    }
    const Vec4 groundDiffuse = vec(.4f, .4f, .4f, 1);
    const Vec4 ambient = vec(.4f * 0.05f, .4f * 0.05f, .4f * 0.05f, 1 * 0.05f);
    {
        Plane ground;
        const float off[3] = {0, 1, 0};
        ground.transform = translate(ground.transform, off);
        ground.material.diffuse = groundDiffuse;
        ground.material.ambient = ambient;
        scene.planes.push_back(ground);
    }
    struct Wall { float angle, ax, ay, az, ty; };
    const Wall walls[5] = {{-90.f, 1, 0, 0, 2}, {-90.f, 0, 0, 1, 10}, {90.f, 0, 0, 1, 8}, {90.f, 1, 0, 0, 6}, {180.f, 1, 0, 0, 2}};
    for (const Wall& w : walls) {
        Plane wall;
        const float axis[3] = {w.ax, w.ay, w.az};
        const float off[3] = {0, w.ty, 0};
        wall.transform = rotate(wall.transform, w.angle, axis);
        wall.transform = translate(wall.transform, off);
        wall.material.diffuse = vec(1.f, 1.f, 1.f, 1);
        wall.material.ambient = ambient;
        scene.planes.push_back(wall);
    }
    const float eye[3] = {-6, -0.8f, 4}, center[3] = {0, 0, 0}, up[3] = {0, -1, 0};
    scene.camera.projection = perspective(45.f, 4.f / 3.f, .1f, 100.f);
    scene.camera.transform = lookAt(eye, center, up);
    scene.backgroundColor = vec(.0f, .0f, .0f, 1);
}

} // namespace scene
