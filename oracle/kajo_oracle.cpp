// kajo_oracle.cpp -- TEST INFRASTRUCTURE: the parity oracle. Not product code.
//
// A scalar CPU restatement of the per-pixel Monte-Carlo integrator of skyostil/kajo's
// renderer/cpu, written from the algorithm, not from the reference's text. Only tests/,
// __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it; the product
// (kajo_amd/) never does.
//
// PINNING: every entry point below is checked by tests/test_oracle_golden.py against
// known-answer vectors captured from the compiled reference itself (oracle/_ref, built by
// oracle/Makefile from /root/reference; vectors in tests/golden/, generator
// tests/golden/make_golden.py).
//
// Two numerics modes share one algorithm (template parameter M):
//   LibmMath   -- sinf/cosf/asinf/acosf/powf from the C library, as the reference calls them.
//   StrictMath -- the same five functions from include/kajo_strictmath.h, so that the HIP
//                 kernels' STRICT mode can be compared with this oracle bit for bit.
// Everything else is IEEE binary32 (and binary64 exactly where the reference's expressions
// promote through the double constants M_PI / M_1_PI), evaluated in the reference's operand
// order, compiled with -ffp-contract=off -fno-fast-math.
//
// Structural differences from the reference, none of which changes a decision:
//   * object ids are 1-based indices (planes first, then spheres = the traversal order of
//     renderer/cpu/Raytracer.cpp:131-132) instead of object addresses; 0 = miss;
//   * the sphere/plane shading frame is computed once for the closest hit instead of at every
//     accepted intersection (Raytracer.cpp:50-71,87-97 are pure functions of ray, object, t);
//   * Shader::shade's linear recursion (renderer/cpu/Shader.cpp:113-215) is run as a loop
//     carrying a path throughput: w0*(E0 + w1*(E1 + ...)) becomes sum_k (w0*..*w_{k-1})*E_k;
//     this reassociates float products (rounding-level change) but no branch depends on it;
//   * Shader::calculateLightProbabilities (Shader.cpp:88-111) re-traces, for every light, the
//     very ray Shader.cpp:197-200 has just traced; the loop uses that hit instead (SURVEY.md
//     section 8a row 10) -- identical answer, fewer traversals.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <memory>
#include <thread>
#include <vector>

#include "kajo_scene.h"
#include "kajo_stream.h"
#include "kajo_strictmath.h"

namespace
{

// ------------------------------------------------------------------------------------
// numerics policies
// ------------------------------------------------------------------------------------

struct LibmMath
{
    static float sin(float x) { return ::sinf(x); }
    static float cos(float x) { return ::cosf(x); }
    static float asin(float x) { return ::asinf(x); }
    static float acos(float x) { return ::acosf(x); }
    static float pow(float x, float y) { return ::powf(x, y); }
};

struct StrictMath
{
    static float sin(float x) { return kajo_sinf(x); }
    static float cos(float x) { return kajo_cosf(x); }
    static float asin(float x) { return kajo_asinf(x); }
    static float acos(float x) { return kajo_acosf(x); }
    static float pow(float x, float y) { return kajo_powf(x, y); }
};

const double kPi = 3.14159265358979323846;      // M_PI
const double kInvPi = 0.31830988618379067154;   // M_1_PI
const float kSurfaceEpsilon = 0.001f;           // Shader.cpp:23

// ------------------------------------------------------------------------------------
// small vector algebra in glm's operand order (third_party/glm/glm/core/func_geometric.inl)
// ------------------------------------------------------------------------------------

struct V3
{
    float x, y, z;
};

inline V3 v3(float x, float y, float z) { return V3{x, y, z}; }
inline V3 operator+(V3 a, V3 b) { return v3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline V3 operator-(V3 a, V3 b) { return v3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline V3 operator-(V3 a) { return v3(-a.x, -a.y, -a.z); }
inline V3 operator*(V3 a, float s) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(float s, V3 a) { return v3(a.x * s, a.y * s, a.z * s); }
inline V3 operator*(V3 a, V3 b) { return v3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline V3 operator/(V3 a, float s) { return v3(a.x / s, a.y / s, a.z / s); }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; } // func_geometric.inl:157-166
inline V3 cross(V3 a, V3 b)                                                  // :199-211
{
    return v3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y);
}
inline V3 normalize(V3 a) // :239-248 with inversesqrt = 1/sqrt (func_exponential.inl:145-153)
{
    float sqr = a.x * a.x + a.y * a.y + a.z * a.z;
    return a * (1.0f / std::sqrt(sqr));
}
inline float length(V3 a) { return std::sqrt(a.x * a.x + a.y * a.y + a.z * a.z); } // :59-68
inline V3 reflect(V3 I, V3 N) { return I - N * dot(N, I) * 2.0f; }               // :276-283

// ------------------------------------------------------------------------------------
// 4x4 matrices, column-major m[col*4+row] (glm memory order)
// ------------------------------------------------------------------------------------

struct M4
{
    float m[16];
    float at(int c, int r) const { return m[c * 4 + r]; }
};

// third_party/glm/glm/core/type_mat4x4.inl:757-779: each result column is the left
// matrix's columns weighted by one right column, summed left to right.
M4 mul(const M4& a, const M4& b)
{
    M4 r;
    for (int j = 0; j < 4; j++)
        for (int i = 0; i < 4; i++)
            r.m[j * 4 + i] = a.at(0, i) * b.at(j, 0) + a.at(1, i) * b.at(j, 1) + a.at(2, i) * b.at(j, 2) +
                             a.at(3, i) * b.at(j, 3);
    return r;
}

// type_mat4x4.inl:689-700
void mulVec4(const M4& a, const float v[4], float out[4])
{
    for (int i = 0; i < 4; i++)
        out[i] = a.at(0, i) * v[0] + a.at(1, i) * v[1] + a.at(2, i) * v[2] + a.at(3, i) * v[3];
}

// glm::mat3(m) * v (type_mat3x3.inl): rows of the upper-left 3x3
V3 mulMat3(const M4& a, V3 v)
{
    return v3(a.at(0, 0) * v.x + a.at(1, 0) * v.y + a.at(2, 0) * v.z,
              a.at(0, 1) * v.x + a.at(1, 1) * v.y + a.at(2, 1) * v.z,
              a.at(0, 2) * v.x + a.at(1, 2) * v.y + a.at(2, 2) * v.z);
}

// (m * vec4(p, 1)).xyz
V3 mulPoint(const M4& a, V3 p)
{
    return v3(a.at(0, 0) * p.x + a.at(1, 0) * p.y + a.at(2, 0) * p.z + a.at(3, 0) * 1.0f,
              a.at(0, 1) * p.x + a.at(1, 1) * p.y + a.at(2, 1) * p.z + a.at(3, 1) * 1.0f,
              a.at(0, 2) * p.x + a.at(1, 2) * p.y + a.at(2, 2) * p.z + a.at(3, 2) * 1.0f);
}

// glm::determinant(mat4), core/func_matrix.inl:446-470
float determinant(const M4& a)
{
    float s00 = a.at(2, 2) * a.at(3, 3) - a.at(3, 2) * a.at(2, 3);
    float s01 = a.at(2, 1) * a.at(3, 3) - a.at(3, 1) * a.at(2, 3);
    float s02 = a.at(2, 1) * a.at(3, 2) - a.at(3, 1) * a.at(2, 2);
    float s03 = a.at(2, 0) * a.at(3, 3) - a.at(3, 0) * a.at(2, 3);
    float s04 = a.at(2, 0) * a.at(3, 2) - a.at(3, 0) * a.at(2, 2);
    float s05 = a.at(2, 0) * a.at(3, 1) - a.at(3, 0) * a.at(2, 1);
    float c0 = +(a.at(1, 1) * s00 - a.at(1, 2) * s01 + a.at(1, 3) * s02);
    float c1 = -(a.at(1, 0) * s00 - a.at(1, 2) * s03 + a.at(1, 3) * s04);
    float c2 = +(a.at(1, 0) * s01 - a.at(1, 1) * s03 + a.at(1, 3) * s05);
    float c3 = -(a.at(1, 0) * s02 - a.at(1, 1) * s04 + a.at(1, 2) * s05);
    return a.at(0, 0) * c0 + a.at(0, 1) * c1 + a.at(0, 2) * c2 + a.at(0, 3) * c3;
}

// glm::inverse(mat4), core/func_matrix.inl:523-580: 2x2 sub-determinants, four cofactor
// columns with alternating signs, determinant from the first row, then a true division.
M4 inverse(const M4& a)
{
    float c00 = a.at(2, 2) * a.at(3, 3) - a.at(3, 2) * a.at(2, 3);
    float c02 = a.at(1, 2) * a.at(3, 3) - a.at(3, 2) * a.at(1, 3);
    float c03 = a.at(1, 2) * a.at(2, 3) - a.at(2, 2) * a.at(1, 3);
    float c04 = a.at(2, 1) * a.at(3, 3) - a.at(3, 1) * a.at(2, 3);
    float c06 = a.at(1, 1) * a.at(3, 3) - a.at(3, 1) * a.at(1, 3);
    float c07 = a.at(1, 1) * a.at(2, 3) - a.at(2, 1) * a.at(1, 3);
    float c08 = a.at(2, 1) * a.at(3, 2) - a.at(3, 1) * a.at(2, 2);
    float c10 = a.at(1, 1) * a.at(3, 2) - a.at(3, 1) * a.at(1, 2);
    float c11 = a.at(1, 1) * a.at(2, 2) - a.at(2, 1) * a.at(1, 2);
    float c12 = a.at(2, 0) * a.at(3, 3) - a.at(3, 0) * a.at(2, 3);
    float c14 = a.at(1, 0) * a.at(3, 3) - a.at(3, 0) * a.at(1, 3);
    float c15 = a.at(1, 0) * a.at(2, 3) - a.at(2, 0) * a.at(1, 3);
    float c16 = a.at(2, 0) * a.at(3, 2) - a.at(3, 0) * a.at(2, 2);
    float c18 = a.at(1, 0) * a.at(3, 2) - a.at(3, 0) * a.at(1, 2);
    float c19 = a.at(1, 0) * a.at(2, 2) - a.at(2, 0) * a.at(1, 2);
    float c20 = a.at(2, 0) * a.at(3, 1) - a.at(3, 0) * a.at(2, 1);
    float c22 = a.at(1, 0) * a.at(3, 1) - a.at(3, 0) * a.at(1, 1);
    float c23 = a.at(1, 0) * a.at(2, 1) - a.at(2, 0) * a.at(1, 1);

    const float f0[4] = {c00, c00, c02, c03};
    const float f1[4] = {c04, c04, c06, c07};
    const float f2[4] = {c08, c08, c10, c11};
    const float f3[4] = {c12, c12, c14, c15};
    const float f4[4] = {c16, c16, c18, c19};
    const float f5[4] = {c20, c20, c22, c23};
    const float v0[4] = {a.at(1, 0), a.at(0, 0), a.at(0, 0), a.at(0, 0)};
    const float v1[4] = {a.at(1, 1), a.at(0, 1), a.at(0, 1), a.at(0, 1)};
    const float v2[4] = {a.at(1, 2), a.at(0, 2), a.at(0, 2), a.at(0, 2)};
    const float v3_[4] = {a.at(1, 3), a.at(0, 3), a.at(0, 3), a.at(0, 3)};
    const float sa[4] = {+1, -1, +1, -1};
    const float sb[4] = {-1, +1, -1, +1};

    M4 r;
    for (int i = 0; i < 4; i++) {
        r.m[0 * 4 + i] = sa[i] * (v1[i] * f0[i] - v2[i] * f1[i] + v3_[i] * f2[i]);
        r.m[1 * 4 + i] = sb[i] * (v0[i] * f0[i] - v2[i] * f3[i] + v3_[i] * f4[i]);
        r.m[2 * 4 + i] = sa[i] * (v0[i] * f1[i] - v1[i] * f3[i] + v3_[i] * f5[i]);
        r.m[3 * 4 + i] = sb[i] * (v0[i] * f2[i] - v1[i] * f4[i] + v2[i] * f5[i]);
    }
    float det = a.at(0, 0) * r.at(0, 0) + a.at(0, 1) * r.at(1, 0) + a.at(0, 2) * r.at(2, 0) + a.at(0, 3) * r.at(3, 0);
    for (int i = 0; i < 16; i++)
        r.m[i] = r.m[i] / det;
    return r;
}

// ------------------------------------------------------------------------------------
// RNG: cpu::Random, SSE2 branch (renderer/cpu/Random.cpp:13-21,27-53)
// ------------------------------------------------------------------------------------

struct Rng
{
    uint64_t lo, hi;

    // _mm_set1_epi16(seed): every 16-bit word = (uint16)seed
    void setSeed(unsigned seed)
    {
        uint64_t w = seed & 0xffffu;
        lo = hi = w | (w << 16) | (w << 32) | (w << 48);
    }

    // shufflelo/shufflehi with 0x1e pick words (2,3,1,0) of each 64-bit half;
    // unpackhi_epi64 puts the two HIGH halves side by side, so both lanes of the addend come
    // from hi: lo += hi; hi += perm(hi), perm([a,b,c,d]) = [c,d,b,a] (a = least significant).
    void step()
    {
        uint64_t h = hi;
        uint64_t p = (h >> 32) | ((h & 0xffff0000ull) << 16) | ((h & 0xffffull) << 48);
        lo = lo + h;
        hi = h + p;
    }

    // _mm_cvtepi32_ps of the four 32-bit lanes times (1.0f / 0x7fffffff); the divisor
    // rounds to 2^31 in binary32, so the scale is exactly 2^-31.
    static float lane(uint32_t bits) { return (float)(int32_t)bits * (1.0f / 0x7fffffff); }

    void generate(float out[4])
    {
        step();
        out[0] = lane((uint32_t)lo);
        out[1] = lane((uint32_t)(lo >> 32));
        out[2] = lane((uint32_t)hi);
        out[3] = lane((uint32_t)(hi >> 32));
    }

    // flipCoin (Random.cpp:111-117): returns value, writes probability
    bool flipCoin(float probability, float* outProbability)
    {
        float g[4];
        generate(g);
        float r = g[0] * .5f + .5f;
        if (probability != 0.0f && r <= probability) {
            *outProbability = probability;
            return true;
        }
        *outProbability = 1 - probability;
        return false;
    }
};

// ------------------------------------------------------------------------------------
// staged scene (cpu::Scene, renderer/cpu/Scene.cpp:9-38) + per-object constants
// ------------------------------------------------------------------------------------

struct Mat
{
    V3 diffuse, specular, emission, transparency;
    float emissionA;
    float exponent, ior;
};

struct OPlane
{
    M4 M, inv;
    float det;
    V3 normal, tangent, binormal; // Raytracer.cpp:91-93, constants of the plane
    Mat mat;
};

struct OSphere
{
    M4 M, inv;
    float det;
    float radius;
    V3 centre; // M * (0,0,0,1), Light.cpp:37
    bool isLight; // emission != vec4(0), Shader.cpp:57
    Mat mat;
};

Mat toMat(const KajoMaterial& m)
{
    Mat r;
    r.diffuse = v3(m.diffuse[0], m.diffuse[1], m.diffuse[2]);
    r.specular = v3(m.specular[0], m.specular[1], m.specular[2]);
    r.emission = v3(m.emission[0], m.emission[1], m.emission[2]);
    r.transparency = v3(m.transparency[0], m.transparency[1], m.transparency[2]);
    r.emissionA = m.emission[3];
    r.exponent = m.specularExponent;
    r.ior = m.refractiveIndex;
    return r;
}

struct alignas(64) Counters // one cache line each: per-thread instances sit in a vector
{
    uint64_t paths, traversals, vertices, primitiveTests;
};

// Optional per-path event log (koracle_debug_path): vertex ids and lobe kinds, for diagnosing a
// mismatching path. Never set during normal rendering.
thread_local std::vector<float>* g_pathLog = nullptr;
inline void logEvent(float code, float a = 0, float b = 0, float c = 0)
{
    if (g_pathLog) {
        g_pathLog->push_back(code);
        g_pathLog->push_back(a);
        g_pathLog->push_back(b);
        g_pathLog->push_back(c);
    }
}

struct Hit
{
    int id; // 0 miss, 1..nPlanes planes, nPlanes+1.. spheres
    float t;
    V3 position, view, normal, tangent, binormal;
};

struct Oracle
{
    std::vector<OPlane> planes;
    std::vector<OSphere> spheres;
    V3 background;
    M4 view, proj;
    V3 p1, p2, p3, origin;
    // Timing only (koracle_render_native): also execute the closest-hit walks that
    // Shader::calculateLightProbabilities repeats for every light (Shader.cpp:88-111), so that the port does the
    // reference's amount of work when it stands in for it as the CPU baseline. Results do not depend on it.
    bool timingRetrace = false;

    explicit Oracle(const KajoScene& s)
    {
        background = v3(s.backgroundColor[0], s.backgroundColor[1], s.backgroundColor[2]);
        std::memcpy(view.m, s.camera.transform, 64);
        std::memcpy(proj.m, s.camera.projection, 64);
        for (int i = 0; i < s.nPlanes; i++) {
            OPlane p;
            std::memcpy(p.M.m, s.planes[i].transform, 64);
            p.inv = inverse(p.M);
            p.det = determinant(p.M);
            p.normal = mulMat3(p.M, -v3(0, 1, 0));
            p.tangent = mulMat3(p.M, v3(1, 0, 0));
            p.binormal = cross(p.normal, p.tangent);
            p.mat = toMat(s.planes[i].material);
            planes.push_back(p);
        }
        for (int i = 0; i < s.nSpheres; i++) {
            OSphere q;
            std::memcpy(q.M.m, s.spheres[i].transform, 64);
            q.inv = inverse(q.M);
            q.det = determinant(q.M);
            q.radius = s.spheres[i].radius;
            const float zero[4] = {0, 0, 0, 1};
            float c[4];
            mulVec4(q.M, zero, c);
            q.centre = v3(c[0], c[1], c[2]);
            q.mat = toMat(s.spheres[i].material);
            q.isLight = !(q.mat.emission.x == 0 && q.mat.emission.y == 0 && q.mat.emission.z == 0 &&
                          q.mat.emissionA == 0);
            spheres.push_back(q);
        }
        cameraBasis();
    }

    // glm::unProject (gtc/matrix_transform.inl:337-356) with viewport (0,0,1,1)
    V3 unProject(V3 win) const
    {
        M4 inv = inverse(mul(proj, view));
        float tmp[4] = {win.x, win.y, win.z, 1.0f};
        tmp[0] = (tmp[0] - 0.0f) / 1.0f;
        tmp[1] = (tmp[1] - 0.0f) / 1.0f;
        for (int i = 0; i < 4; i++)
            tmp[i] = tmp[i] * 2.0f - 1.0f;
        float obj[4];
        mulVec4(inv, tmp, obj);
        return v3(obj[0] / obj[3], obj[1] / obj[3], obj[2] / obj[3]);
    }

    // Renderer.cpp:29-34
    void cameraBasis()
    {
        p1 = unProject(v3(0, 0, 0));
        p2 = unProject(v3(1, 0, 0));
        p3 = unProject(v3(0, 1, 0));
        const float zero[4] = {0, 0, 0, 1};
        float o[4];
        mulVec4(inverse(view), zero, o);
        origin = v3(o[0], o[1], o[2]);
    }

    const Mat& material(int id) const
    {
        int np = (int)planes.size();
        return id <= np ? planes[id - 1].mat : spheres[id - 1 - np].mat;
    }

    // ---------------------------------------------------------------------------------
    // closest hit: Raytracer.cpp:21-138
    // ---------------------------------------------------------------------------------
    Hit trace(V3 O, V3 d, Counters* ctr) const
    {
        const int np = (int)planes.size();
        float tMax = std::numeric_limits<float>::infinity(); // Ray.cpp:10-13
        const float tMin = 0.0f;
        int best = 0;
        float bestT0 = 0.0f;
        if (ctr) {
            ctr->traversals++;
            ctr->primitiveTests += planes.size() + spheres.size();
        }

        for (int i = 0; i < np; i++) { // Raytracer.cpp:74-98
            const OPlane& p = planes[i];
            V3 dir = mulMat3(p.inv, d);
            V3 o = mulPoint(p.inv, O);
            V3 n = v3(0, 1, 0);
            float denom = dot(dir, n);
            if (std::fabs(denom) < std::numeric_limits<float>::epsilon())
                continue;
            float t = -dot(o, n) / denom;
            if (t < 0)
                continue;
            float ts = t * p.det;
            if (ts > tMax || ts < tMin) // Raytracer.cpp:115
                continue;
            tMax = ts;
            best = 1 + i;
        }
        for (int i = 0; i < (int)spheres.size(); i++) { // Raytracer.cpp:21-72
            const OSphere& s = spheres[i];
            V3 dir = mulMat3(s.inv, d);
            V3 o = mulPoint(s.inv, O);
            float a = dot(dir, dir);
            float b = 2 * dot(dir, o);
            float c = dot(o, o) - s.radius * s.radius;
            float discr = b * b - 4 * a * c;
            if (discr < 0)
                continue;
            float q;
            if (b < 0)
                q = (-b - std::sqrt(discr)) * .5f;
            else
                q = (-b + std::sqrt(discr)) * .5f;
            float t0 = q / a;
            float t1 = c / q;
            if (t0 > t1)
                std::swap(t0, t1);
            if (t1 < 0)
                continue;
            if (t0 < 0)
                t0 = t1;
            float ts = t0 * s.det;
            if (ts > tMax || ts < tMin)
                continue;
            tMax = ts;
            best = 1 + np + i;
            bestT0 = t0;
        }

        Hit h;
        h.id = best;
        h.t = tMax;
        h.view = d;
        h.position = h.normal = h.tangent = h.binormal = v3(0, 0, 0);
        if (!best)
            return h;
        if (best <= np) {
            const OPlane& p = planes[best - 1];
            h.normal = p.normal;
            h.tangent = p.tangent;
            h.binormal = p.binormal;
        } else {
            const OSphere& s = spheres[best - 1 - np];
            V3 dir = mulMat3(s.inv, d);
            V3 o = mulPoint(s.inv, O);
            V3 n = o + dir * bestT0;
            n = normalize(mulMat3(s.M, n));
            // Raytracer.cpp:55-63: tangent from the smallest component, by exact equality
            V3 tg;
            float smallest = std::min(n.z, std::min(n.x, n.y));
            if (n.x == smallest)
                tg = v3(0, -n.z, n.y);
            else if (n.y == smallest)
                tg = v3(-n.z, 0, n.x);
            else
                tg = v3(-n.y, n.x, 0);
            tg = normalize(tg);
            h.normal = n;
            h.tangent = tg;
            h.binormal = cross(n, tg);
        }
        h.position = O + d * tMax; // Raytracer.cpp:134-135
        return h;
    }

    // ---------------------------------------------------------------------------------
    // BSDFs (renderer/cpu/BSDF.cpp). kind: 0 Lambert, 1 Phong, 2 IdealReflector,
    // 3 IdealTransmission
    // ---------------------------------------------------------------------------------
    struct Bsdf
    {
        int kind;
        V3 color;
        float param; // exponent or refractive index
    };

    template <class M>
    static V3 bsdfGenerate(const Bsdf& f, const Hit& sp, Rng& rng, float* pdf)
    {
        if (f.kind == 0) { // BSDF.cpp:20-28 + Random.cpp:77-88
            float g[4];
            rng.generate(g);
            float u = .5f * g[0] + .5f;
            float v = .5f * g[1] + .5f;
            float r = std::sqrt(u);
            float phi = (float)((double)(v * 2) * kPi);
            float x = r * M::cos(phi);
            float y = r * M::sin(phi);
            float z = std::sqrt(std::max(0.f, 1.f - u));
            *pdf = (float)((double)z * kInvPi);
            return sp.tangent * x + sp.binormal * y + sp.normal * z;
        }
        if (f.kind == 1) { // BSDF.cpp:48-60 + Random.cpp:90-102
            float g[4];
            rng.generate(g);
            float u = .5f * g[0] + .5f;
            float v = .5f * g[1] + .5f;
            float a = M::acos(M::pow(u, 1.f / (f.param + 1)));
            float phi = (float)(2 * kPi * (double)v);
            V3 s = v3(M::sin(a) * M::cos(phi), M::sin(a) * M::sin(phi), M::cos(a));
            *pdf = (float)((double)(f.param + 1) / (2 * kPi) * (double)M::pow(M::cos(a), f.param));
            V3 R = reflect(sp.view, sp.normal);
            V3 uu = normalize(cross(v3(0, 0, 1), R));
            V3 vv = cross(uu, R);
            // glm::mat3(u, v, R) * s
            return v3(uu.x * s.x + vv.x * s.y + R.x * s.z, uu.y * s.x + vv.y * s.y + R.y * s.z,
                      uu.z * s.x + vv.z * s.y + R.z * s.z);
        }
        if (f.kind == 2) { // BSDF.cpp:82-85
            *pdf = 1.f;
            return reflect(sp.view, sp.normal);
        }
        // BSDF.cpp:105-124
        float cosA = dot(sp.view, sp.normal);
        bool entering = cosA < 0;
        V3 n = entering ? sp.normal : -sp.normal;
        float eta = entering ? 1.f / f.param : f.param / 1.f;
        cosA = dot(sp.view, n);
        *pdf = 1.f;
        if (1 - eta * eta * (1 - cosA * cosA) < 0)
            return reflect(sp.view, n);
        // glm::refract, func_geometric.inl:306-322
        float dv = dot(n, sp.view);
        float k = 1.f - eta * eta * (1.f - dv * dv);
        if (k < 0.f)
            return v3(0, 0, 0);
        return eta * sp.view - (eta * dv + std::sqrt(k)) * n;
    }

    template <class M>
    static V3 bsdfEvaluate(const Bsdf& f, const Hit& sp, V3 dir)
    {
        if (f.kind == 0) // BSDF.cpp:30-33
            return f.color * (float)kInvPi;
        if (f.kind == 1) { // BSDF.cpp:62-67
            V3 R = reflect(sp.view, sp.normal);
            float cosA = std::max(0.f, dot(R, dir));
            float s = (float)((double)(f.param + 1) / (2 * kPi));
            return (s * f.color) * M::pow(cosA, f.param);
        }
        if (f.kind == 2) { // BSDF.cpp:87-91
            float cosA = std::max(0.f, dot(dir, sp.normal));
            return f.color / cosA;
        }
        float cosA = std::fabs(dot(dir, sp.normal)); // BSDF.cpp:126-130
        return f.color / cosA;
    }

    template <class M>
    static float bsdfProbability(const Bsdf& f, const Hit& sp, V3 dir)
    {
        if (f.kind == 0) { // BSDF.cpp:35-39
            float cosT = dot(dir, sp.normal);
            return (float)(kInvPi * (double)cosT);
        }
        if (f.kind == 1) { // BSDF.cpp:69-74
            V3 R = reflect(sp.view, sp.normal);
            float cosA = std::max(0.f, dot(R, dir));
            return (float)((double)(f.param + 1) / (2 * kPi) * (double)M::pow(cosA, f.param));
        }
        return 0.f; // BSDF.cpp:93-96,132-135
    }

    // ---------------------------------------------------------------------------------
    // SphericalLight (renderer/cpu/Light.cpp:26-62)
    // ---------------------------------------------------------------------------------
    template <class M>
    static float solidAngle(const OSphere& s, V3 P)
    {
        float dist = length(s.centre - P);
        if (dist < s.radius)
            return (float)(4 * kPi);
        return (float)(2 * kPi * (double)(1 - M::cos(M::asin(s.radius / dist))));
    }

    template <class M>
    static V3 lightGenerate(const OSphere& s, V3 P, Rng& rng, float* pdf)
    {
        float g[4];
        rng.generate(g);
        float s1 = (g[0] * .5f) + .5f;
        float s2 = (g[1] * .5f) + .5f;
        float s3 = (g[2] * .5f) + .5f;
        float x = s.radius * std::sqrt(s1) * M::cos((float)(2 * kPi * (double)s2));
        float y = s.radius * std::sqrt(s1) * M::sin((float)(2 * kPi * (double)s2));
        float z = std::sqrt(s.radius * s.radius - x * x - y * y) * M::sin((float)(kPi * (double)(s3 - .5f)));
        V3 dir = normalize(s.centre + v3(x, y, z) - P);
        *pdf = 1 / solidAngle<M>(s, P);
        return dir;
    }

    // ---------------------------------------------------------------------------------
    // Shader::shade as a loop (renderer/cpu/Shader.cpp:50-215). `hit` is the surface point
    // of the camera ray. Returns RGB.
    // ---------------------------------------------------------------------------------
    template <class M>
    V3 shade(Hit sp, Rng& rng, int depthLimit, Counters* ctr) const
    {
        const int np = (int)planes.size();
        V3 L = v3(0, 0, 0);
        V3 T = v3(1, 1, 1);
        bool collectEmission = true; // SampleAllObjects
        for (int depth = 0;; depth++) {
            if (!sp.id) { // Shader.cpp:116-117
                L = L + T * background;
                break;
            }
            if (ctr)
                ctr->vertices++;
            logEvent(1, (float)sp.id, (float)depth, sp.t);
            logEvent(5, sp.position.x, sp.position.y, sp.position.z);
            logEvent(6, sp.normal.x, sp.normal.y, sp.normal.z);
            const Mat& m = material(sp.id);
            V3 E = collectEmission ? m.emission : v3(0, 0, 0); // :121

            // Russian roulette, :124-127 + Random.cpp:104-109
            V3 mx = v3(std::max(std::max(m.diffuse.x, m.specular.x), m.transparency.x),
                       std::max(std::max(m.diffuse.y, m.specular.y), m.transparency.y),
                       std::max(std::max(m.diffuse.z, m.specular.z), m.transparency.z));
            float pRR = std::max(mx.x, std::max(mx.y, mx.z));
            float pc;
            bool cont = rng.flipCoin(pRR, &pc);
            if (!cont || depth >= depthLimit) {
                L = L + T * (1 / pc * E);
                break;
            }

            // lobe selection, :130-134
            float totalD = m.diffuse.x + m.diffuse.y + m.diffuse.z;
            float totalS = m.specular.x + m.specular.y + m.specular.z;
            float totalT = m.transparency.x + m.transparency.y + m.transparency.z;
            float pTransp = totalT / (totalD + totalS + totalT);
            float pt;
            bool transparent = rng.flipCoin(pTransp, &pt);

            if (transparent) { // :137-151
                logEvent(2, 3);
                Bsdf f{3, m.specular, m.ior};
                float pdf;
                V3 d = bsdfGenerate<M>(f, sp, rng, &pdf);
                V3 o = sp.position + d * kSurfaceEpsilon;
                logEvent(8, o.x, o.y, o.z);
                logEvent(9, d.x, d.y, d.z);
                Hit next = trace(o, d, ctr);
                V3 w = (1 / pc * 1 / pt * bsdfEvaluate<M>(f, sp, d)) * std::fabs(dot(sp.normal, d));
                L = L + T * (w * E);
                T = T * w;
                sp = next;
                continue;
            }

            float pDiffuse = totalD / (totalD + totalS); // :153-154
            float pd;
            bool diffuse = rng.flipCoin(pDiffuse, &pd);
            Bsdf f;
            if (!diffuse) { // :157-171
                if (m.exponent != 0.0f)
                    f = Bsdf{1, m.specular, m.exponent};
                else
                    f = Bsdf{2, m.specular, 0.f};
            } else {
                f = Bsdf{0, m.diffuse, 0.f}; // :173
            }
            float s = 1 / pc * 1 / pt * 1 / pd;
            logEvent(2, (float)f.kind, s);

            // shadeWithBSDF, :180-215
            V3 Ld = v3(0, 0, 0);
            for (int i = 0; i < (int)spheres.size(); i++) { // sampleLights, :50-86
                const OSphere& light = spheres[i];
                if (!light.isLight)
                    continue;
                if (1 + np + i == sp.id)
                    continue;
                float pl;
                V3 l = lightGenerate<M>(light, sp.position, rng, &pl);
                if (pl == 0.0f)
                    continue;
                V3 so = sp.position + l * kSurfaceEpsilon;
                logEvent(7, l.x, l.y, l.z);
                Hit sh = trace(so, l, ctr);
                if (sh.id != 1 + np + i)
                    continue;
                float pb = bsdfProbability<M>(f, sp, l);
                if (pb == 0.0f)
                    continue;
                Ld = Ld + ((1 / (pb + pl) * bsdfEvaluate<M>(f, sp, l)) * std::max(0.f, dot(sp.normal, l))) *
                              light.mat.emission;
                logEvent(3, pb, pl, Ld.x);
            }

            float p;
            V3 d = bsdfGenerate<M>(f, sp, rng, &p); // :192-194
            if (p == 0.0f) {
                L = L + T * (s * (E + Ld));
                break;
            }
            V3 o = sp.position + d * kSurfaceEpsilon; // :197-200
            logEvent(8, o.x, o.y, o.z);
            logEvent(9, d.x, d.y, d.z);
            Hit next = trace(o, d, ctr);

            // calculateLightProbabilities, :88-111 -- every term re-traces (o, d); only the
            // light that (o, d) actually hits can contribute.
            float pL = 0;
            for (int i = 0; i < (int)spheres.size(); i++) {
                const OSphere& light = spheres[i];
                if (!light.isLight || 1 + np + i == sp.id)
                    continue;
                if (timingRetrace && trace(o, d, nullptr).id != next.id) // Raytracer.cpp:140-144, same ray: never taken
                    continue;
                if (next.id != 1 + np + i)
                    continue;
                pL += 1 / solidAngle<M>(light, sp.position);
            }

            V3 wb = (1 / (pL + p) * bsdfEvaluate<M>(f, sp, d)) * std::max(0.f, dot(sp.normal, d)); // :208-212
            logEvent(4, p, pL, wb.x);
            L = L + T * (s * (E + Ld));
            T = T * (s * wb);
            collectEmission = false; // SampleNonEmissiveObjects
            sp = next;
        }
        return L;
    }

    // ---------------------------------------------------------------------------------
    // one camera path (Renderer.cpp:55-66): jitter draw, ray, trace, shade
    // ---------------------------------------------------------------------------------
    struct FrameConsts
    {
        int W, H, n;
        float pixelWidth, pixelHeight, sampleWidth, sampleHeight;
    };

    static FrameConsts frameConsts(int W, int H, int S)
    {
        FrameConsts c;
        c.W = W;
        c.H = H;
        c.n = (int)std::sqrt((double)(unsigned)S); // int samplesPerAxis = sqrt(m_samples), :38
        c.pixelWidth = 1.f / W;
        c.pixelHeight = 1.f / H;
        c.sampleWidth = c.pixelWidth / c.n;
        c.sampleHeight = c.pixelHeight / c.n;
        return c;
    }

    template <class M>
    V3 cameraPath(const FrameConsts& c, int x, int y, int sampleX, int sampleY, Rng& rng, int depthLimit,
                  Counters* ctr) const
    {
        float g[4];
        rng.generate(g);
        float offX = g[0] * .5f + .5f;
        float offY = g[1] * .5f + .5f;
        float sx = x * c.pixelWidth + sampleX * c.sampleWidth + offX * c.sampleWidth;
        float sy = (c.H - y) * c.pixelHeight + sampleY * c.sampleHeight + offY * c.sampleHeight;
        V3 direction = p1 + (p2 - p1) * sx + (p3 - p1) * sy - origin;
        direction = normalize(direction);
        if (ctr)
            ctr->paths++;
        Hit sp = trace(origin, direction, ctr);
        return shade<M>(sp, rng, depthLimit, ctr);
    }

    // rows [y0, y1) step `stride`: per-sample streams; accum += sum / S per pass
    template <class M>
    void renderRows(const FrameConsts& c, int S, int firstPass, int nPasses, uint64_t seed, int depthLimit,
                    int x0, int w, int yBegin, int yEnd, int stride, float* accum, Counters* ctr) const
    {
        Rng rng;
        for (int pass = firstPass; pass < firstPass + nPasses; pass++) {
            for (int y = yBegin; y < yEnd; y += stride) {
                for (int x = x0; x < x0 + w; x++) {
                    V3 radiance = v3(0, 0, 0);
                    for (int sampleY = 0; sampleY < c.n; sampleY++) {
                        for (int sampleX = 0; sampleX < c.n; sampleX++) {
                            uint64_t st[2];
                            kajo_stream_state(seed, (uint32_t)pass, (uint32_t)(sampleY * c.n + sampleX),
                                              (uint32_t)(y * c.W + x), st);
                            rng.lo = st[0];
                            rng.hi = st[1];
                            radiance = radiance + cameraPath<M>(c, x, y, sampleX, sampleY, rng, depthLimit, ctr);
                        }
                    }
                    V3 r = radiance / (float)(unsigned)S; // Renderer.cpp:71
                    float* dst = accum + 4 * ((size_t)y * c.W + x);
                    dst[0] += r.x;
                    dst[1] += r.y;
                    dst[2] += r.z;
                }
            }
        }
    }
};

struct Handle
{
    std::unique_ptr<Oracle> o;
    int math; // 0 libm, 1 strict
};

V3 ld3(const float* p, int i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
void st3(float* p, int i, V3 v)
{
    p[3 * i] = v.x;
    p[3 * i + 1] = v.y;
    p[3 * i + 2] = v.z;
}

template <class M>
uint32_t resolvePixel(const float* a, int pass)
{
    float c[3];
    for (int k = 0; k < 3; k++) {
        float v = a[k] / (float)pass;                  // Renderer.cpp:73
        v = std::min(std::max(v, 0.f), 1.f);           // glm::clamp = min(max(x, lo), hi)
        c[k] = M::pow(v, 1 / 2.2f);                    // Image.cpp:14-17
    }
    int r = (int)(c[0] * 255.f + .5f);                 // Image.cpp:19-27
    int g = (int)(c[1] * 255.f + .5f);
    int b = (int)(c[2] * 255.f + .5f);
    int al = (int)(1.f * 255.f + .5f);
    return ((uint32_t)al << 24) | ((uint32_t)r << 16) | ((uint32_t)g << 8) | (uint32_t)b;
}

} // namespace

extern "C" {

void* koracle_create(const KajoScene* scene, int math)
{
    Handle* h = new Handle;
    h->o.reset(new Oracle(*scene));
    h->math = math;
    return h;
}

void koracle_destroy(void* h)
{
    delete static_cast<Handle*>(h);
}

// inverse(16) + determinant per object, planes first (cf. kref_staged)
void koracle_staged(void* hh, float* out)
{
    Oracle& o = *static_cast<Handle*>(hh)->o;
    for (const OPlane& p : o.planes) {
        std::memcpy(out, p.inv.m, 64);
        out[16] = p.det;
        out += 17;
    }
    for (const OSphere& s : o.spheres) {
        std::memcpy(out, s.inv.m, 64);
        out[16] = s.det;
        out += 17;
    }
}

void koracle_camera_basis(void* hh, float out[12])
{
    Oracle& o = *static_cast<Handle*>(hh)->o;
    st3(out, 0, o.p1);
    st3(out, 1, o.p2);
    st3(out, 2, o.p3);
    st3(out, 3, o.origin);
}

void koracle_rng_from_seed(unsigned seed, int n, float* out, uint64_t finalState[2])
{
    Rng r;
    r.setSeed(seed);
    for (int i = 0; i < n; i++)
        r.generate(out + 4 * i);
    finalState[0] = r.lo;
    finalState[1] = r.hi;
}

void koracle_rng_from_state(const uint64_t state[2], int n, float* out, uint64_t finalState[2])
{
    Rng r;
    r.lo = state[0];
    r.hi = state[1];
    for (int i = 0; i < n; i++)
        r.generate(out + 4 * i);
    finalState[0] = r.lo;
    finalState[1] = r.hi;
}

void koracle_flip_coin(const uint64_t state[2], float p, int* value, float* probability)
{
    Rng r;
    r.lo = state[0];
    r.hi = state[1];
    *value = r.flipCoin(p, probability);
}

void koracle_trace(void* hh, int n, const float* origins, const float* dirs, int* objIndex, float* t,
                   float* position, float* normal, float* tangent, float* binormal)
{
    Oracle& o = *static_cast<Handle*>(hh)->o;
    for (int i = 0; i < n; i++) {
        Hit h = o.trace(ld3(origins, i), ld3(dirs, i), nullptr);
        objIndex[i] = h.id;
        t[i] = h.t;
        st3(position, i, h.position);
        st3(normal, i, h.normal);
        st3(tangent, i, h.tangent);
        st3(binormal, i, h.binormal);
    }
}

// kinds as in oracle/ref_harness.cpp kref_sample: 0 Lambert, 1 Phong, 2 IdealReflector,
// 3 IdealTransmission, 4 SphericalLight of sphere `lightSphere`
void koracle_sample(void* hh, int kind, int n, const float* origins, const float* dirs, const uint64_t* states,
                    const float color[4], float param, int lightSphere, int* hit, float* outDir, float* outPdf,
                    float* outF, float* outPq, uint64_t* finalStates)
{
    Handle* H = static_cast<Handle*>(hh);
    Oracle& o = *H->o;
    for (int i = 0; i < n; i++) {
        Hit sp = o.trace(ld3(origins, i), ld3(dirs, i), nullptr);
        Rng rng;
        rng.lo = states[2 * i];
        rng.hi = states[2 * i + 1];
        hit[i] = sp.id;
        V3 d = v3(0, 0, 0), f = v3(0, 0, 0);
        float pdf = 0, pq = 0;
        if (sp.id) {
            if (kind == 4) {
                const OSphere& light = o.spheres[lightSphere];
                d = H->math ? Oracle::lightGenerate<StrictMath>(light, sp.position, rng, &pdf)
                            : Oracle::lightGenerate<LibmMath>(light, sp.position, rng, &pdf);
                f = light.mat.emission;
                pq = 1 / (H->math ? Oracle::solidAngle<StrictMath>(light, sp.position)
                                  : Oracle::solidAngle<LibmMath>(light, sp.position));
            } else {
                Oracle::Bsdf b{kind, v3(color[0], color[1], color[2]), param};
                if (H->math) {
                    d = Oracle::bsdfGenerate<StrictMath>(b, sp, rng, &pdf);
                    f = Oracle::bsdfEvaluate<StrictMath>(b, sp, d);
                    pq = Oracle::bsdfProbability<StrictMath>(b, sp, d);
                } else {
                    d = Oracle::bsdfGenerate<LibmMath>(b, sp, rng, &pdf);
                    f = Oracle::bsdfEvaluate<LibmMath>(b, sp, d);
                    pq = Oracle::bsdfProbability<LibmMath>(b, sp, d);
                }
            }
        }
        st3(outDir, i, d);
        st3(outF, i, f);
        outPdf[i] = pdf;
        outPq[i] = pq;
        finalStates[2 * i] = rng.lo;
        finalStates[2 * i + 1] = rng.hi;
    }
}

void koracle_shade(void* hh, int n, const float* origins, const float* dirs, const uint64_t* states,
                   int depthLimit, float* rgb, uint64_t* finalStates)
{
    Handle* H = static_cast<Handle*>(hh);
    Oracle& o = *H->o;
    for (int i = 0; i < n; i++) {
        Rng rng;
        rng.lo = states[2 * i];
        rng.hi = states[2 * i + 1];
        Hit sp = o.trace(ld3(origins, i), ld3(dirs, i), nullptr);
        V3 c = H->math ? o.shade<StrictMath>(sp, rng, depthLimit, nullptr)
                       : o.shade<LibmMath>(sp, rng, depthLimit, nullptr);
        st3(rgb, i, c);
        finalStates[2 * i] = rng.lo;
        finalStates[2 * i + 1] = rng.hi;
    }
}

// Per-sample-stream frame; accum (W*H float4, row 0 = top) is ADDED to; rows of the
// rectangle are dealt round-robin to nThreads host threads (the result does not depend on
// the split). counters (4 x uint64: paths, traversals, vertices, primitive tests) may be
// null. Returns wall seconds.
double koracle_render(void* hh, int W, int Hh, int S, int firstPass, int nPasses, uint64_t seed, int depthLimit,
                      int x0, int y0, int w, int hgt, float* accum, int nThreads, uint64_t* counters)
{
    Handle* H = static_cast<Handle*>(hh);
    const Oracle& o = *H->o;
    Oracle::FrameConsts c = Oracle::frameConsts(W, Hh, S);
    if (nThreads < 1)
        nThreads = 1;
    std::vector<Counters> ctrs(nThreads, Counters{0, 0, 0, 0});
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int t = 0; t < nThreads; t++) {
        pool.emplace_back([&, t] {
            Counters* ctr = counters ? &ctrs[t] : nullptr;
            if (H->math)
                o.renderRows<StrictMath>(c, S, firstPass, nPasses, seed, depthLimit, x0, w, y0 + t, y0 + hgt,
                                         nThreads, accum, ctr);
            else
                o.renderRows<LibmMath>(c, S, firstPass, nPasses, seed, depthLimit, x0, w, y0 + t, y0 + hgt,
                                       nThreads, accum, ctr);
        });
    }
    for (auto& th : pool)
        th.join();
    auto t1 = std::chrono::steady_clock::now();
    if (counters) {
        counters[0] = counters[1] = counters[2] = counters[3] = 0;
        for (const Counters& k : ctrs) {
            counters[0] += k.paths;
            counters[1] += k.traversals;
            counters[2] += k.vertices;
            counters[3] += k.primitiveTests;
        }
    }
    return std::chrono::duration<double>(t1 - t0).count();
}

// Renderer.cpp:73-75 + Image.cpp:14-27
void koracle_resolve(int math, int n, const float* accum, int pass, uint32_t* pixels)
{
    for (int i = 0; i < n; i++)
        pixels[i] = math ? resolvePixel<StrictMath>(accum + 4 * i, pass) : resolvePixel<LibmMath>(accum + 4 * i, pass);
}

// The reference's own stream discipline and threading, for CPU-baseline timing when
// oracle/_ref is not available: ONE serial stream per row slice seeded 0715517 * (y0 + 1)
// (Renderer.cpp:27), slices of (H + 1) / nThreads rows, one thread each
// (cpu/Scheduler.cpp:32-42; the last slice is clamped to the image), S = 32 (Renderer.cpp:21).
// Returns wall seconds for `passes` passes; accum receives the float sums.
double koracle_render_native(void* hh, int W, int Hh, int passes, int nThreads, int depthLimit, float* accum)
{
    Handle* H = static_cast<Handle*>(hh);
    H->o->timingRetrace = true;
    const Oracle& o = *H->o;
    const int S = 32;
    Oracle::FrameConsts c = Oracle::frameConsts(W, Hh, S);
    int slice = (Hh + 1) / std::max(1, nThreads);
    if (slice < 1)
        slice = 1;
    auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> pool;
    for (int ys = 0; ys < Hh; ys += slice) {
        pool.emplace_back([&, ys] {
            Rng rng;
            rng.setSeed(0715517u * (unsigned)(ys + 1));
            int ye = std::min(ys + slice, Hh);
            for (int pass = 1; pass <= passes; pass++)
                for (int y = ys; y < ye; y++)
                    for (int x = 0; x < W; x++) {
                        V3 radiance = v3(0, 0, 0);
                        for (int sy = 0; sy < c.n; sy++)
                            for (int sx = 0; sx < c.n; sx++)
                                radiance = radiance + (H->math ? o.cameraPath<StrictMath>(c, x, y, sx, sy, rng,
                                                                                          depthLimit, nullptr)
                                                               : o.cameraPath<LibmMath>(c, x, y, sx, sy, rng,
                                                                                        depthLimit, nullptr));
                        V3 r = radiance / (float)S;
                        float* dst = accum + 4 * ((size_t)y * W + x);
                        dst[0] += r.x;
                        dst[1] += r.y;
                        dst[2] += r.z;
                    }
        });
    }
    for (auto& th : pool)
        th.join();
    auto t1 = std::chrono::steady_clock::now();
    H->o->timingRetrace = false;
    return std::chrono::duration<double>(t1 - t0).count();
}

// include/kajo_strictmath.h, element-wise, for tests/test_strictmath.py.
// fn: 0 sin, 1 cos, 2 asin, 3 acos, 4 pow(x, y)
void koracle_strictmath(int fn, int n, const float* x, const float* y, float* out)
{
    for (int i = 0; i < n; i++) {
        switch (fn) {
        case 0: out[i] = kajo_sinf(x[i]); break;
        case 1: out[i] = kajo_cosf(x[i]); break;
        case 2: out[i] = kajo_asinf(x[i]); break;
        case 3: out[i] = kajo_acosf(x[i]); break;
        default: out[i] = kajo_powf(x[i], y[i]); break;
        }
    }
}

// Camera ray of ONE path (Renderer.cpp:51-64 under the stream protocol) and the generator state after its
// jitter draw: what kajo_hip_kat_shade / koracle_shade take, so that single paths of a frame can be replayed.
int koracle_camera_ray(void* hh, int W, int Hh, int S, int pass, uint64_t seed, int x, int y, int sample, float* ray6,
                       uint64_t* state2)
{
    Handle* H = static_cast<Handle*>(hh);
    const Oracle& o = *H->o;
    Oracle::FrameConsts c = Oracle::frameConsts(W, Hh, S);
    Rng rng;
    uint64_t st[2];
    kajo_stream_state(seed, (uint32_t)pass, (uint32_t)sample, (uint32_t)(y * W + x), st);
    rng.lo = st[0];
    rng.hi = st[1];
    float g[4];
    rng.generate(g);
    const int sampleX = sample % c.n, sampleY = sample / c.n;
    float offX = g[0] * .5f + .5f;
    float offY = g[1] * .5f + .5f;
    float sx = x * c.pixelWidth + sampleX * c.sampleWidth + offX * c.sampleWidth;
    float sy = (c.H - y) * c.pixelHeight + sampleY * c.sampleHeight + offY * c.sampleHeight;
    V3 direction = normalize(o.p1 + (o.p2 - o.p1) * sx + (o.p3 - o.p1) * sy - o.origin);
    st3(ray6, 0, o.origin);
    st3(ray6, 1, direction);
    state2[0] = rng.lo;
    state2[1] = rng.hi;
    return 0;
}

// Event log of ONE camera path (pixel x,y; sample index; pass): records of 4 floats
// (code, a, b, c): 1 = vertex (id, depth, t), 2 = lobe (kind, s), 3 = light sample kept (pb, pl, Ld.x),
// 4 = BSDF extension (p, pL, wb.x), 5 / 6 = vertex position / normal, 7 = light sample direction, 8 / 9 = origin /
// direction of the next segment. Returns the number of floats written (<= cap); rgb gets the path radiance.
int koracle_debug_path(void* hh, int W, int Hh, int S, int pass, uint64_t seed, int depthLimit, int x, int y,
                       int sample, float* out, int cap, float* rgb)
{
    Handle* H = static_cast<Handle*>(hh);
    const Oracle& o = *H->o;
    Oracle::FrameConsts c = Oracle::frameConsts(W, Hh, S);
    std::vector<float> log;
    g_pathLog = &log;
    Rng rng;
    uint64_t st[2];
    kajo_stream_state(seed, (uint32_t)pass, (uint32_t)sample, (uint32_t)(y * W + x), st);
    rng.lo = st[0];
    rng.hi = st[1];
    V3 r = H->math ? o.cameraPath<StrictMath>(c, x, y, sample % c.n, sample / c.n, rng, depthLimit, nullptr)
                   : o.cameraPath<LibmMath>(c, x, y, sample % c.n, sample / c.n, rng, depthLimit, nullptr);
    g_pathLog = nullptr;
    st3(rgb, 0, r);
    int n = (int)std::min<size_t>(log.size(), (size_t)cap);
    std::memcpy(out, log.data(), n * sizeof(float));
    return n;
}

} // extern "C"
