// stage.cpp -- host-side scene staging for the HIP backend.
//
// Does once per scene what cpu::Scene's constructor does (renderer/cpu/Scene.cpp:9-38:
// invMatrix = glm::inverse(matrix), determinant = glm::determinant(matrix)) plus the camera
// basis of renderer/cpu/Renderer.cpp:29-34 (three glm::unProject calls and inverse(view) *
// (0,0,0,1)), and lays the result out as device_scene.h describes.
//
// The arithmetic follows glm 0.9.3.4's formulas in their operand order
// (third_party/glm/glm/core/func_matrix.inl:446-580, gtc/matrix_transform.inl:337-356) in IEEE
// binary32; this file must be compiled with FP contraction off (-ffp-contract=off), because the
// STRICT kernels are compared bit for bit with a CPU evaluation of the same expressions.
#include "stage.h"
#include "tuning.h"

#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>

namespace kajo
{

namespace
{

// column-major 4x4, e(c, r) = element in column c, row r (glm's m[c][r])
struct Mat4
{
    float v[16];
    float e(int c, int r) const { return v[4 * c + r]; }
    float& e(int c, int r) { return v[4 * c + r]; }
};

Mat4 load(const float* p)
{
    Mat4 m;
    std::memcpy(m.v, p, sizeof m.v);
    return m;
}

float det2(float a, float b, float c, float d) // a*b - c*d
{
    return a * b - c * d;
}

// cofactor expansion along the first column pair, func_matrix.inl:446-470
float determinant(const Mat4& m)
{
    const float f00 = det2(m.e(2, 2), m.e(3, 3), m.e(3, 2), m.e(2, 3));
    const float f01 = det2(m.e(2, 1), m.e(3, 3), m.e(3, 1), m.e(2, 3));
    const float f02 = det2(m.e(2, 1), m.e(3, 2), m.e(3, 1), m.e(2, 2));
    const float f03 = det2(m.e(2, 0), m.e(3, 3), m.e(3, 0), m.e(2, 3));
    const float f04 = det2(m.e(2, 0), m.e(3, 2), m.e(3, 0), m.e(2, 2));
    const float f05 = det2(m.e(2, 0), m.e(3, 1), m.e(3, 0), m.e(2, 1));
    const float k0 = +(m.e(1, 1) * f00 - m.e(1, 2) * f01 + m.e(1, 3) * f02);
    const float k1 = -(m.e(1, 0) * f00 - m.e(1, 2) * f03 + m.e(1, 3) * f04);
    const float k2 = +(m.e(1, 0) * f01 - m.e(1, 1) * f03 + m.e(1, 3) * f05);
    const float k3 = -(m.e(1, 0) * f02 - m.e(1, 1) * f04 + m.e(1, 2) * f05);
    return m.e(0, 0) * k0 + m.e(0, 1) * k1 + m.e(0, 2) * k2 + m.e(0, 3) * k3;
}

// func_matrix.inl:523-580. The 18 two-by-two minors are indexed [pair of rows][pair of cols].
Mat4 inverse(const Mat4& m)
{
    // minors of rows (2,3), (1,3), (1,2) over column pairs
    float A[3], B[3], Cc[3], D[3], E[3], F[3];
    const int ra[3] = {2, 1, 1}, rb[3] = {3, 3, 2};
    for (int k = 0; k < 3; k++) {
        const int p = ra[k], q = rb[k];
        A[k] = det2(m.e(p, 2), m.e(q, 3), m.e(q, 2), m.e(p, 3));
        B[k] = det2(m.e(p, 1), m.e(q, 3), m.e(q, 1), m.e(p, 3));
        Cc[k] = det2(m.e(p, 1), m.e(q, 2), m.e(q, 1), m.e(p, 2));
        D[k] = det2(m.e(p, 0), m.e(q, 3), m.e(q, 0), m.e(p, 3));
        E[k] = det2(m.e(p, 0), m.e(q, 2), m.e(q, 0), m.e(p, 2));
        F[k] = det2(m.e(p, 0), m.e(q, 1), m.e(q, 0), m.e(p, 1));
    }
    Mat4 r;
    for (int i = 0; i < 4; i++) {
        const int k = i < 2 ? 0 : i - 1;    // which row pair feeds result row i
        const int c = i == 0 ? 1 : 0;       // source column of m for the weights
        const float w0 = m.e(c, 0), w1 = m.e(c, 1), w2 = m.e(c, 2), w3 = m.e(c, 3);
        const float sa = (i & 1) ? -1.f : +1.f;
        const float sb = -sa;
        r.e(0, i) = sa * (w1 * A[k] - w2 * B[k] + w3 * Cc[k]);
        r.e(1, i) = sb * (w0 * A[k] - w2 * D[k] + w3 * E[k]);
        r.e(2, i) = sa * (w0 * B[k] - w1 * D[k] + w3 * F[k]);
        r.e(3, i) = sb * (w0 * Cc[k] - w1 * E[k] + w2 * F[k]);
    }
    const float det = m.e(0, 0) * r.e(0, 0) + m.e(0, 1) * r.e(1, 0) + m.e(0, 2) * r.e(2, 0) + m.e(0, 3) * r.e(3, 0);
    for (float& x : r.v)
        x = x / det;
    return r;
}

// type_mat4x4.inl:757-779
Mat4 multiply(const Mat4& a, const Mat4& b)
{
    Mat4 r;
    for (int c = 0; c < 4; c++)
        for (int i = 0; i < 4; i++)
            r.e(c, i) = a.e(0, i) * b.e(c, 0) + a.e(1, i) * b.e(c, 1) + a.e(2, i) * b.e(c, 2) + a.e(3, i) * b.e(c, 3);
    return r;
}

// type_mat4x4.inl:689-700
void transform(const Mat4& a, const float in[4], float out[4])
{
    for (int i = 0; i < 4; i++)
        out[i] = a.e(0, i) * in[0] + a.e(1, i) * in[1] + a.e(2, i) * in[2] + a.e(3, i) * in[3];
}

// gtc/matrix_transform.inl:337-356 with viewport (0, 0, 1, 1)
void unProject(const Mat4& invPV, float wx, float wy, float wz, float out[3])
{
    float t[4] = {wx, wy, wz, 1.f};
    t[0] = (t[0] - 0.f) / 1.f;
    t[1] = (t[1] - 0.f) / 1.f;
    for (float& c : t)
        c = c * 2.f - 1.f;
    float o[4];
    transform(invPV, t, o);
    for (int i = 0; i < 3; i++)
        out[i] = o[i] / o[3];
}

bool isPureTranslation(const Mat4& m)
{
    for (int c = 0; c < 3; c++)
        for (int r = 0; r < 4; r++)
            if (m.e(c, r) != (c == r ? 1.f : 0.f))
                return false;
    return m.e(3, 3) == 1.f;
}

float max2(float a, float b) { return a < b ? b : a; } // glm::max / std::max

DMaterial stageMaterial(const KajoMaterial& k)
{
    DMaterial d;
    std::memset(&d, 0, sizeof d);
    for (int i = 0; i < 3; i++) {
        d.diffuse[i] = k.diffuse[i];
        d.specular[i] = k.specular[i];
        d.emission[i] = k.emission[i];
        d.transparency[i] = k.transparency[i];
    }
    // Shader.cpp:124-125 + Random.cpp:104-109
    float mx[3];
    for (int i = 0; i < 3; i++)
        mx[i] = max2(max2(k.diffuse[i], k.specular[i]), k.transparency[i]);
    d.pRR = max2(mx[0], max2(mx[1], mx[2]));
    // Shader.cpp:130-133,153
    const float sd = k.diffuse[0] + k.diffuse[1] + k.diffuse[2];
    const float ss = k.specular[0] + k.specular[1] + k.specular[2];
    const float st = k.transparency[0] + k.transparency[1] + k.transparency[2];
    d.pT = st / (sd + ss + st);
    d.pD = sd / (sd + ss);
    d.exponent = k.specularExponent;
    d.ior = k.refractiveIndex;
    d.flags = (!(k.emission[0] == 0 && k.emission[1] == 0 && k.emission[2] == 0 && k.emission[3] == 0) ? KAJO_MAT_IS_LIGHT : 0u) |
              (k.specularExponent != 0.0f ? KAJO_MAT_HAS_EXPONENT : 0u);
    d.sTransparent = 1.f / d.pRR * 1.f / d.pT;
    d.sDiffuse = 1.f / d.pRR * 1.f / (1.f - d.pT) * 1.f / d.pD;
    d.sSpecular = 1.f / d.pRR * 1.f / (1.f - d.pT) * 1.f / (1.f - d.pD);
    d.sStop = 1.f / (1.f - d.pRR);
    d.sDepth = 1.f / d.pRR;
    return d;
}

// World-space bounding box of sphere i: the unit-radius-r sphere under M is an ellipsoid whose
// half-extent along world axis k is r * |row k of mat3(M)|.
// `E`: what binary32 rounding can add to r^2 - (distance of the ray's line from the centre)^2, the quantity whose sign decides
// "hit" in Raytracer.cpp:26-30, for ray origins up to the distance the caller sized it for (floatHitSlack below): the reference
// reports a hit on a sphere the exact line misses by up to sqrt(r^2 + E) - r, and a culling structure must not hide that sphere.
void sphereBounds(const KajoSphere& sp, double E, float lo[3], float hi[3])
{
    const Mat4 M = load(sp.transform);
    for (int k = 0; k < 3; k++) {
        const double e0 = (double)sp.radius * std::sqrt((double)M.e(0, k) * M.e(0, k) + (double)M.e(1, k) * M.e(1, k) +
                                                         (double)M.e(2, k) * M.e(2, k));
        const double e = std::sqrt(e0 * e0 + E);
        const double c = M.e(3, k);
        const double pad = 1e-4 * (e + std::fabs(c)) + 1e-5; // registration margin >> float rounding of the DDA
        lo[k] = (float)(c - e - pad);
        hi[k] = (float)(c + e + pad);
    }
}

// Rounding of the discriminant b^2 - 4ac of Raytracer.cpp:26-30 in units of (4a): with o = O - c, |d| = 1 the terms are b^2 ~
// 4 (d.o)^2 and 4ac ~ 4 (|o|^2 - r^2), each a sum of three products rounded to binary32 (relative 2^-24 per operation, ~1e-6 |o|^2
// in all after the squaring of b) -- 8e-7 |o|^2 per unit of 4a, doubled for safety. `reach`: the largest |O - c| it must cover.
double floatHitSlack(double reach)
{
    return 1.6e-6 * reach * reach;
}

// Where can a ray of a path START? At the camera, on a sphere, or on a plane -- and a point of a plane can be arbitrarily far away
// unless the planes close the scene in. The convex region the planes leave around the camera (every plane taken on the camera's
// side, as an unbounded two-sided plane is seen from there): when it is BOUNDED -- a room, as in every scene of the reference's
// data/ -- and no plane lets light through (Shader.cpp:130-151), each ray of each path starts inside it (the first plane a ray from
// inside meets is on its boundary, and an opaque plane sends the path back in), and its bounding box, widened to hold every sphere,
// bounds |O - c| for the culling structures below. Vertices by intersecting plane triples; bounded iff it has a vertex (is
// pointed) and no direction n_i x n_j stays inside all half-spaces.
void findRoom(const KajoScene& s, StagedScene& out)
{
    out.roomClosed = false;
    const int np = s.nPlanes;
    if (np < 4 || np > 64)
        return;
    for (int i = 0; i < np; i++) {
        const float* tr = s.planes[i].material.transparency;
        if (tr[0] != 0.f || tr[1] != 0.f || tr[2] != 0.f || !(tr[0] == tr[0] && tr[1] == tr[1] && tr[2] == tr[2]))
            return; // a path can pass through this plane and go on outside
    }
    double cam[3];
    {
        const Mat4 view = load(s.camera.transform);
        const float zero[4] = {0.f, 0.f, 0.f, 1.f};
        float o[4];
        transform(inverse(view), zero, o);
        for (int k = 0; k < 3; k++)
            cam[k] = o[k];
    }
    std::vector<double> n(4 * (size_t)np); // s_i * (row y of the inverse): g_i(P) = n.P + n[3] > 0 on the camera's side
    for (int i = 0; i < np; i++) {
        const DFloat4 r = out.planeRow[i];
        double g = (double)r.x * cam[0] + (double)r.y * cam[1] + (double)r.z * cam[2] + r.w;
        const double len = std::sqrt((double)r.x * r.x + (double)r.y * r.y + (double)r.z * r.z);
        if (!(len > 1e-12) || !(std::fabs(g) > 1e-6 * len))
            return; // a degenerate plane, or the camera on a plane
        const double sg = (g > 0 ? 1.0 : -1.0) / len;
        n[4 * i] = sg * r.x, n[4 * i + 1] = sg * r.y, n[4 * i + 2] = sg * r.z, n[4 * i + 3] = sg * r.w;
    }
    auto inside = [&](const double P[3], double tol) {
        for (int m = 0; m < np; m++)
            if (n[4 * m] * P[0] + n[4 * m + 1] * P[1] + n[4 * m + 2] * P[2] + n[4 * m + 3] < -tol)
                return false;
        return true;
    };
    double lo[3] = {1e300, 1e300, 1e300}, hi[3] = {-1e300, -1e300, -1e300}, scale = 1.0;
    for (int k = 0; k < 3; k++)
        scale = std::fmax(scale, std::fabs(cam[k]));
    int vertices = 0;
    for (int i = 0; i < np; i++)
        for (int j = i + 1; j < np; j++) {
            const double* a = &n[4 * i];
            const double* b = &n[4 * j];
            const double u[3] = {a[1] * b[2] - a[2] * b[1], a[2] * b[0] - a[0] * b[2], a[0] * b[1] - a[1] * b[0]};
            const double ul = std::sqrt(u[0] * u[0] + u[1] * u[1] + u[2] * u[2]);
            if (ul > 1e-9) { // an edge direction: the region is unbounded if it, or its opposite, leaves through no plane
                for (int sgn = -1; sgn <= 1; sgn += 2) {
                    bool escapes = true;
                    for (int m = 0; m < np && escapes; m++)
                        escapes = sgn * (n[4 * m] * u[0] + n[4 * m + 1] * u[1] + n[4 * m + 2] * u[2]) >= -1e-9 * ul;
                    if (escapes)
                        return;
                }
            }
            for (int k = j + 1; k < np; k++) {
                const double* c = &n[4 * k];
                const double det = c[0] * u[0] + c[1] * u[1] + c[2] * u[2];
                if (!(std::fabs(det) > 1e-9))
                    continue;
                // x = -(a3 (b x c) + b3 (c x a) + c3 (a x b)) / det
                const double bc[3] = {b[1] * c[2] - b[2] * c[1], b[2] * c[0] - b[0] * c[2], b[0] * c[1] - b[1] * c[0]};
                const double ca[3] = {c[1] * a[2] - c[2] * a[1], c[2] * a[0] - c[0] * a[2], c[0] * a[1] - c[1] * a[0]};
                double P[3];
                for (int q = 0; q < 3; q++)
                    P[q] = -(a[3] * bc[q] + b[3] * ca[q] + c[3] * u[q]) / det;
                const double mag = std::fmax(scale, std::fmax(std::fabs(P[0]), std::fmax(std::fabs(P[1]), std::fabs(P[2]))));
                if (!inside(P, 1e-7 * mag))
                    continue;
                vertices++;
                for (int q = 0; q < 3; q++) {
                    lo[q] = std::fmin(lo[q], P[q]);
                    hi[q] = std::fmax(hi[q], P[q]);
                }
            }
        }
    if (vertices < 4)
        return;
    // A region that closes only far away -- two walls a rounding error from parallel meet 1e7 units out -- bounds nothing usefully:
    // margins sized for it would put every sphere into every cell. Such a scene is treated as open (rays from far out walk every
    // sphere); "far" = beyond 64 times the extent of the spheres and the camera.
    {
        double slo[3] = {cam[0], cam[1], cam[2]}, shi[3] = {cam[0], cam[1], cam[2]}, sd2 = 0, rd2 = 0;
        for (int i = 0; i < s.nSpheres; i++) {
            float bl[3], bh[3];
            sphereBounds(s.spheres[i], 0.0, bl, bh);
            for (int k = 0; k < 3; k++) {
                slo[k] = std::fmin(slo[k], (double)bl[k]);
                shi[k] = std::fmax(shi[k], (double)bh[k]);
            }
        }
        for (int k = 0; k < 3; k++) {
            sd2 += (shi[k] - slo[k]) * (shi[k] - slo[k]);
            rd2 += (hi[k] - lo[k]) * (hi[k] - lo[k]);
        }
        if (!(rd2 <= 64.0 * 64.0 * std::fmax(sd2, 1e-6)))
            return;
    }
    // (a sphere that a plane cuts, or that lies beyond one, is seen from inside all the same: the box holds them all)
    for (int i = 0; i < s.nSpheres; i++) {
        float bl[3], bh[3];
        sphereBounds(s.spheres[i], 0.0, bl, bh);
        for (int k = 0; k < 3; k++) {
            lo[k] = std::fmin(lo[k], (double)bl[k]);
            hi[k] = std::fmax(hi[k], (double)bh[k]);
        }
    }
    for (int k = 0; k < 3; k++) {
        if (!(lo[k] > -1e30 && hi[k] < 1e30))
            return;
        const double pad = 1e-6 * (std::fabs(lo[k]) + std::fabs(hi[k])) + 1e-6;
        out.roomLo[k] = lo[k] - pad;
        out.roomHi[k] = hi[k] + pad;
    }
    out.roomClosed = true;
}

void buildGrid(const KajoScene& s, StagedScene& out, int gridMinSpheres)
{
    out.gridEnabled = false;
    const int n = s.nSpheres;
    if (n < gridMinSpheres || gridMinSpheres <= 0 || n > 65535) // (cell lists hold 16-bit sphere indices)
        return;
    // The walk compares the reported distances t * determinant (Raytracer.cpp:71,97) with WORLD-space cell
    // boundaries, which is only sound when the two agree: every sphere determinant exactly 1 (t itself is the world-space
    // ray parameter under any affine transform) and every plane rigid to within float rounding (far inside the
    // registration margin below). A scaled sphere (det != 1) anywhere keeps the every-sphere walk.
    if (!out.planesRigid)
        return;
    for (int i = 0; i < n; i++)
        if (out.invDet[17 * ((size_t)s.nPlanes + i) + 16] != 1.f)
            return;
    // How far from the spheres a ray may start and still be answered through the grid: the registration margins are sized for it
    // (floatHitSlack), and the walk sends a ray that starts farther out -- a vertex far away on an open floor -- to the every-sphere
    // loop instead (integrator.inc.hip trace(): `far`). A closed room bounds every origin: nothing is ever sent there.
    double center[3], half2 = 0;
    {
        double blo[3] = {1e300, 1e300, 1e300}, bhi[3] = {-1e300, -1e300, -1e300};
        for (int i = 0; i < n; i++) {
            float l[3], h[3];
            sphereBounds(s.spheres[i], 0.0, l, h);
            for (int k = 0; k < 3; k++) {
                if (!(l[k] > -3e37f && h[k] < 3e37f))
                    return; // non-finite geometry: keep the brute-force walk
                blo[k] = std::fmin(blo[k], (double)l[k]);
                bhi[k] = std::fmax(bhi[k], (double)h[k]);
            }
        }
        for (int k = 0; k < 3; k++) {
            center[k] = 0.5 * (blo[k] + bhi[k]);
            half2 += 0.25 * (bhi[k] - blo[k]) * (bhi[k] - blo[k]);
        }
    }
    double reach = 4.0 * std::sqrt(half2) + 1.0; // an open scene: origins up to twice the spheres' diameter from their centre
    if (out.roomClosed) {
        reach = 0;
        for (int c = 0; c < 8; c++) {
            double d2 = 0;
            for (int k = 0; k < 3; k++) {
                const double v = ((c >> k) & 1 ? out.roomHi[k] : out.roomLo[k]) - center[k];
                d2 += v * v;
            }
            reach = std::fmax(reach, std::sqrt(d2));
        }
    }
    // (|O - c_i| <= |O - centre| + |centre - c_i| <= reach + half diagonal)
    const double E = floatHitSlack(reach * 1.0001 + std::sqrt(half2));
    for (int k = 0; k < 3; k++)
        out.gridCenter[k] = (float)center[k];
    out.gridReach2 = out.roomClosed ? 3e38f : (float)(reach * reach);
    std::vector<float> lo(3 * (size_t)n), hi(3 * (size_t)n);
    float bmin[3] = {3e38f, 3e38f, 3e38f}, bmax[3] = {-3e38f, -3e38f, -3e38f};
    for (int i = 0; i < n; i++) {
        sphereBounds(s.spheres[i], E, &lo[3 * i], &hi[3 * i]);
        for (int k = 0; k < 3; k++) {
            if (!(lo[3 * i + k] > -3e37f && hi[3 * i + k] < 3e37f))
                return; // non-finite geometry: keep the brute-force walk
            bmin[k] = std::fmin(bmin[k], lo[3 * i + k]);
            bmax[k] = std::fmax(bmax[k], hi[3 * i + k]);
        }
    }
    // about four cells per sphere, cells as cubic as the bounds allow, at most 128 per axis
    double ext[3], vol = 1;
    for (int k = 0; k < 3; k++) {
        ext[k] = std::fmax((double)bmax[k] - bmin[k], 1e-3);
        vol *= ext[k];
    }
    double cellsPerSphere = 1.0; // measured on the 1000-sphere scene: 0.5 .. 2 within 10 %, finer grids lose to the cell stepping
    KAJO_TUNE_DOUBLE("KAJO_GRID_CELLS_PER_SPHERE", cellsPerSphere);
    const double side = std::cbrt(vol / (cellsPerSphere * n));
    size_t cells = 1;
    for (int k = 0; k < 3; k++) {
        int d = (int)std::ceil(ext[k] / side);
        d = d < 1 ? 1 : (d > 128 ? 128 : d);
        out.gridDim[k] = d;
        out.gridMin[k] = bmin[k];
        out.gridMax[k] = bmax[k];
        out.gridCell[k] = (float)(ext[k] / d);
        cells *= (size_t)d;
    }
    auto cellRange = [&](int i, int k, int& a, int& b) {
        a = (int)std::floor((lo[3 * i + k] - bmin[k]) / out.gridCell[k]);
        b = (int)std::floor((hi[3 * i + k] - bmin[k]) / out.gridCell[k]);
        a = a < 0 ? 0 : (a >= out.gridDim[k] ? out.gridDim[k] - 1 : a);
        b = b < 0 ? 0 : (b >= out.gridDim[k] ? out.gridDim[k] - 1 : b);
    };
    // A sphere is registered in the cells of its bounding box that it really reaches: for a sphere whose matrix is a pure
    // translation (a ball of radius r in world space) the corner cells of the box whose nearest point is farther than r + the
    // registration margin from the centre are left out (a quarter of the items on the 1000-sphere scene). Conservative: the
    // margin is the one of sphereBounds; other matrices (rotations, shears of determinant 1) keep their whole box.
    std::vector<unsigned char> ball(n, 0);
    for (int i = 0; i < n; i++) {
        const Mat4 M = load(s.spheres[i].transform);
        bool identity3 = true;
        for (int c = 0; c < 3; c++)
            for (int r = 0; r < 3; r++)
                identity3 = identity3 && M.e(c, r) == (c == r ? 1.f : 0.f);
        ball[i] = identity3 && !KAJO_TUNE_SET("KAJO_GRID_BOX_REGISTRATION"); // (tuning builds: register the whole box, as round 2 did)
    }
    auto reaches = [&](int i, int x, int y, int z) {
        if (!ball[i])
            return true;
        const Mat4 M = load(s.spheres[i].transform);
        const int cell[3] = {x, y, z};
        double d2 = 0, reach = 0;
        for (int k = 0; k < 3; k++) {
            const double c = M.e(3, k), lo_ = (double)bmin[k] + cell[k] * (double)out.gridCell[k], hi_ = lo_ + (double)out.gridCell[k];
            const double d = c < lo_ ? lo_ - c : (c > hi_ ? c - hi_ : 0.0);
            d2 += d * d;
            reach = std::fmax(reach, (double)hi[3 * i + k] - c); // r + pad of sphereBounds
        }
        return d2 <= reach * reach * (1 + 1e-6) + 1e-12;
    };
    std::vector<uint32_t> count(cells + 1, 0);
    for (int pass = 0; pass < 2; pass++) {
        for (int i = 0; i < n; i++) { // ascending sphere index within every cell
            int x0, x1, y0, y1, z0, z1;
            cellRange(i, 0, x0, x1);
            cellRange(i, 1, y0, y1);
            cellRange(i, 2, z0, z1);
            for (int z = z0; z <= z1; z++)
                for (int y = y0; y <= y1; y++)
                    for (int x = x0; x <= x1; x++) {
                        if (!reaches(i, x, y, z))
                            continue;
                        const size_t c = ((size_t)z * out.gridDim[1] + y) * out.gridDim[0] + x;
                        if (pass == 0)
                            count[c + 1]++;
                        else
                            out.gridItems[count[c]++] = (uint16_t)i;
                    }
        }
        if (pass == 0) {
            for (size_t c = 0; c < cells; c++)
                count[c + 1] += count[c];
            out.gridCellStart = count;
            out.gridItems.assign(count[cells], 0);
        }
    }
    out.gridEnabled = true;
}

// Per-light visibility lists (device_scene.h DShadowLists). Conservative by construction: a sphere is left out of a bin only
// if NO ray from the light's centre C through the bin comes within R_i = sqrt(radius_i^2 + E) + r_light + pad of its centre,
// where E covers what binary32 rounding can add to the discriminant b^2 - 4ac of Raytracer.cpp:26-30 at this scene's
// distances (a sphere the exact line misses by less than sqrt(radius^2 + E) - radius may still be "hit" in float arithmetic,
// and the brute-force walk would see that hit), and pad the rounding of the binning itself.
void buildShadowLists(const KajoScene& s, StagedScene& out, bool wanted)
{
    out.shadowEnabled = false;
    const int n = s.nSpheres, nL = (int)out.light.size();
    // (the lists answer a query from candidates chosen by geometry alone: they need a bound on where a shadow ray can start -- a
    // closed room, findRoom -- to know how near a miss the reference's arithmetic can still report as a hit. Open scenes keep the
    // grid walk for their shadow rays, which sends far origins to the every-sphere loop ray by ray.)
    if (!wanted || !out.gridEnabled || !out.allTranslated || nL == 0 || !out.roomClosed)
        return;
    // bins per cube-face axis. 1000 spheres / 16 lights: 16 -> 15.1 candidate spheres per query, 0.9 MB of lists, 1.64 G paths/s;
    // 32 -> 7.8, 2.2 MB, 2.35 G; 64 -> 5.1, 6.8 MB, 2.74 G; 96 and 128 -> 2.3 G again (the lists fall out of the L2). The tube of
    // radius r_light around the segment alone holds ~4 spheres. Scenes of very many lights get coarser bins (4 M bins in all).
    int N = 64;
    while (N > 8 && (size_t)nL * 6 * N * N > ((size_t)1 << 22))
        N /= 2;
    KAJO_TUNE_INT("KAJO_SHADOW_BINS", 2, 128, N);
    const size_t binsPerLight = (size_t)6 * N * N;
    if ((size_t)nL * binsPerLight > ((size_t)1 << 23))
        return;
    // every shadow ray starts inside the room, every sphere lies inside it: |O - c| is at most the room's diagonal
    // (rounds 1-4 took the extent of the spheres and the camera here, which a vertex far out on a plane exceeds: advisor finding)
    double diag2 = 0;
    for (int k = 0; k < 3; k++)
        diag2 += (out.roomHi[k] - out.roomLo[k]) * (out.roomHi[k] - out.roomLo[k]);
    const double E = floatHitSlack(std::sqrt(diag2));
    // bins: centre direction and bounding half-angle (as cosine / sine pairs are not needed: angles are compared directly)
    struct Bin
    {
        double q[3], beta;
    };
    std::vector<Bin> bins(binsPerLight);
    auto dirOf = [&](int f, double ua, double ub, double d[3]) {
        const int m = f >> 1, a = (m + 1) % 3, b = (m + 2) % 3;
        d[m] = (f & 1) ? -1.0 : 1.0;
        d[a] = ua;
        d[b] = ub;
        const double l = std::sqrt(d[0] * d[0] + d[1] * d[1] + d[2] * d[2]);
        d[0] /= l, d[1] /= l, d[2] /= l;
    };
    auto angle = [](const double x[3], const double y[3]) {
        const double c = x[0] * y[0] + x[1] * y[1] + x[2] * y[2];
        return std::acos(c > 1 ? 1 : (c < -1 ? -1 : c));
    };
    for (int f = 0; f < 6; f++)
        for (int ib = 0; ib < N; ib++)
            for (int ia = 0; ia < N; ia++) {
                Bin& B = bins[((size_t)f * N + ib) * N + ia];
                dirOf(f, (ia + .5) * 2.0 / N - 1, (ib + .5) * 2.0 / N - 1, B.q);
                B.beta = 0;
                for (int c = 0; c < 4; c++) {
                    double d[3];
                    dirOf(f, (ia + (c & 1)) * 2.0 / N - 1, (ib + (c >> 1)) * 2.0 / N - 1, d);
                    B.beta = std::fmax(B.beta, angle(B.q, d));
                }
            }
    // blocks of 8 x 8 bins with a bounding cone of their own: a sphere's cone of directions is tested against the block first
    const int BS = 8, NB = (N + BS - 1) / BS;
    std::vector<Bin> blocks((size_t)6 * NB * NB);
    for (int f = 0; f < 6; f++)
        for (int jb = 0; jb < NB; jb++)
            for (int ja = 0; ja < NB; ja++) {
                Bin& B = blocks[((size_t)f * NB + jb) * NB + ja];
                const int a0 = ja * BS, a1 = std::min(N, a0 + BS), b0 = jb * BS, b1 = std::min(N, b0 + BS);
                dirOf(f, (a0 + a1) * 1.0 / N - 1, (b0 + b1) * 1.0 / N - 1, B.q);
                B.beta = 0;
                for (int c = 0; c < 4; c++) {
                    double d[3];
                    dirOf(f, ((c & 1) ? a1 : a0) * 2.0 / N - 1, ((c >> 1) ? b1 : b0) * 2.0 / N - 1, d);
                    B.beta = std::fmax(B.beta, angle(B.q, d));
                }
            }
    const double kPiHalf = 1.5707963267948966, faceCone = 0.9553166181245093 /* acos(1 / sqrt 3) */, angPad = 1e-4;
    std::vector<std::vector<DShadowItem>> lists((size_t)nL * binsPerLight);
    for (int L = 0; L < nL; L++) {
        const int sl = out.light[L];
        const Mat4 ML = load(s.spheres[sl].transform);
        const double C[3] = {ML.e(3, 0), ML.e(3, 1), ML.e(3, 2)}, rL = s.spheres[sl].radius;
        for (int i = 0; i < n; i++) {
            if (i == sl)
                continue; // the light itself is always tested, first
            const Mat4 M = load(s.spheres[i].transform);
            const double w[3] = {M.e(3, 0) - C[0], M.e(3, 1) - C[1], M.e(3, 2) - C[2]}, rho = s.spheres[i].radius;
            const double D = std::sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
            const double mag = std::fabs(C[0]) + std::fabs(C[1]) + std::fabs(C[2]) + std::fabs(w[0]) + std::fabs(w[1]) + std::fabs(w[2]);
            const double R = std::sqrt(rho * rho + E) + rL + 2e-3 + 1e-5 * mag;
            const DShadowItem item{(float)((D - R) * (1 - 1e-6) - 1e-6), (uint32_t)i}; // (rounded down: a key never exceeds the true bound)
            const bool everywhere = D <= R;
            const double alpha = everywhere ? 4.0 : std::asin(R / D) + angPad;
            const double wd[3] = {everywhere ? 1.0 : w[0] / D, everywhere ? 0.0 : w[1] / D, everywhere ? 0.0 : w[2] / D};
            for (int f = 0; f < 6; f++) {
                if (!everywhere) {
                    double axis[3] = {0, 0, 0};
                    axis[f >> 1] = (f & 1) ? -1.0 : 1.0;
                    if (angle(wd, axis) > alpha + faceCone + angPad && alpha < kPiHalf)
                        continue; // the cone of directions toward the sphere misses this face
                }
                for (int jb = 0; jb < NB; jb++)
                    for (int ja = 0; ja < NB; ja++) {
                        const Bin& K = blocks[((size_t)f * NB + jb) * NB + ja];
                        if (!everywhere && angle(wd, K.q) > alpha + K.beta + angPad)
                            continue;
                        for (int ib = jb * BS; ib < std::min(N, jb * BS + BS); ib++)
                            for (int ia = ja * BS; ia < std::min(N, ja * BS + BS); ia++) {
                                const size_t b = ((size_t)f * N + ib) * N + ia;
                                if (everywhere || angle(wd, bins[b].q) <= alpha + bins[b].beta)
                                    lists[(size_t)L * binsPerLight + b].push_back(item);
                            }
                    }
            }
        }
    }
    size_t total = 0;
    for (auto& l : lists)
        total += l.size();
    if (total > ((size_t)1 << 25))
        return;
    out.shadowStart.assign(lists.size() + 1, 0);
    out.shadowItems.clear();
    out.shadowItems.reserve(total);
    for (size_t b = 0; b < lists.size(); b++) {
        auto& l = lists[b];
        std::stable_sort(l.begin(), l.end(), [](const DShadowItem& x, const DShadowItem& y) { return x.key < y.key; });
        out.shadowStart[b] = (uint32_t)out.shadowItems.size();
        out.shadowItems.insert(out.shadowItems.end(), l.begin(), l.end());
    }
    out.shadowStart[lists.size()] = (uint32_t)out.shadowItems.size();
    out.shadowN = N;
    // The kernels' form (device_scene.h DShadowLists): 16-bit keys in units of a per-light scale, rounded DOWN (a key only bounds from
    // below where a sphere can first be touched: a smaller key keeps the item in more walks, never drops it from one) and re-sorted
    // stably by the quantised key; 16-bit bin starts relative to a 32-bit base per row of bins.
    out.shadowInvKeyScale.assign(nL, 1.f);
    out.shadowPacked.assign(out.shadowItems.size(), 0u);
    out.shadowRowBase.assign((size_t)nL * 6 * N, 0u);
    out.shadowOff16.assign((size_t)nL * 6 * N * (N + 1), 0);
    for (int L = 0; L < nL; L++) {
        float maxKey = 0.f;
        for (size_t j = out.shadowStart[(size_t)L * binsPerLight]; j < out.shadowStart[(size_t)(L + 1) * binsPerLight]; j++)
            maxKey = std::fmax(maxKey, out.shadowItems[j].key);
        const double scale = std::fmax((double)maxKey, 1e-6) / 65535.0 * (1 + 1e-6);
        out.shadowInvKeyScale[L] = (float)(1.0 / scale);
        for (size_t row = (size_t)L * 6 * N; row < (size_t)(L + 1) * 6 * N; row++) {
            const size_t b0 = row * N;
            const uint32_t base = out.shadowStart[b0];
            if (out.shadowStart[b0 + N] - base > 65535u)
                return; // (a row of bins with more than 65 535 items: no lists for this scene)
            out.shadowRowBase[row] = base;
            for (int ia = 0; ia <= N; ia++)
                out.shadowOff16[row * (N + 1) + ia] = (uint16_t)(out.shadowStart[b0 + ia] - base);
            for (int ia = 0; ia < N; ia++) {
                const uint32_t a = out.shadowStart[b0 + ia], e = out.shadowStart[b0 + ia + 1];
                for (uint32_t j = a; j < e; j++) { // (ascending float keys quantise to ascending integers: the order stands)
                    const DShadowItem& it = out.shadowItems[j];
                    double q = std::floor((double)std::fmax(it.key, 0.f) / scale);
                    q = q > 65535.0 ? 65535.0 : q;
                    out.shadowPacked[j] = ((uint32_t)q << 16) | (it.index & 0xffffu);
                    out.shadowItems[j].key = (float)(q * scale * (1 - 1e-6)); // what the kernel's comparison amounts to (never above the true key)
                }
            }
        }
    }
    out.shadowEnabled = true;
}

} // namespace

void coordinateRange(const KajoScene& s, float* lo, float* hi)
{
    float mn = 0.f, mx = 0.f;
    bool bad = false;
    auto see = [&](float v) {
        const float a = std::fabs(v);
        if (!(a == a) || std::isinf(a)) {
            bad = true;
            return;
        }
        if (a == 0.f)
            return;
        mn = (mx == 0.f || a < mn) ? a : mn;
        mx = a > mx ? a : mx;
    };
    for (int i = 0; i < s.nPlanes; i++)
        for (float v : s.planes[i].transform)
            see(v);
    for (int i = 0; i < s.nSpheres; i++) {
        for (float v : s.spheres[i].transform)
            see(v);
        see(s.spheres[i].radius);
    }
    for (float v : s.camera.transform)
        see(v);
    *lo = mn;
    *hi = bad ? std::nanf("") : mx;
}

void stageScene(const KajoScene& s, StagedScene& out, int gridMinSpheres, bool shadowLists)
{
    out = StagedScene();
    out.nPlanes = s.nPlanes;
    out.nSpheres = s.nSpheres;
    for (int i = 0; i < 3; i++)
        out.background[i] = s.backgroundColor[i];

    for (int i = 0; i < s.nPlanes; i++) {
        const Mat4 M = load(s.planes[i].transform);
        const Mat4 inv = inverse(M);
        const float det = determinant(M);
        out.invDet.insert(out.invDet.end(), inv.v, inv.v + 16);
        out.invDet.push_back(det);
        out.planeRow.push_back(DFloat4{inv.e(0, 1), inv.e(1, 1), inv.e(2, 1), inv.e(3, 1)});
        out.planeDet.push_back(det);
        // FAST numerics drop the "* determinant" of Raytracer.cpp:97 when every plane is rigid to
        // within float rounding of a rotation matrix (|det - 1| <= 2^-20; spheres.json: 0.9999997)
        if (!(std::fabs(det - 1.f) <= 9.5367431640625e-7f))
            out.planesRigid = 0;
        // Raytracer.cpp:91-93: normal = mat3(M) * -(0,1,0); tangent = mat3(M) * (1,0,0); binormal = n x t
        float n[3], t[3];
        for (int r = 0; r < 3; r++) {
            n[r] = M.e(0, r) * -0.f + M.e(1, r) * -1.f + M.e(2, r) * -0.f;
            t[r] = M.e(0, r) * 1.f + M.e(1, r) * 0.f + M.e(2, r) * 0.f;
        }
        const float b[3] = {n[1] * t[2] - t[1] * n[2], n[2] * t[0] - t[2] * n[0], n[0] * t[1] - t[0] * n[1]};
        out.planeFrame.push_back(DFloat4{n[0], n[1], n[2], 0.f});
        out.planeFrame.push_back(DFloat4{t[0], t[1], t[2], 0.f});
        out.planeFrame.push_back(DFloat4{b[0], b[1], b[2], 0.f});
        out.material.push_back(stageMaterial(s.planes[i].material));
    }

    out.allTranslated = 1;
    for (int i = 0; i < s.nSpheres; i++) {
        const Mat4 M = load(s.spheres[i].transform);
        const Mat4 inv = inverse(M);
        const float det = determinant(M);
        out.invDet.insert(out.invDet.end(), inv.v, inv.v + 16);
        out.invDet.push_back(det);
        const float radius = s.spheres[i].radius;
        const float r2 = radius * radius;
        const float zero[4] = {0.f, 0.f, 0.f, 1.f};
        float c[4];
        transform(M, zero, c); // Light.cpp:37
        DSphereCold cold;
        std::memset(&cold, 0, sizeof cold);
        cold.cx = c[0];
        cold.cy = c[1];
        cold.cz = c[2];
        cold.radius = radius;
        for (int r = 0; r < 3; r++)
            for (int k = 0; k < 3; k++)
                cold.m[3 * r + k] = M.e(k, r);
        const bool translated = isPureTranslation(M) && det == 1.f && isPureTranslation(inv);
        cold.general = !translated;
        cold.invRadius = 1.f / radius;
        cold.invTwoPiR2 = 1.f / (6.28318530717958647692f * r2);
        uint32_t off = (uint32_t)out.sphereHot.size();
        if (translated) {
            // inverse is translate(-c) exactly: object-space origin = O + inv[3] = O - c
            out.sphereHot.push_back(DFloat4{inv.e(3, 0), inv.e(3, 1), inv.e(3, 2), r2});
        } else {
            out.allTranslated = 0;
            off |= KAJO_SPHERE_GENERAL;
            for (int r = 0; r < 3; r++)
                out.sphereHot.push_back(DFloat4{inv.e(0, r), inv.e(1, r), inv.e(2, r), inv.e(3, r)});
            out.sphereHot.push_back(DFloat4{r2, det, 0.f, 0.f});
        }
        out.sphereHotOffset.push_back(off);
        out.sphereCold.push_back(cold);
        DMaterial m = stageMaterial(s.spheres[i].material);
        out.material.push_back(m);
        if (m.flags & KAJO_MAT_IS_LIGHT)
            out.light.push_back(i);
    }

    findRoom(s, out);
    buildGrid(s, out, gridMinSpheres);
    buildShadowLists(s, out, shadowLists);

    // camera basis, Renderer.cpp:29-34
    const Mat4 view = load(s.camera.transform);
    const Mat4 proj = load(s.camera.projection);
    const Mat4 invPV = inverse(multiply(proj, view));
    float p1[3], p2[3], p3[3];
    unProject(invPV, 0.f, 0.f, 0.f, p1);
    unProject(invPV, 1.f, 0.f, 0.f, p2);
    unProject(invPV, 0.f, 1.f, 0.f, p3);
    const float zero[4] = {0.f, 0.f, 0.f, 1.f};
    float o[4];
    transform(inverse(view), zero, o);
    for (int i = 0; i < 3; i++) {
        out.p1[i] = p1[i];
        out.p2[i] = p2[i];
        out.p3[i] = p3[i];
        out.origin[i] = o[i];
    }
}

} // namespace kajo
