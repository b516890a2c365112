// Host-side value types of the registration interface, layout-compatible with what the reference passes across its
// SlamFunc boundary (source/common/testrunner.h:7-8) so that a caller holding reference types can memcpy / reinterpret:
//   Point_f  == Common::Point<float>  {x,y,z}, 12 bytes, no padding           (source/common/point.h:61-63)
//   Mat3     == glm::mat3             column-major, m[col][row]               (glm/detail/type_mat3x3.hpp)
//   Vec3     == glm::vec3
//   Mat4     == glm::mat4             column-major, translation in column 3   (common.cpp:353-358)
// Nothing here depends on glm or Eigen.
#pragma once
#include <cstddef>
#include <vector>

namespace Common {

struct Point_f {
    float x = 0.f, y = 0.f, z = 0.f;
};
static_assert(sizeof(Point_f) == 12, "Point_f must stay a packed 12-byte xyz triple");

struct Vec3 {
    float x = 0.f, y = 0.f, z = 0.f;
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
};

struct Mat3 {
    float m[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};   // m[col][row]
    float* operator[](int col) { return m[col]; }
    const float* operator[](int col) const { return m[col]; }
    const float* data() const { return &m[0][0]; }
    float* data() { return &m[0][0]; }
};

struct Mat4 {
    float m[4][4] = {{1, 0, 0, 0}, {0, 1, 0, 0}, {0, 0, 1, 0}, {0, 0, 0, 1}};   // m[col][row]
    const float* data() const { return &m[0][0]; }
    float* data() { return &m[0][0]; }
};

using CpuCloud = std::vector<Point_f>;

inline Vec3 operator*(const Mat3& a, const Vec3& v)
{
    Vec3 r;
    for (int i = 0; i < 3; i++) r[i] = a.m[0][i] * v.x + a.m[1][i] * v.y + a.m[2][i] * v.z;
    return r;
}

inline Mat3 operator*(float s, const Mat3& a)
{
    Mat3 r;
    for (int c = 0; c < 3; c++)
        for (int rr = 0; rr < 3; rr++) r.m[c][rr] = a.m[c][rr] * s;
    return r;
}

}  // namespace Common
