// cgrt_vec.h -- vector types of the host mirror.
// With -DCGRT_HOST_USE_GLM the mirror uses the caller's glm (the reference's own types, so reference-style
// callers compile unchanged).  glm is not in this image, so by default a minimal POD with the same layout
// (3 packed floats) and the scalar formulas of glm 0.9.9.8 is used.  It lives in namespace cgrt, is used only
// by this repo's own host code, and is never put on an include path to compile reference sources.
#pragma once
#include <cmath>
#include <cstdint>

#ifdef CGRT_HOST_USE_GLM
#include <glm/geometric.hpp>
#include <glm/vec2.hpp>
#include <glm/vec3.hpp>
namespace cgrt {
using vec2 = glm::vec2;
using vec3 = glm::vec3;
using uvec3 = glm::uvec3;
using glm::cross;
using glm::dot;
using glm::length;
using glm::normalize;
using glm::reflect;
}  // namespace cgrt
#else
namespace cgrt {
struct vec2 {
    float x = 0, y = 0;
    vec2() = default;
    vec2(float x_, float y_) : x(x_), y(y_) {}
};
struct vec3 {
    float x = 0, y = 0, z = 0;
    vec3() = default;
    explicit vec3(float s) : x(s), y(s), z(s) {}
    vec3(float x_, float y_, float z_) : x(x_), y(y_), z(z_) {}
    float& operator[](int i) { return (&x)[i]; }
    const float& operator[](int i) const { return (&x)[i]; }
    vec3& operator+=(const vec3& b) {
        x += b.x, y += b.y, z += b.z;
        return *this;
    }
};
struct uvec3 {
    uint32_t x = 0, y = 0, z = 0;
    uvec3() = default;
    uvec3(uint32_t a, uint32_t b, uint32_t c) : x(a), y(b), z(c) {}
    uint32_t& operator[](int i) { return (&x)[i]; }
    const uint32_t& operator[](int i) const { return (&x)[i]; }
};
inline vec3 operator+(vec3 a, vec3 b) { return vec3(a.x + b.x, a.y + b.y, a.z + b.z); }
inline vec3 operator-(vec3 a, vec3 b) { return vec3(a.x - b.x, a.y - b.y, a.z - b.z); }
inline vec3 operator-(vec3 a) { return vec3(-a.x, -a.y, -a.z); }
inline vec3 operator*(vec3 a, float s) { return vec3(a.x * s, a.y * s, a.z * s); }
inline vec3 operator*(float s, vec3 a) { return vec3(s * a.x, s * a.y, s * a.z); }
inline vec3 operator*(vec3 a, vec3 b) { return vec3(a.x * b.x, a.y * b.y, a.z * b.z); }
inline vec3 operator/(vec3 a, float s) { return vec3(a.x / s, a.y / s, a.z / s); }
inline float dot(vec3 a, vec3 b) {
    vec3 t = a * b;
    return t.x + t.y + t.z;
}
inline vec3 cross(vec3 a, vec3 b) { return vec3(a.y * b.z - b.y * a.z, a.z * b.x - b.z * a.x, a.x * b.y - b.x * a.y); }
inline vec3 normalize(vec3 v) { return v * (1.0f / std::sqrt(dot(v, v))); }
inline float length(vec3 v) { return std::sqrt(dot(v, v)); }
inline vec3 reflect(vec3 I, vec3 N) { return I - N * dot(N, I) * 2.0f; }
}  // namespace cgrt
#endif
static_assert(sizeof(cgrt::vec3) == 12, "vec3 must be 3 packed floats");
