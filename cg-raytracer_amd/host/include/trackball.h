// Camera state of framework/include/trackball.h:13-56 without the window/mouse plumbing.
#pragma once
#include <cmath>
#include <limits>

#include "../../../include/cgrt.h"
#include "ray.h"

class Trackball {
public:
    Trackball(float fovy, float aspect, float distanceFromLookAt = 4.0f) : m_fovy(fovy), m_aspect(aspect), m_distanceFromLookAt(distanceFromLookAt) {}
    void setCamera(const cgrt::vec3 lookAt, const cgrt::vec3 rotations, const float dist) {  // trackball.cpp:58-63
        m_lookAt = lookAt;
        m_rotationEulerAngles = rotations;
        m_distanceFromLookAt = dist;
    }
    // glm::quat(eulerAngles) * v (glm 0.9.9.8 type_quat.inl: the half-angle products, then v + 2 (w (q x v) + q x (q x v))), the
    // arithmetic the device's fused ray generation uses (csrc/capi.cpp make_camera, walk_exact.h primary_ray): same bits.
    cgrt::vec3 rotate(const cgrt::vec3& v) const {
        const float hx = m_rotationEulerAngles.x * 0.5f, hy = m_rotationEulerAngles.y * 0.5f, hz = m_rotationEulerAngles.z * 0.5f;
        const float cx = std::cos(hx), cy = std::cos(hy), cz = std::cos(hz);
        const float sx = std::sin(hx), sy = std::sin(hy), sz = std::sin(hz);
        const float qw = cx * cy * cz + sx * sy * sz;
        const cgrt::vec3 q(sx * cy * cz - cx * sy * sz, cx * sy * cz + sx * cy * sz, cx * cy * sz - sx * sy * cz);
        const cgrt::vec3 uv = cgrt::cross(q, v);
        const cgrt::vec3 uuv = cgrt::cross(q, uv);
        return v + (uv * qw + uuv) * 2.0f;
    }
    cgrt::vec3 position() const { return m_lookAt + rotate(cgrt::vec3(0.0f, 0.0f, -m_distanceFromLookAt)); }  // trackball.cpp:70-73
    // trackball.cpp:92-103: the ray through `pixel` (normalised coordinates in [-1, 1]) of the virtual image plane
    Ray generateRay(const cgrt::vec2& pixel) const {
        const float halfScreenPlaceHeight = std::tan(m_fovy / 2.0f);
        const float halfScreenPlaceWidth = m_aspect * halfScreenPlaceHeight;
        const cgrt::vec3 cameraSpaceDirection = cgrt::normalize(cgrt::vec3(-pixel.x * halfScreenPlaceWidth, pixel.y * halfScreenPlaceHeight, 1.0f));
        Ray ray;
        ray.origin = position();
        ray.direction = rotate(cameraSpaceDirection);
        ray.t = std::numeric_limits<float>::max();
        return ray;
    }
    CgrtCamera abi() const {
        CgrtCamera c{{m_lookAt.x, m_lookAt.y, m_lookAt.z}, {m_rotationEulerAngles.x, m_rotationEulerAngles.y, m_rotationEulerAngles.z},
                     m_distanceFromLookAt, m_fovy, m_aspect};
        return c;
    }

private:
    float m_fovy, m_aspect;
    cgrt::vec3 m_lookAt{0.0f};
    float m_distanceFromLookAt;
    cgrt::vec3 m_rotationEulerAngles{0.0f};
};
