// Camera state of framework/include/trackball.h:13-56 without the window/mouse plumbing.
#pragma once
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
