// The free functions of src/ray_tracing.h:10-20 over the element-wise device entries of include/cgrt.h.
#include "ray_tracing.h"

#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <string>

#include "../../../include/cgrt.h"

namespace {
int dev() {
    const char* e = std::getenv("CGRT_DEVICE");
    return e ? std::atoi(e) : 0;
}
void check(int rc, const char* what) {
    if (rc != CGRT_OK) throw std::runtime_error(std::string(what) + ": " + cgrt_last_error());
}
}  // namespace

bool intersectRayWithPlane(const Plane& plane, Ray& ray) {
    const float p[4] = {plane.D, plane.normal.x, plane.normal.y, plane.normal.z};
    uint8_t hit = 0;
    float t = ray.t;
    check(cgrt_ray_plane_batch(dev(), p, reinterpret_cast<const CgrtRay*>(&ray), 1, &t, &hit), "cgrt_ray_plane_batch");
    ray.t = t;
    return hit != 0;
}

bool pointInTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, const cgrt::vec3& n, const cgrt::vec3& p) {
    const float in[15] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, n.x, n.y, n.z, p.x, p.y, p.z};
    uint8_t out = 0;
    check(cgrt_point_in_triangle_batch(dev(), in, 1, &out), "cgrt_point_in_triangle_batch");
    return out != 0;
}

Plane trianglePlane(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2) {
    const float in[9] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z};
    float out[4];
    check(cgrt_triangle_plane_batch(dev(), in, 1, out), "cgrt_triangle_plane_batch");
    Plane pl;
    pl.D = out[0];
    pl.normal = cgrt::vec3(out[1], out[2], out[3]);
    return pl;
}

bool intersectRayWithTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, Ray& ray, HitInfo& hitInfo,
                              const cgrt::vec3& n1, const cgrt::vec3& n2, const cgrt::vec3& n3) {
    const float tri[18] = {v0.x, v0.y, v0.z, v1.x, v1.y, v1.z, v2.x, v2.y, v2.z, n1.x, n1.y, n1.z, n2.x, n2.y, n2.z, n3.x, n3.y, n3.z};
    uint8_t hit = 0;
    float t = ray.t, nrm[3] = {hitInfo.normal.x, hitInfo.normal.y, hitInfo.normal.z};
    check(cgrt_ray_triangle_batch(dev(), tri, reinterpret_cast<const CgrtRay*>(&ray), 1, &t, &hit, nrm), "cgrt_ray_triangle_batch");
    ray.t = t;
    if (hit) hitInfo.normal = cgrt::vec3(nrm[0], nrm[1], nrm[2]);
    return hit != 0;
}

bool intersectRayWithShape(const Sphere& sphere, Ray& ray, HitInfo& hitInfo) {
    const float s[4] = {sphere.center.x, sphere.center.y, sphere.center.z, sphere.radius};
    uint8_t hit = 0;
    float t = ray.t, nrm[3] = {hitInfo.normal.x, hitInfo.normal.y, hitInfo.normal.z};
    check(cgrt_ray_sphere_batch(dev(), s, reinterpret_cast<const CgrtRay*>(&ray), 1, &t, &hit, nrm), "cgrt_ray_sphere_batch");
    ray.t = t;
    if (hit) hitInfo.normal = cgrt::vec3(nrm[0], nrm[1], nrm[2]);
    return hit != 0;
}

bool intersectRayWithShape(const AxisAlignedBox& box, Ray& ray) {
    const float b[6] = {box.lower.x, box.lower.y, box.lower.z, box.upper.x, box.upper.y, box.upper.z};
    uint8_t hit = 0;
    float t = ray.t;
    check(cgrt_ray_box_batch(dev(), b, reinterpret_cast<const CgrtRay*>(&ray), 1, &t, &hit, nullptr), "cgrt_ray_box_batch");
    ray.t = t;  // on success the reference leaves the box parameter in ray.t (ray_tracing.cpp:198)
    return hit != 0;
}

// ray_tracing.cpp:202-213: every triangle of the mesh, in order, against the running ray.t
bool intersectRayWithShape(const Mesh& mesh, Ray& ray, HitInfo& hitInfo) {
    bool hit = false;
    for (const auto& tri : mesh.triangles) {
        const Vertex &v0 = mesh.vertices[tri[0]], &v1 = mesh.vertices[tri[1]], &v2 = mesh.vertices[tri[2]];
        hit |= intersectRayWithTriangle(v0.p, v1.p, v2.p, ray, hitInfo, v0.n, v1.n, v2.n);
    }
    return hit;
}
