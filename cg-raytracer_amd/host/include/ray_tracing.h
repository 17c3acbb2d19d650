// Mirror of src/ray_tracing.h:4-20.  Every function runs on the GPU through the C-ABI (one-element batch):
// there is no host implementation of the arithmetic in the product.
#pragma once
#include "scene.h"

struct HitInfo {
    cgrt::vec3 normal;
    Material material;
};
static_assert(sizeof(HitInfo) == 44, "HitInfo layout (ray_tracing.h:4-8)");

bool intersectRayWithPlane(const Plane& plane, Ray& ray);
bool pointInTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, const cgrt::vec3& n, const cgrt::vec3& p);
Plane trianglePlane(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2);
bool intersectRayWithTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, Ray& ray, HitInfo& hitInfo,
                              const cgrt::vec3& n1, const cgrt::vec3& n2, const cgrt::vec3& n3);
bool intersectRayWithShape(const Sphere& sphere, Ray& ray, HitInfo& hitInfo);
bool intersectRayWithShape(const AxisAlignedBox& box, Ray& ray);
bool intersectRayWithShape(const Mesh& mesh, Ray& ray, HitInfo& hitInfo);
