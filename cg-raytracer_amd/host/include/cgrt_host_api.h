// cgrt_host_api.h -- the callable surface of the reference's hot path over the C-ABI (include/cgrt.h):
//   free functions of src/ray_tracing.h:10-20 and class BoundingVolumeHierarchy of src/bounding_volume_hierarchy.h:15-56.
// Every function runs on the GPU (one-element batches for the per-ray forms): the product has no host implementation
// of the traversal or of the intersection arithmetic.  ray_tracing.h / bounding_volume_hierarchy.h only forward here.
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "cgrt_host_types.h"

struct CgrtScene;
struct CgrtCamera;

// ---- src/ray_tracing.h:10-20 -----------------------------------------------------------------------------------
Plane trianglePlane(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2);
bool pointInTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, const cgrt::vec3& n, const cgrt::vec3& p);
bool intersectRayWithPlane(const Plane& plane, Ray& ray);
bool intersectRayWithTriangle(const cgrt::vec3& v0, const cgrt::vec3& v1, const cgrt::vec3& v2, Ray& ray, HitInfo& hitInfo,
                              const cgrt::vec3& n1, const cgrt::vec3& n2, const cgrt::vec3& n3);
bool intersectRayWithShape(const AxisAlignedBox& box, Ray& ray);
bool intersectRayWithShape(const Sphere& sphere, Ray& ray, HitInfo& hitInfo);
bool intersectRayWithShape(const Mesh& mesh, Ray& ray, HitInfo& hitInfo);

// ---- src/bounding_volume_hierarchy.h:15-56 -----------------------------------------------------------------------
// Same public members; copy-assignable like the original (main.cpp:776-777 assigns a fresh BVH on scene change).
class BoundingVolumeHierarchy {
public:
    BoundingVolumeHierarchy(Scene* pScene);  // builds on the host, uploads to HIP device 0 (CGRT_DEVICE env overrides)
    BoundingVolumeHierarchy(Scene* pScene, int device);  // a replica on a given device (renderRayTracingOnDevices)

    void debugDraw(int level);  // GL-only upstream (bvh.cpp:469-525): kept as a no-op
    int numLevels() const;

    // Return true if something is hit (bvh.cpp:850-881).  One-ray batch through the GPU path; re-entrant: the reference calls
    // it on one const object from an omp parallel for (main.cpp:653-656), each call here runs on a private stream of the
    // library and waits for that stream only.  Batch callers should use intersectBatch / tracePrimary.
    bool intersect(Ray& ray, HitInfo& hitInfo) const;

    // ---- batched extensions (no upstream counterpart) ----
    // n independent intersect() calls: rays[i].t and hitInfos[i] are updated exactly like n sequential calls;
    // hit[i] receives the return value; primIds (optional) the primitive id of include/cgrt.h.
    void intersectBatch(Ray* rays, HitInfo* hitInfos, uint8_t* hit, size_t n, uint32_t* primIds = nullptr) const;
    // Whole primary frame with on-device ray generation; outputs indexed y*W+x.
    void tracePrimary(const CgrtCamera& cam, int W, int H, Ray* rays, HitInfo* hitInfos, uint8_t* hit) const;

    CgrtScene* handle() const { return m_handle.get(); }

private:
    Scene* m_pScene;
    std::shared_ptr<CgrtScene> m_handle;
    std::vector<Material> m_materials;  // per mesh, to fill HitInfo::material from material_id
};
