// Mirror of src/bounding_volume_hierarchy.h:15-56: same public declarations, backed by a CgrtScene
// (include/cgrt.h).  Copy-assignable like the original (main.cpp:776-777 assigns a fresh BVH on scene change).
#pragma once
#include <cstdint>
#include <memory>
#include <vector>

#include "ray_tracing.h"
#include "scene.h"

struct CgrtScene;
struct CgrtCamera;

class BoundingVolumeHierarchy {
public:
    BoundingVolumeHierarchy(Scene* pScene);  // builds on the host, uploads to HIP device 0 (CGRT_DEVICE env overrides)

    void debugDraw(int level);  // GL-only upstream (bvh.cpp:469-525): kept as a no-op
    int numLevels() const;

    // Return true if something is hit (bvh.cpp:850-881).  One-ray batch through the GPU path: correct, slow;
    // batch callers should use intersectBatch / tracePrimary.
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
