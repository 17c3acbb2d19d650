// BoundingVolumeHierarchy over the C-ABI (include/cgrt.h).  Replaces src/bounding_volume_hierarchy.cpp.
#include "bounding_volume_hierarchy.h"

#include <cstdlib>
#include <cstring>
#include <memory>
#include <stdexcept>
#include <string>

#include "../../../include/cgrt.h"
#include "host_threads.h"

namespace {
int host_device() {
    const char* e = std::getenv("CGRT_DEVICE");
    return e ? std::atoi(e) : 0;
}
[[noreturn]] void fail(const char* what) { throw std::runtime_error(std::string(what) + ": " + cgrt_last_error()); }
}  // namespace

BoundingVolumeHierarchy::BoundingVolumeHierarchy(Scene* pScene) : BoundingVolumeHierarchy(pScene, host_device()) {}

BoundingVolumeHierarchy::BoundingVolumeHierarchy(Scene* pScene, int device) : m_pScene(pScene) {
    // flatten Scene::meshes in load order (global primitive id = prefix over meshes + index in mesh.triangles)
    std::vector<float> pos_nrm, mats, sph;
    std::vector<uint32_t> tri, tri_mesh;
    uint32_t voff = 0, m = 0;
    for (const Mesh& mesh : pScene->meshes) {
        for (const Vertex& v : mesh.vertices) pos_nrm.insert(pos_nrm.end(), {v.p.x, v.p.y, v.p.z, v.n.x, v.n.y, v.n.z});
        for (const Triangle& t : mesh.triangles) {
            tri.insert(tri.end(), {t[0] + voff, t[1] + voff, t[2] + voff});
            tri_mesh.push_back(m);
        }
        const Material& k = mesh.material;
        mats.insert(mats.end(), {k.kd.x, k.kd.y, k.kd.z, k.ks.x, k.ks.y, k.ks.z, k.shininess, k.transparency});
        m_materials.push_back(k);
        voff += (uint32_t)mesh.vertices.size();
        m++;
    }
    for (const Sphere& s : pScene->spheres) sph.insert(sph.end(), {s.center.x, s.center.y, s.center.z, s.radius, -1.0f});
    CgrtScene* h = nullptr;
    if (cgrt_scene_create(pos_nrm.data(), voff, tri.data(), tri_mesh.data(), (uint32_t)tri_mesh.size(), mats.data(), m, sph.data(),
                          (uint32_t)pScene->spheres.size(), device, &h) != CGRT_OK)
        fail("cgrt_scene_create");
    m_handle = std::shared_ptr<CgrtScene>(h, cgrt_scene_destroy);
}

void BoundingVolumeHierarchy::debugDraw(int) {}
int BoundingVolumeHierarchy::numLevels() const { return cgrt_num_levels(m_handle.get()); }

void BoundingVolumeHierarchy::intersectBatch(Ray* rays, HitInfo* hitInfos, uint8_t* hit, size_t n, uint32_t* primIds) const {
    if (n == 0) return;
    static_assert(sizeof(Ray) == sizeof(CgrtRay), "Ray == CgrtRay");
    // HitInfo is left untouched on a miss: normal and material are taken over only where the ray hit.  A handful of rays (the
    // per-ray call sites: one) stays on the stack; the loops over a wavefront's list run on the host's cores.
    constexpr size_t SMALL = 64;
    CgrtHit hits_small[SMALL];
    float normals_small[3 * SMALL];
    std::unique_ptr<CgrtHit[]> hits_big;
    std::unique_ptr<float[]> normals_big;
    CgrtHit* hits = hits_small;
    float* normals = normals_small;
    if (n > SMALL) {
        hits_big.reset(new CgrtHit[n]);  // not value-initialised: the call writes every hit, and the normals are read only with a hit
        normals_big.reset(new float[3 * n]);
        hits = hits_big.get();
        normals = normals_big.get();
    }
    if (cgrt_intersect_batch(m_handle.get(), reinterpret_cast<const CgrtRay*>(rays), n, hits, normals) != CGRT_OK) fail("cgrt_intersect_batch");
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam()) if (n > 16384)
    for (size_t i = 0; i < n; i++) {
        rays[i].t = hits[i].t;
        hit[i] = (uint8_t)hits[i].hit;
        if (hits[i].hit) hitInfos[i].normal = cgrt::vec3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
        if (hits[i].material_id >= 0) hitInfos[i].material = m_materials[hits[i].material_id];  // bvh.cpp:547
        if (primIds) primIds[i] = hits[i].prim_id;
    }
}

bool BoundingVolumeHierarchy::intersect(Ray& ray, HitInfo& hitInfo) const {
    uint8_t h = 0;
    intersectBatch(&ray, &hitInfo, &h, 1);
    return h != 0;
}

void BoundingVolumeHierarchy::tracePrimary(const CgrtCamera& cam, int W, int H, Ray* rays, HitInfo* hitInfos, uint8_t* hit) const {
    const size_t n = (size_t)W * H;
    std::unique_ptr<CgrtHit[]> hits(new CgrtHit[n]);
    std::unique_ptr<float[]> normals(new float[3 * n]);
    if (cgrt_trace_primary(m_handle.get(), &cam, W, H, 0, 0, W, H, 0, 1, hits.get(), normals.get()) != CGRT_OK) fail("cgrt_trace_primary");
    if (cgrt_generate_rays(m_handle.get(), &cam, W, H, 0, 0, W, H, reinterpret_cast<CgrtRay*>(rays)) != CGRT_OK) fail("cgrt_generate_rays");
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam()) if (n > 16384)
    for (size_t i = 0; i < n; i++) {
        rays[i].t = hits[i].t;
        hit[i] = (uint8_t)hits[i].hit;
        if (hits[i].hit) hitInfos[i].normal = cgrt::vec3(normals[3 * i], normals[3 * i + 1], normals[3 * i + 2]);
        if (hits[i].material_id >= 0) hitInfos[i].material = m_materials[hits[i].material_id];
    }
}
