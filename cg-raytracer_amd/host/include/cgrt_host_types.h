// cgrt_host_types.h -- the plain data types the reference's hot path exchanges, gathered in one header:
//   Ray                                   framework/include/ray.h:9-13
//   Vertex, Material, Triangle, Mesh      src/mesh.h:12-35
//   Plane, AxisAlignedBox, Sphere, lights, Scene, SceneType   src/scene.h:12-60
//   HitInfo                               src/ray_tracing.h:4-8
// Names, member names, member order and sizes are the reference's (static_asserts below), so code written against the
// reference's headers compiles against these; ray.h / mesh.h / scene.h in this directory only forward here.
#pragma once
#include <cstdint>
#include <filesystem>
#include <limits>
#include <vector>

#include "cgrt_vec.h"

// ---- geometry records ----------------------------------------------------------------------------------------
struct Vertex {
    cgrt::vec3 p;  // position
    cgrt::vec3 n;  // normal
};
using Triangle = cgrt::uvec3;  // three indices into Mesh::vertices

struct Material {
    cgrt::vec3 kd;        // diffuse colour
    cgrt::vec3 ks{0.0f};  // specular colour
    float shininess{1.0f};
    float transparency{1.0f};
};

struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<Triangle> triangles;
    Material material;
};

struct AxisAlignedBox {
    cgrt::vec3 lower{0.0f};
    cgrt::vec3 upper{1.0f};
};
struct Sphere {
    cgrt::vec3 center{0.0f};
    float radius = 1.0f;
    Material material;
};
struct Plane {
    float D = 0.0f;
    cgrt::vec3 normal{0.0f, 1.0f, 0.0f};
};

// ---- rays and hits ---------------------------------------------------------------------------------------------
struct Ray {
    cgrt::vec3 origin{0.0f};
    cgrt::vec3 direction{0.0f, 0.0f, -1.0f};
    float t{std::numeric_limits<float>::max()};
};
struct HitInfo {
    cgrt::vec3 normal;
    Material material;
};

// ---- scene -----------------------------------------------------------------------------------------------------
struct PointLight {
    cgrt::vec3 position;
    cgrt::vec3 color;
};
struct SphericalLight {
    cgrt::vec3 position;
    float radius;
    cgrt::vec3 color;
};
struct Scene {
    std::vector<Mesh> meshes;
    std::vector<Sphere> spheres;
    std::vector<PointLight> pointLights;
    std::vector<SphericalLight> sphericalLight;
};
enum SceneType { SingleTriangle, Cube, CornellBox, CornellBoxSphericalLight, Monkey, Dragon, Spheres, Custom };

static_assert(sizeof(Ray) == 28 && sizeof(Vertex) == 24 && sizeof(Material) == 32 && sizeof(HitInfo) == 44 &&
                  sizeof(AxisAlignedBox) == 24 && sizeof(Triangle) == 12,
              "layouts of the reference's PODs");

// ---- loading (src/mesh.cpp:58-166, src/scene.cpp:4-69) ------------------------------------------------------------
// loadMesh without assimp: OBJ + MTL with assimp 5.0.1's observable semantics (one vertex per face corner, one Mesh per
// object/group x material run, fan triangulation, flat normals when `vn` is absent, sibling objects in reverse file
// order because mesh.cpp walks the node tree with a stack).
[[nodiscard]] std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool normalize = false);
void centerAndScaleToUnitMesh(std::vector<Mesh>& meshes);
// Dragon throws when data/dragon.obj is absent (it is, upstream).
Scene loadScene(SceneType type, const std::filesystem::path& dataDir);
