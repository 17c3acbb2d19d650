// Mirror of src/mesh.h:12-35.
#pragma once
#include <filesystem>
#include <vector>

#include "cgrt_vec.h"

struct Vertex {
    cgrt::vec3 p;  // Position.
    cgrt::vec3 n;  // Normal.
};

struct Material {
    cgrt::vec3 kd;  // Diffuse color.
    cgrt::vec3 ks{0.0f};
    float shininess{1.0f};
    float transparency{1.0f};
};
static_assert(sizeof(Vertex) == 24 && sizeof(Material) == 32, "mesh.h layouts");

using Triangle = cgrt::uvec3;

struct Mesh {
    std::vector<Vertex> vertices;
    std::vector<Triangle> triangles;
    Material material;
};

// loadMesh (src/mesh.cpp:58-141) without assimp: OBJ + MTL with assimp 5.0.1's observable semantics
// (one vertex per face corner, one Mesh per object/group x material run, fan triangulation, flat normals
// when `vn` is absent, sibling objects in reverse file order because mesh.cpp walks the node tree with a stack).
[[nodiscard]] std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool normalize = false);
// centerAndScaleToUnitMesh (src/mesh.cpp:143-166)
void centerAndScaleToUnitMesh(std::vector<Mesh>& meshes);
