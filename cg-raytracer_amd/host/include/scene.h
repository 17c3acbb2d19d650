// Mirror of src/scene.h:12-63.
#pragma once
#include <filesystem>
#include <vector>

#include "mesh.h"
#include "ray.h"

enum SceneType { SingleTriangle, Cube, CornellBox, CornellBoxSphericalLight, Monkey, Dragon, Spheres, Custom };

struct Plane {
    float D = 0.0f;
    cgrt::vec3 normal{0.0f, 1.0f, 0.0f};
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

// loadScene (src/scene.cpp:4-69).  Dragon throws when data/dragon.obj is absent (it is, upstream).
Scene loadScene(SceneType type, const std::filesystem::path& dataDir);
