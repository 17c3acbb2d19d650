// loadScene (src/scene.cpp:4-69), table-driven: which file, whether loadMesh normalises, which lights.
#include <iterator>

#include "scene.h"

namespace {
struct Preset {
    SceneType type;
    const char* file;  // nullptr: no mesh file
    bool normalize;
    int nlights;
    float lights[2][6];  // position, colour
};
const Preset kPresets[] = {
    {SingleTriangle, "triangle.obj", false, 1, {{-1, 1, -1, 1, 1, 1}}},
    {Cube, "cube.obj", false, 1, {{-1, 1, -1, 1, 1, 1}}},
    {CornellBox, "CornellBox-Mirror-Rotated.obj", true, 1, {{0, 0.58f, 0, 1, 1, 1}}},
    {CornellBoxSphericalLight, "CornellBox-Mirror-Rotated.obj", true, 0, {}},
    {Monkey, "monkey-rotated.obj", true, 2, {{-1, 1, -1, 1, 1, 1}, {1, -1, -1, 1, 1, 1}}},
    {Dragon, "dragon.obj", true, 1, {{-1, 1, -1, 1, 1, 1}}},
    {Spheres, nullptr, false, 1, {{3, 0, 3, 15, 15, 15}}},
    {Custom, "custom.obj", false, 1, {{-1, 1, -1, 1, 1, 1}}},
};
Sphere make_sphere(float x, float y, float z, float r, float kr, float kg, float kb) {
    Sphere s;
    s.center = cgrt::vec3(x, y, z);
    s.radius = r;
    s.material.kd = cgrt::vec3(kr, kg, kb);
    return s;
}
}  // namespace

Scene loadScene(SceneType type, const std::filesystem::path& dataDir) {
    Scene scene;
    for (const Preset& p : kPresets) {
        if (p.type != type) continue;
        if (p.file) {
            std::vector<Mesh> sub = loadMesh(dataDir / p.file, p.normalize);
            if (type == SingleTriangle) sub[0].material.kd = cgrt::vec3(1.0f);  // scene.cpp:11
            std::move(sub.begin(), sub.end(), std::back_inserter(scene.meshes));
        }
        for (int l = 0; l < p.nlights; l++)
            scene.pointLights.push_back(PointLight{cgrt::vec3(p.lights[l][0], p.lights[l][1], p.lights[l][2]),
                                                   cgrt::vec3(p.lights[l][3], p.lights[l][4], p.lights[l][5])});
        if (type == CornellBoxSphericalLight)
            scene.sphericalLight.push_back(SphericalLight{cgrt::vec3(0, 0.45f, 0), 0.1f, cgrt::vec3(1.0f)});
        if (type == Spheres) {  // scene.cpp:51-56
            scene.spheres.push_back(make_sphere(3.0f, -2.0f, 10.2f, 1.0f, 0.8f, 0.2f, 0.2f));
            scene.spheres.push_back(make_sphere(-2.0f, 2.0f, 4.0f, 2.0f, 0.6f, 0.8f, 0.2f));
            scene.spheres.push_back(make_sphere(0.0f, 0.0f, 6.0f, 0.75f, 0.2f, 0.2f, 0.8f));
        }
    }
    return scene;
}
