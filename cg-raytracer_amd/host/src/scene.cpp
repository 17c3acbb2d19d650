// loadScene (src/scene.cpp:4-69).
#include "scene.h"

#include <iterator>

Scene loadScene(SceneType type, const std::filesystem::path& dataDir) {
    Scene scene;
    auto take = [&](std::vector<Mesh> sub) { std::move(sub.begin(), sub.end(), std::back_inserter(scene.meshes)); };
    switch (type) {
        case SingleTriangle: {
            auto sub = loadMesh(dataDir / "triangle.obj");
            sub[0].material.kd = cgrt::vec3(1.0f);
            take(std::move(sub));
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
        } break;
        case Cube:
            take(loadMesh(dataDir / "cube.obj"));
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
            break;
        case CornellBox:
            take(loadMesh(dataDir / "CornellBox-Mirror-Rotated.obj", true));
            scene.pointLights.push_back(PointLight{cgrt::vec3(0, 0.58f, 0), cgrt::vec3(1.0f)});
            break;
        case CornellBoxSphericalLight:
            take(loadMesh(dataDir / "CornellBox-Mirror-Rotated.obj", true));
            scene.sphericalLight.push_back(SphericalLight{cgrt::vec3(0, 0.45f, 0), 0.1f, cgrt::vec3(1.0f)});
            break;
        case Monkey:
            take(loadMesh(dataDir / "monkey-rotated.obj", true));
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
            scene.pointLights.push_back(PointLight{cgrt::vec3(1, -1, -1), cgrt::vec3(1.0f)});
            break;
        case Dragon:
            take(loadMesh(dataDir / "dragon.obj", true));
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
            break;
        case Spheres: {
            Material a, b, c;
            a.kd = cgrt::vec3(0.8f, 0.2f, 0.2f);
            b.kd = cgrt::vec3(0.6f, 0.8f, 0.2f);
            c.kd = cgrt::vec3(0.2f, 0.2f, 0.8f);
            scene.spheres.push_back(Sphere{cgrt::vec3(3.0f, -2.0f, 10.2f), 1.0f, a});
            scene.spheres.push_back(Sphere{cgrt::vec3(-2.0f, 2.0f, 4.0f), 2.0f, b});
            scene.spheres.push_back(Sphere{cgrt::vec3(0.0f, 0.0f, 6.0f), 0.75f, c});
            scene.pointLights.push_back(PointLight{cgrt::vec3(3, 0, 3), cgrt::vec3(15.0f)});
        } break;
        case Custom:
            take(loadMesh(dataDir / "custom.obj"));
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
            break;
    }
    return scene;
}
