// Headless stand-in for the reference's interactive main() (src/main.cpp:722-939): load a scene preset or an OBJ,
// build the BVH, render with the reference's default camera, write render.bmp, print the timing the reference prints
// (main.cpp:791-797).
//   render [--gpus N | --per-ray [--threads T]] <data-dir> <triangle|cube|cornell|monkey|dragon|custom|file.obj> [W H [maxLevel [out.bmp]]]
// --gpus N: the frame is split over N devices (replica i on device i % cgrt_device_count(), super-tiles i % N), the whole
// shading driver on the devices, one Screen (renderRayTracingOnDevices).
// --per-ray: the reference's own structure, literally (main.cpp:265-310, :648-696): `omp parallel for` over rows, per-pixel
// recursive getFinalColor, ONE bvh.intersect call per ray from T threads (default: OpenMP's) -- what an unchanged main.cpp does
// to the library; concurrent calls share launches inside it (cgrt_set_call_combining).
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <memory>

#include "render.h"

extern "C" int cgrt_device_count(void);

int main(int argc, char** argv) {
    int gpus = 0, threads = 0;
    bool perRay = false;
    for (;;) {
        if (argc > 2 && std::strcmp(argv[1], "--gpus") == 0) {
            gpus = std::atoi(argv[2]);
            argv += 2;
            argc -= 2;
        } else if (argc > 1 && std::strcmp(argv[1], "--per-ray") == 0) {
            perRay = true;
            argv += 1;
            argc -= 1;
        } else if (argc > 2 && std::strcmp(argv[1], "--threads") == 0) {
            threads = std::atoi(argv[2]);
            argv += 2;
            argc -= 2;
        } else {
            break;
        }
    }
    if (argc < 3) {
        std::cerr << "usage: render [--gpus N | --per-ray [--threads T]] <data-dir> <scene|file.obj> [W H [maxLevel [out.bmp]]]\n";
        return 2;
    }
    const std::filesystem::path dataDir = argv[1];
    const std::string what = argv[2];
    const int W = argc > 4 ? std::atoi(argv[3]) : 800, H = argc > 4 ? std::atoi(argv[4]) : 800;  // windowResolution, main.cpp:29
    const int maxLevel = argc > 5 ? std::atoi(argv[5]) : 2;
    const std::string out = argc > 6 ? argv[6] : "render.bmp";
    try {
        Scene scene;
        if (what == "triangle") scene = loadScene(SingleTriangle, dataDir);
        else if (what == "cube") scene = loadScene(Cube, dataDir);
        else if (what == "cornell") scene = loadScene(CornellBox, dataDir);
        else if (what == "cornell-spherical") scene = loadScene(CornellBoxSphericalLight, dataDir);
        else if (what == "monkey") scene = loadScene(Monkey, dataDir);
        else if (what == "dragon") scene = loadScene(Dragon, dataDir);
        else if (what == "custom") scene = loadScene(Custom, dataDir);
        else {
            auto sub = loadMesh(what, true);
            scene.meshes = std::move(sub);
            scene.pointLights.push_back(PointLight{cgrt::vec3(-1, 1, -1), cgrt::vec3(1.0f)});
        }
        BoundingVolumeHierarchy bvh{&scene};
        const float rad = 0.01745329251994329576923690768489f;
        Trackball camera{50.0f * rad, float(W) / float(H), 3.0f};  // main.cpp:730-731
        camera.setCamera(cgrt::vec3(0.0f), cgrt::vec3(20.0f * rad, 20.0f * rad, 0.0f), 3.0f);
        Screen screen{W, H};
        const auto start = std::chrono::high_resolution_clock::now();
        // CGRT_RENDER_ON_DEVICE=1: the whole shading/recursion driver on the GPU instead of the host-driven wavefront
        const char* od = std::getenv("CGRT_RENDER_ON_DEVICE");
        RenderStats st;
        if (perRay) {
            st = renderRayTracingPerRay(scene, camera, bvh, screen, maxLevel, nullptr, threads);
        } else if (gpus > 0) {
            const int ndev = cgrt_device_count() > 0 ? cgrt_device_count() : 1;
            std::vector<std::unique_ptr<BoundingVolumeHierarchy>> own;
            std::vector<const BoundingVolumeHierarchy*> bvhs;
            for (int i = 0; i < gpus; i++) {
                own.emplace_back(new BoundingVolumeHierarchy(&scene, i % ndev));
                bvhs.push_back(own.back().get());
            }
            st = renderRayTracingOnDevices(scene, camera, bvhs, screen, maxLevel);
        } else {
            st = (od && od[0] == '1') ? renderRayTracingOnDevice(scene, camera, bvh, screen, maxLevel)
                                      : renderRayTracing(scene, camera, bvh, screen, maxLevel);
        }
        const auto end = std::chrono::high_resolution_clock::now();
        std::cout << "Time to render image: " << std::chrono::duration<float, std::milli>(end - start).count() << " milliseconds" << std::endl;
        std::cout << "BVH levels " << bvh.numLevels() << "; rays: " << st.primary << " primary, " << st.shadow << " shadow, " << st.reflection
                  << " reflection, " << st.softShadow << " soft-shadow; device share " << st.seconds_device << " s" << std::endl;
        screen.writeBitmapToFile(out);
    } catch (const std::exception& e) {
        std::cerr << "error: " << e.what() << std::endl;
        return 1;
    }
    return 0;
}
