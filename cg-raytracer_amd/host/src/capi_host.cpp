// C entry points of the host mirror for the Python test harness (ctypes): build a Scene from flat arrays or from
// an OBJ file, run the wavefront render driver, dump a loaded scene back to flat arrays.
#include <cstring>
#include <string>

#include "render.h"

namespace {
thread_local std::string g_err;
Scene scene_from_arrays(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                        const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights) {
    Scene sc;
    sc.meshes.resize(nmesh);
    std::vector<int64_t> vmin(nmesh, -1), vmax(nmesh, -1);
    for (uint32_t t = 0; t < ntris; t++)
        for (int k = 0; k < 3; k++) {
            const int64_t v = tri[3 * t + k];
            const uint32_t m = tri_mesh[t];
            if (vmin[m] < 0 || v < vmin[m]) vmin[m] = v;
            if (v > vmax[m]) vmax[m] = v;
        }
    for (uint32_t m = 0; m < nmesh; m++) {
        Mesh& M = sc.meshes[m];
        const float* k = materials + 8 * m;
        M.material.kd = cgrt::vec3(k[0], k[1], k[2]);
        M.material.ks = cgrt::vec3(k[3], k[4], k[5]);
        M.material.shininess = k[6];
        M.material.transparency = k[7];
        for (int64_t v = vmin[m]; v >= 0 && v <= vmax[m]; v++) {
            const float* p = pos_nrm + 6 * v;
            M.vertices.push_back(Vertex{cgrt::vec3(p[0], p[1], p[2]), cgrt::vec3(p[3], p[4], p[5])});
        }
    }
    for (uint32_t t = 0; t < ntris; t++) {
        const uint32_t m = tri_mesh[t];
        const uint32_t o = (uint32_t)vmin[m];
        sc.meshes[m].triangles.emplace_back(tri[3 * t] - o, tri[3 * t + 1] - o, tri[3 * t + 2] - o);
    }
    for (uint32_t l = 0; l < nlights; l++)
        sc.pointLights.push_back(PointLight{cgrt::vec3(lights[6 * l], lights[6 * l + 1], lights[6 * l + 2]),
                                            cgrt::vec3(lights[6 * l + 3], lights[6 * l + 4], lights[6 * l + 5])});
    (void)nverts;
    return sc;
}
}  // namespace

extern "C" {
const char* cgrt_host_last_error() { return g_err.c_str(); }

// cam: look_at(3) euler(3) distance fovy aspect.  stats (optional): primary, shadow, reflection ray counts, device seconds, total seconds.
int cgrt_host_render(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                     const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights, const float* cam, int W, int H,
                     int maxLevel, float* rgb, double* stats) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, lights, nlights);
        BoundingVolumeHierarchy bvh(&sc);
        Trackball camera(cam[7], cam[8], cam[6]);
        camera.setCamera(cgrt::vec3(cam[0], cam[1], cam[2]), cgrt::vec3(cam[3], cam[4], cam[5]), cam[6]);
        RenderStats st = renderToBuffer(sc, camera, bvh, W, H, rgb, maxLevel);
        if (stats) {
            stats[0] = (double)st.primary;
            stats[1] = (double)st.shadow;
            stats[2] = (double)st.reflection;
            stats[3] = st.seconds_device;
            stats[4] = st.seconds_total;
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Loads an OBJ with the host loader and reports sizes; a second call with buffers copies the flat arrays out.
int cgrt_host_load_obj(const char* path, int normalize, uint32_t* nverts, uint32_t* ntris, uint32_t* nmesh, float* pos_nrm, uint32_t* tri,
                       uint32_t* tri_mesh, float* materials) {
    try {
        std::vector<Mesh> meshes = loadMesh(path, normalize != 0);
        uint32_t nv = 0, nt = 0;
        for (const Mesh& m : meshes) {
            nv += (uint32_t)m.vertices.size();
            nt += (uint32_t)m.triangles.size();
        }
        *nverts = nv;
        *ntris = nt;
        *nmesh = (uint32_t)meshes.size();
        if (!pos_nrm) return 0;
        uint32_t vo = 0, to = 0, mi = 0;
        for (const Mesh& m : meshes) {
            for (const Vertex& v : m.vertices) {
                const float r[6] = {v.p.x, v.p.y, v.p.z, v.n.x, v.n.y, v.n.z};
                std::memcpy(pos_nrm + 6 * (size_t)(vo++), r, 24);
            }
            const uint32_t base = vo - (uint32_t)m.vertices.size();
            for (const Triangle& t : m.triangles) {
                tri[3 * to] = t[0] + base;
                tri[3 * to + 1] = t[1] + base;
                tri[3 * to + 2] = t[2] + base;
                tri_mesh[to++] = mi;
            }
            const Material& k = m.material;
            const float r[8] = {k.kd.x, k.kd.y, k.kd.z, k.ks.x, k.ks.y, k.ks.z, k.shininess, k.transparency};
            std::memcpy(materials + 8 * (size_t)mi, r, 32);
            mi++;
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Writes rgb (W*H*3 floats, index y*W+x, y up) through Screen::setPixel + writeBitmapToFile.
int cgrt_host_write_bmp(const char* path, const float* rgb, int W, int H) {
    try {
        Screen screen(W, H);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) {
                const float* p = rgb + 3 * ((size_t)y * W + x);
                screen.setPixel(x, y, cgrt::vec3(p[0], p[1], p[2]));
            }
        screen.writeBitmapToFile(path);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
}
