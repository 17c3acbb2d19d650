// C entry points of the host mirror for the Python test harness (ctypes): build a Scene from flat arrays or from
// an OBJ file, run the wavefront render driver, dump a loaded scene back to flat arrays.
#include <algorithm>
#include <chrono>
#include <cstring>
#include <memory>
#include <string>
#include <thread>

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

// The same frame through the reference's driver taken literally (renderToBufferPerRay: omp parallel for over rows, per-pixel
// recursion, ONE BoundingVolumeHierarchy::intersect call per ray) with `threads` caller threads (0 = OpenMP's default).
int cgrt_host_render_per_ray(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                             const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights, const float* cam, int W, int H,
                             int maxLevel, int threads, float* rgb, double* stats) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, lights, nlights);
        BoundingVolumeHierarchy bvh(&sc);
        Trackball camera(cam[7], cam[8], cam[6]);
        camera.setCamera(cgrt::vec3(cam[0], cam[1], cam[2]), cgrt::vec3(cam[3], cam[4], cam[5]), cam[6]);
        RenderStats st = renderToBufferPerRay(sc, camera, bvh, W, H, rgb, maxLevel, nullptr, threads);
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

// Whole-call times of the mirror's drivers that fill a Screen (what a caller of renderRayTracing* waits for), BVH built once:
// ms[0] = renderRayTracingOnDevice (device driver -> pinned frame -> Screen::setFrame), ms[1] = renderRayTracing (host-driven
// wavefront), each the best of `reps` calls; ms[2] = device share of the best OnDevice call.
int cgrt_host_time_screen_render(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                                 const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights, const float* cam, int W, int H,
                                 int maxLevel, int reps, double* ms) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, lights, nlights);
        BoundingVolumeHierarchy bvh(&sc);
        Trackball camera(cam[7], cam[8], cam[6]);
        camera.setCamera(cgrt::vec3(cam[0], cam[1], cam[2]), cgrt::vec3(cam[3], cam[4], cam[5]), cam[6]);
        Screen screen(W, H);
        using Clk = std::chrono::steady_clock;
        ms[0] = ms[1] = 1e30;
        ms[2] = 0;
        for (int r = 0; r < reps + 1; r++) {  // (the first call grows the library's workspaces)
            const auto t0 = Clk::now();
            const RenderStats st = renderRayTracingOnDevice(sc, camera, bvh, screen, maxLevel);
            const double t = std::chrono::duration<double, std::milli>(Clk::now() - t0).count();
            if (r > 0 && t < ms[0]) {
                ms[0] = t;
                ms[2] = st.seconds_device * 1e3;
            }
        }
        for (int r = 0; r < reps; r++) {
            const auto t0 = Clk::now();
            (void)renderRayTracing(sc, camera, bvh, screen, maxLevel);
            ms[1] = std::min(ms[1], std::chrono::duration<double, std::milli>(Clk::now() - t0).count());
        }
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// + spherical lights (S x 7: position, radius, color) sampled from `units` (nunits x 3, 0 -> SoftShadowSampler::gaussian()).
// stats: primary, shadow, reflection, soft-shadow ray counts, device seconds, total seconds.
// on_device: 0 = the host-driven wavefront (renderToBuffer), 1 = the whole driver on the device (renderToBufferOnDevice).
int cgrt_host_render_soft(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                          const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights, const float* spherical,
                          uint32_t nspherical, const float* units, uint32_t nunits, uint32_t samples, uint32_t seed, const float* cam, int W,
                          int H, int maxLevel, float* rgb, double* stats, int on_device) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, lights, nlights);
        for (uint32_t i = 0; i < nspherical; i++) {
            const float* q = spherical + 7 * i;
            sc.sphericalLight.push_back(SphericalLight{cgrt::vec3(q[0], q[1], q[2]), q[3], cgrt::vec3(q[4], q[5], q[6])});
        }
        SoftShadowSampler sampler;
        if (nunits) {
            for (uint32_t i = 0; i < nunits; i++) sampler.units.push_back(cgrt::vec3(units[3 * i], units[3 * i + 1], units[3 * i + 2]));
        } else {
            sampler = SoftShadowSampler::gaussian();
        }
        sampler.samples = samples;
        sampler.seed = seed;
        BoundingVolumeHierarchy bvh(&sc);
        Trackball camera(cam[7], cam[8], cam[6]);
        camera.setCamera(cgrt::vec3(cam[0], cam[1], cam[2]), cgrt::vec3(cam[3], cam[4], cam[5]), cam[6]);
        RenderStats st = on_device ? renderToBufferOnDevice(sc, camera, bvh, W, H, rgb, maxLevel, &sampler)
                                   : renderToBuffer(sc, camera, bvh, W, H, rgb, maxLevel, &sampler);
        if (stats) {
            stats[0] = (double)st.primary;
            stats[1] = (double)st.shadow;
            stats[2] = (double)st.reflection;
            stats[3] = (double)st.softShadow;
            stats[4] = st.seconds_device;
            stats[5] = st.seconds_total;
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

// Exercises the drop-in C++ surface the way reference code uses it (main.cpp:276, :115, :776-777): per-ray
// BoundingVolumeHierarchy::intersect(Ray&, HitInfo&), copy-assignment of a freshly built BVH, numLevels(), and the
// free functions of ray_tracing.h, and checks them against the batched entry on the same rays.  rays: n x 7 floats.
// Returns the number of disagreements (0 = pass), -1 on an exception.
int cgrt_host_selftest(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                       const float* materials, uint32_t nmesh, const float* rays7, uint32_t nrays, int* levels_out) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, nullptr, 0);
        BoundingVolumeHierarchy bvh(&sc);
        {
            Scene empty;
            BoundingVolumeHierarchy other(&empty);
            other = BoundingVolumeHierarchy(&sc);  // main.cpp:776-777
            bvh = other;
        }
        if (levels_out) *levels_out = bvh.numLevels();
        bvh.debugDraw(0);
        std::vector<Ray> rays(nrays);
        for (uint32_t i = 0; i < nrays; i++) {
            const float* p = rays7 + 7 * i;
            rays[i].origin = cgrt::vec3(p[0], p[1], p[2]);
            rays[i].direction = cgrt::vec3(p[3], p[4], p[5]);
            rays[i].t = p[6];
        }
        std::vector<Ray> batch = rays;
        std::vector<HitInfo> bhi(nrays);
        std::vector<uint8_t> bhit(nrays);
        for (auto& h : bhi) {
            h.normal = cgrt::vec3(7.0f, 8.0f, 9.0f);
            h.material.shininess = -3.0f;
        }
        bvh.intersectBatch(batch.data(), bhi.data(), bhit.data(), nrays);
        int bad = 0;
        for (uint32_t i = 0; i < nrays; i++) {
            Ray r = rays[i];
            HitInfo hi;
            hi.normal = cgrt::vec3(7.0f, 8.0f, 9.0f);
            hi.material.shininess = -3.0f;
            const bool hit = bvh.intersect(r, hi);  // one ray at a time, as the reference calls it
            bad += hit != (bhit[i] != 0);
            bad += std::memcmp(&r.t, &batch[i].t, 4) != 0;
            bad += std::memcmp(&hi.normal, &bhi[i].normal, 12) != 0;
            bad += std::memcmp(&hi.material, &bhi[i].material, sizeof(Material)) != 0;
            if (!hit) bad += !(hi.normal.x == 7.0f && hi.material.shininess == -3.0f);  // HitInfo untouched on a miss
        }
        // free functions on the first triangle of the scene against the same ray through the Mesh overload
        if (!sc.meshes.empty() && !sc.meshes[0].triangles.empty() && nrays > 0) {
            const Mesh& m = sc.meshes[0];
            const Triangle& t = m.triangles[0];
            const Vertex &v0 = m.vertices[t[0]], &v1 = m.vertices[t[1]], &v2 = m.vertices[t[2]];
            const Plane pl = trianglePlane(v0.p, v1.p, v2.p);
            for (uint32_t i = 0; i < nrays && i < 64; i++) {
                Ray a = rays[i], b = rays[i];
                HitInfo ha, hb;
                const bool h1 = intersectRayWithTriangle(v0.p, v1.p, v2.p, a, ha, v0.n, v1.n, v2.n);
                bool h2 = false;
                const float prev = b.t;
                if (intersectRayWithPlane(pl, b)) {
                    if (pointInTriangle(v0.p, v1.p, v2.p, pl.normal, b.origin + b.direction * b.t))
                        h2 = true;
                    else
                        b.t = prev;
                }
                bad += h1 != h2;
                bad += std::memcmp(&a.t, &b.t, 4) != 0;
                Mesh one;
                one.vertices = {v0, v1, v2};
                one.triangles = {Triangle(0, 1, 2)};
                Ray c = rays[i];
                HitInfo hc;
                bad += intersectRayWithShape(one, c, hc) != h1;
                bad += std::memcmp(&c.t, &a.t, 4) != 0;
                AxisAlignedBox box{cgrt::vec3(-0.5f), cgrt::vec3(0.5f)};
                Ray e = rays[i];
                (void)intersectRayWithShape(box, e);
                Sphere sp{cgrt::vec3(0.0f), 0.5f, Material{}};
                Ray f = rays[i];
                HitInfo hf;
                (void)intersectRayWithShape(sp, f, hf);
            }
        }
        return bad;
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

// The reference calls BoundingVolumeHierarchy::intersect on ONE const object from the threads of an omp parallel for
// (main.cpp:653-656).  nthreads std::threads do the same here, ray i on thread i % nthreads, one ray per call, and every
// result is compared bit for bit with intersectBatch on the same rays.  timing (optional, 3 doubles): microseconds per
// per-ray call on one thread, microseconds per call per thread with nthreads threads, aggregate calls per second, then the
// library's call-combining counters over the whole test (cgrt_debug_combiner_stats): timing has 8 slots.
// Returns the number of disagreements (0 = pass), -1 on an exception.
int cgrt_host_threads_test(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                           const float* materials, uint32_t nmesh, const float* rays7, uint32_t nrays, int nthreads, double* timing) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, nullptr, 0);
        const BoundingVolumeHierarchy bvh(&sc);
        std::vector<Ray> rays(nrays);
        for (uint32_t i = 0; i < nrays; i++) {
            const float* p = rays7 + 7 * i;
            rays[i].origin = cgrt::vec3(p[0], p[1], p[2]);
            rays[i].direction = cgrt::vec3(p[3], p[4], p[5]);
            rays[i].t = p[6];
        }
        std::vector<Ray> batch = rays;
        std::vector<HitInfo> bhi(nrays);
        std::vector<uint8_t> bhit(nrays);
        bvh.intersectBatch(batch.data(), bhi.data(), bhit.data(), nrays);
        using Clk = std::chrono::steady_clock;
        // one thread, per-ray calls (also warms the library's call lanes)
        const uint32_t nsingle = nrays < 2000 ? nrays : 2000;
        const auto s0 = Clk::now();
        for (uint32_t i = 0; i < nsingle; i++) {
            Ray r = rays[i];
            HitInfo hi;
            (void)bvh.intersect(r, hi);
        }
        const double single_us = std::chrono::duration<double, std::micro>(Clk::now() - s0).count() / (nsingle ? nsingle : 1);
        std::vector<Ray> tr = rays;
        std::vector<HitInfo> thi(nrays);
        std::vector<uint8_t> thit(nrays);
        uint64_t cs0[5] = {0, 0, 0, 0, 0};
        (void)cgrt_debug_combiner_stats(bvh.handle(), cs0);
        const auto t0 = Clk::now();
        {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; t++)
                pool.emplace_back([&, t] {
                    for (uint32_t i = (uint32_t)t; i < nrays; i += (uint32_t)nthreads) thit[i] = bvh.intersect(tr[i], thi[i]) ? 1 : 0;
                });
            for (std::thread& th : pool) th.join();
        }
        const double wall_us = std::chrono::duration<double, std::micro>(Clk::now() - t0).count();
        int bad = 0;
        for (uint32_t i = 0; i < nrays; i++) {
            bad += thit[i] != bhit[i];
            bad += std::memcmp(&tr[i].t, &batch[i].t, 4) != 0;
            if (bhit[i]) {
                bad += std::memcmp(&thi[i].normal, &bhi[i].normal, 12) != 0;
                bad += std::memcmp(&thi[i].material, &bhi[i].material, sizeof(Material)) != 0;
            }
        }
        if (timing) {
            timing[0] = single_us;
            timing[1] = wall_us * nthreads / (nrays ? nrays : 1);
            timing[2] = nrays / (wall_us * 1e-6);
            uint64_t cs[5] = {0, 0, 0, 0, 0};  // call combining inside the library: generations, rays, largest generation, leaders' GPU ns, launch ns
            (void)cgrt_debug_combiner_stats(bvh.handle(), cs);
            timing[3] = (double)(cs[0] - cs0[0]);  // (the threaded part only)
            timing[4] = (double)(cs[1] - cs0[1]);
            timing[5] = (double)cs[2];
            timing[6] = (double)(cs[3] - cs0[3]);
            timing[7] = (double)(cs[4] - cs0[4]);
        }
        return bad;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}

// Loader -> scene -> device render -> Screen -> BMP, end to end: `nreplicas` BVH replicas on device 0 (> 1: the frame is
// split over them like over that many GPUs, renderRayTracingOnDevices), the frame goes through Screen::setPixel (y flip,
// screen.cpp:30-36) and writeBitmapToFile (:38-49).  rgb_out (optional): the float frame before Screen, index y*W+x.
int cgrt_host_render_bmp(const float* pos_nrm, uint32_t nverts, const uint32_t* tri, const uint32_t* tri_mesh, uint32_t ntris,
                         const float* materials, uint32_t nmesh, const float* lights, uint32_t nlights, const float* cam, int W, int H,
                         int maxLevel, int nreplicas, const char* path, float* rgb_out) {
    try {
        Scene sc = scene_from_arrays(pos_nrm, nverts, tri, tri_mesh, ntris, materials, nmesh, lights, nlights);
        std::vector<std::unique_ptr<BoundingVolumeHierarchy>> own;
        std::vector<const BoundingVolumeHierarchy*> bvhs;
        for (int i = 0; i < (nreplicas < 1 ? 1 : nreplicas); i++) {
            own.emplace_back(new BoundingVolumeHierarchy(&sc, 0));
            bvhs.push_back(own.back().get());
        }
        Trackball camera(cam[7], cam[8], cam[6]);
        camera.setCamera(cgrt::vec3(cam[0], cam[1], cam[2]), cgrt::vec3(cam[3], cam[4], cam[5]), cam[6]);
        Screen screen(W, H);
        if (rgb_out) (void)renderToBufferOnDevices(sc, camera, bvhs, W, H, rgb_out, maxLevel);
        (void)renderRayTracingOnDevices(sc, camera, bvhs, screen, maxLevel);
        screen.writeBitmapToFile(path);
        return 0;
    } catch (const std::exception& e) {
        g_err = e.what();
        return -1;
    }
}
}
