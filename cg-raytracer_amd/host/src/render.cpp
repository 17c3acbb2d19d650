// Wavefront restatement of the reference's render driver (src/main.cpp:61-310, :648-720).  See render.h.
#include "render.h"

#include "../../../include/cgrt.h"

#include <omp.h>

#include "host_threads.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <limits>
#include <mutex>
#include <new>
#include <random>
#include <stdexcept>
#include <string>
#include <type_traits>
#include <vector>

namespace {
using cgrt::vec3;
using Clock = std::chrono::steady_clock;

// main.cpp:84-98
vec3 diffuseOneLight(const PointLight& light, const vec3& fromPosToLight, const HitInfo& hitInfo) {
    const float diffuseCos = cgrt::dot(fromPosToLight, hitInfo.normal);
    if (diffuseCos <= 0) return vec3(0.0f);
    return light.color * hitInfo.material.kd * diffuseCos;
}
// main.cpp:61-82 (pow(float, float) must yield float for `vec3 * pow(...)` to compile upstream: powf)
vec3 specularOneLight(const Ray& ray, const PointLight& light, const vec3& fromPosToLight, const HitInfo& hitInfo) {
    const vec3 reflected = cgrt::normalize(cgrt::reflect(ray.direction, hitInfo.normal));
    const float specularCos = cgrt::dot(reflected, fromPosToLight);
    if (specularCos <= 0) return vec3(0.0f);
    return light.color * hitInfo.material.ks * std::pow(specularCos, hitInfo.material.shininess);
}

struct Path {       // one live ray of the current level
    uint32_t pixel; // index into the level-0 arrays
    uint32_t parent;  // index of the item of the previous level that spawned it
};
struct LevelItem {  // what the backward pass needs: color = direct + reflectedColor * ks  (main.cpp:262)
    vec3 direct;
    vec3 ks;
    bool hit;
    int child;  // item of the next level, -1 if none
    uint32_t parent;
};
uint32_t mix32(uint32_t h) {  // murmur3's 32-bit finaliser
    h ^= h >> 16;
    h *= 0x85ebca6bu;
    h ^= h >> 13;
    h *= 0xc2b2ae35u;
    h ^= h >> 16;
    return h;
}
}  // namespace

SoftShadowSampler SoftShadowSampler::gaussian(uint32_t n, uint32_t engineSeed, uint32_t samples, uint32_t seed) {
    SoftShadowSampler s;
    s.samples = samples;
    s.seed = seed;
    s.units.reserve(n);
    std::default_random_engine generator(engineSeed);
    std::normal_distribution<float> distribution(0.0f, 1.0f);
    for (uint32_t i = 0; i < n; i++) {
        const float y = distribution(generator);
        const float x = distribution(generator);
        const float z = distribution(generator);
        s.units.push_back(cgrt::normalize(vec3(y, x, z)));
    }
    return s;
}
const cgrt::vec3& SoftShadowSampler::draw(uint32_t pixel, uint32_t level, uint32_t light, uint32_t smp) const {
    uint32_t h = mix32(seed ^ 0x9e3779b9u);
    h = mix32(h ^ pixel);
    h = mix32(h ^ (level * 0x01000193u + light));
    h = mix32(h ^ smp);
    return units[h % (uint32_t)units.size()];
}

namespace {
// The driver's large arrays come from a cache of blocks that survives the call: a level of a 1080p frame holds ~300 MB of rays,
// HitInfos and items, and handing them back to the C library means munmap + fresh page faults on every frame (measured: 35 of
// the 60 ms of a Cornell frame).  Best fit with at most 50 % slack; at most kCacheLimit bytes are kept (oldest blocks go first);
// releaseRenderBuffers() empties it.
class BlockCache {
public:
    static constexpr size_t kCacheLimit = size_t(1) << 30, kSmall = size_t(1) << 16;
    void* take(size_t need, size_t& cap) {
        if (need >= kSmall) {
            std::lock_guard<std::mutex> lk(m_);
            size_t best = blocks_.size();
            for (size_t i = 0; i < blocks_.size(); i++)
                if (blocks_[i].cap >= need && blocks_[i].cap <= need + need / 2 && (best == blocks_.size() || blocks_[i].cap < blocks_[best].cap)) best = i;
            if (best != blocks_.size()) {
                void* p = blocks_[best].p;
                cap = blocks_[best].cap;
                bytes_ -= cap;
                blocks_.erase(blocks_.begin() + (std::ptrdiff_t)best);
                return p;
            }
        }
        cap = need;
        void* p = std::malloc(need);
        if (!p) throw std::bad_alloc();
        return p;
    }
    void give(void* p, size_t cap) {
        if (cap < kSmall || cap > kCacheLimit) {
            std::free(p);
            return;
        }
        std::lock_guard<std::mutex> lk(m_);
        while (!blocks_.empty() && bytes_ + cap > kCacheLimit) {
            std::free(blocks_.front().p);
            bytes_ -= blocks_.front().cap;
            blocks_.erase(blocks_.begin());
        }
        blocks_.push_back({p, cap});
        bytes_ += cap;
    }
    void clear() {
        std::lock_guard<std::mutex> lk(m_);
        for (Block& b : blocks_) std::free(b.p);
        blocks_.clear();
        bytes_ = 0;
    }
    ~BlockCache() { clear(); }

private:
    struct Block {
        void* p;
        size_t cap;
    };
    std::mutex m_;
    std::vector<Block> blocks_;
    size_t bytes_ = 0;
};
BlockCache g_blocks;

// An array whose elements the driver's parallel loops write before anything reads them: not value-initialised (a
// std::vector would touch a level's rays and HitInfos on one thread first).
template <class T>
struct Array {
    static_assert(std::is_trivially_copyable<T>::value && std::is_trivially_destructible<T>::value, "plain data only");
    T* p = nullptr;
    size_t n = 0, cap = 0;
    Array() = default;
    explicit Array(size_t count) : n(count) {
        if (count) p = static_cast<T*>(g_blocks.take(count * sizeof(T), cap));
    }
    Array(Array&& o) noexcept : p(o.p), n(o.n), cap(o.cap) { o.p = nullptr, o.n = o.cap = 0; }
    Array& operator=(Array&& o) noexcept {
        if (this != &o) {
            if (p) g_blocks.give(p, cap);
            p = o.p, n = o.n, cap = o.cap;
            o.p = nullptr, o.n = o.cap = 0;
        }
        return *this;
    }
    Array(const Array&) = delete;
    Array& operator=(const Array&) = delete;
    ~Array() {
        if (p) g_blocks.give(p, cap);
    }
    T& operator[](size_t i) { return p[i]; }
    const T& operator[](size_t i) const { return p[i]; }
    size_t size() const { return n; }
    bool empty() const { return n == 0; }
    T* data() { return p; }
};

// off[i] = number of set flags before i; returns their total.  The lists of a level keep the order a sequential loop over the
// items would give them, whichever thread writes an entry.
size_t scanFlags(const uint8_t* flag, size_t n, uint32_t* off) {
    std::vector<size_t> part((size_t)cgrt::hostTeam() + 1, 0);
#pragma omp parallel num_threads(cgrt::hostTeam())
    {
        const size_t T = (size_t)omp_get_num_threads(), t = (size_t)omp_get_thread_num();
        const size_t b = n * t / T, e = n * (t + 1) / T;
        size_t c = 0;
        for (size_t i = b; i < e; i++) c += flag[i] != 0;
        part[t + 1] = c;
#pragma omp barrier
#pragma omp single
        for (size_t k = 0; k < T; k++) part[k + 1] += part[k];
        size_t o = part[t];
        for (size_t i = b; i < e; i++) {
            off[i] = (uint32_t)o;
            o += flag[i] != 0;
        }
    }
    size_t total = 0;
    for (size_t v : part) total = std::max(total, v);
    return total;
}
}  // namespace

void releaseRenderBuffers() { g_blocks.clear(); }

// The loops over a level's items are `omp parallel for`s, as the reference's loop over rows is (main.cpp:653-656): every item writes
// only its own entries, and list positions come from prefix sums, so the frame does not depend on the thread count.
RenderStats renderToBuffer(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb,
                           int maxLevel, const SoftShadowSampler* sampler) {
    RenderStats st;
    SoftShadowSampler fallback;
    if (!scene.sphericalLight.empty() && (!sampler || sampler->units.empty() || sampler->samples == 0)) {
        fallback = SoftShadowSampler::gaussian();
        sampler = &fallback;
    }
    const auto t_begin = Clock::now();
    const size_t npix = (size_t)W * H;
    const float eps = 0.001;  // main.cpp:110, :255
    const HitInfo noHit{};
    Array<Ray> rays(npix);
    Array<HitInfo> his(npix);
    Array<uint8_t> hit(npix);
    Array<uint32_t> parent(npix), pixel(npix);
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
    for (size_t i = 0; i < npix; i++) {
        parent[i] = pixel[i] = (uint32_t)i;
        his[i] = noHit;
        hit[i] = 0;
    }
    std::vector<Array<LevelItem>> levels;
    if (maxLevel >= 1) {  // trace(0): `level >= maxLevel` -> black without tracing (main.cpp:267)
        auto t0 = Clock::now();
        bvh.tracePrimary(camera.abi(), W, H, rays.data(), his.data(), hit.data());
        st.seconds_device += std::chrono::duration<double>(Clock::now() - t0).count();
        st.primary = npix;
    }
    size_t n = npix;
    for (int level = 0; level < maxLevel && n; level++) {
        const size_t L = scene.pointLights.size();
        Array<LevelItem> items(n);
        Array<uint32_t> hoff(n);  // hits before item i
        const size_t nh = scanFlags(hit.data(), n, hoff.data());
        // ---- shadow rays of every hit x light (pointInShadow, main.cpp:104-135) ----
        Array<Ray> srays(nh * L);
        Array<float> sdist(nh * L);
        Array<HitInfo> shi(nh * L);
        Array<uint8_t> shit(nh * L);
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
        for (size_t i = 0; i < n; i++) {
            if (!hit[i]) continue;
            const vec3 pointOn = rays[i].origin + rays[i].direction * rays[i].t;
            for (size_t l = 0; l < L; l++) {
                const size_t q = (size_t)hoff[i] * L + l;
                const vec3 fromPosToLight = scene.pointLights[l].position - pointOn;
                Ray r{pointOn, cgrt::normalize(fromPosToLight), std::numeric_limits<float>::max()};
                r.origin = r.origin + eps * r.direction;
                srays[q] = r;
                sdist[q] = cgrt::length(fromPosToLight);
                shi[q] = noHit;
                shit[q] = 0;
            }
        }
        if (nh && L) {
            auto t0 = Clock::now();
            bvh.intersectBatch(srays.data(), shi.data(), shit.data(), srays.size());
            st.seconds_device += std::chrono::duration<double>(Clock::now() - t0).count();
            st.shadow += srays.size();
        }
        // ---- soft shadows: `samples` rays per hit x spherical light (main.cpp:168-200), in bounded batches ----
        const size_t SL = scene.sphericalLight.size();
        std::vector<uint32_t> litCount(SL ? n * SL : 0, 0);
        if (SL) {
            const uint32_t S = sampler->samples;
            const size_t chunkItems = std::max<size_t>(1, (size_t(1) << 22) / (SL * S));  // ~4 M rays per batch
            for (size_t i0 = 0; i0 < n; i0 += chunkItems) {
                const size_t i1 = std::min(n, i0 + chunkItems);
                const size_t h0 = hoff[i0], h1 = i1 < n ? (size_t)hoff[i1] : nh;
                const size_t nq = (h1 - h0) * SL * S;
                if (!nq) continue;
                Array<Ray> qr(nq);
                Array<float> lightT(nq);
                Array<HitInfo> qh(nq);
                Array<uint8_t> qhit(nq);
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
                for (size_t i = i0; i < i1; i++) {
                    if (!hit[i]) continue;
                    const vec3 pointOn = rays[i].origin + rays[i].direction * rays[i].t;
                    for (size_t l = 0; l < SL; l++) {
                        const SphericalLight& spherical = scene.sphericalLight[l];
                        for (uint32_t k = 0; k < S; k++) {
                            const size_t q = (((size_t)hoff[i] - h0) * SL + l) * S + k;
                            const vec3 randomPointOnSphere = spherical.position + spherical.radius * sampler->draw(pixel[i], (uint32_t)level, (uint32_t)l, k);
                            Ray newRay;  // :178, members in declaration order
                            newRay.origin = pointOn + (float)(0.001) * cgrt::normalize(randomPointOnSphere - pointOn);
                            newRay.direction = cgrt::normalize(randomPointOnSphere - pointOn);
                            newRay.t = cgrt::length(newRay.origin - randomPointOnSphere);
                            qr[q] = newRay;
                            lightT[q] = cgrt::length(newRay.origin - randomPointOnSphere);
                            qh[q] = noHit;
                            qhit[q] = 0;
                        }
                    }
                }
                auto t0 = Clock::now();
                bvh.intersectBatch(qr.data(), qh.data(), qhit.data(), qr.size());
                st.seconds_device += std::chrono::duration<double>(Clock::now() - t0).count();
                st.softShadow += qr.size();
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
                for (size_t i = i0; i < i1; i++) {
                    if (!hit[i]) continue;
                    for (size_t l = 0; l < SL; l++)
                        for (uint32_t k = 0; k < S; k++) {
                            const size_t q = (((size_t)hoff[i] - h0) * SL + l) * S + k;
                            if (!qhit[q] || qr[q].t > lightT[q]) litCount[i * SL + l]++;  // :183-199
                        }
                }
            }
        }
        // ---- direct light (shading, main.cpp:219-232); which items spawn a reflection ray (shade, :241-264) ----
        Array<uint8_t> spawns(n);
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
        for (size_t i = 0; i < n; i++) {
            LevelItem& it = items[i];
            it.hit = hit[i] != 0;
            it.child = -1;
            it.parent = parent[i];
            it.direct = vec3(0.0f);
            it.ks = vec3(0.0f);
            spawns[i] = 0;
            if (!it.hit) continue;
            const vec3 pointOn = rays[i].origin + rays[i].direction * rays[i].t;
            vec3 result(0.0f);
            for (size_t l = 0; l < SL; l++) {  // spherical lights first (:168)
                const PointLight light{scene.sphericalLight[l].position, scene.sphericalLight[l].color};
                const vec3 fromPosToLight = cgrt::normalize(light.position - pointOn);
                const vec3 diffuse = diffuseOneLight(light, fromPosToLight, his[i]);
                const vec3 specular = specularOneLight(rays[i], light, fromPosToLight, his[i]);
                // softShadowCounter: `samples` exact additions of 1.0f, then the division (:200)
                const float softShadowCounter = (float)litCount[i * SL + l] / (float)sampler->samples;
                result += diffuse * softShadowCounter;
                result += specular * softShadowCounter;
            }
            for (size_t l = 0; l < L; l++) {
                const size_t q = (size_t)hoff[i] * L + l;
                const PointLight& light = scene.pointLights[l];
                const vec3 fromPosToLight = cgrt::normalize(light.position - pointOn);
                const bool inShadow = shit[q] && !(srays[q].t + eps >= sdist[q]);  // :118-130
                if (inShadow) continue;
                result += diffuseOneLight(light, fromPosToLight, his[i]);
                result += specularOneLight(rays[i], light, fromPosToLight, his[i]);
            }
            it.direct = result;
            it.ks = his[i].material.ks;
            if (his[i].material.ks.z <= 0.01f) continue;  // :246: the comma operator leaves only ks.z tested
            if (level + 1 >= maxLevel) continue;          // trace(level+1) would return black (:267): color = direct + 0 * ks
            spawns[i] = 1;
        }
        // ---- the reflection rays, in item order ----
        Array<uint32_t> roff(n);
        const size_t nr = scanFlags(spawns.data(), n, roff.data());
        Array<Ray> nrays(nr);
        Array<uint32_t> nparent(nr), npixel(nr);
        Array<HitInfo> nhis(nr);
        Array<uint8_t> nhit(nr);
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
        for (size_t i = 0; i < n; i++) {
            if (!spawns[i]) continue;
            const size_t c = roff[i];
            const vec3 pointOn = rays[i].origin + rays[i].direction * rays[i].t;
            const vec3 reflected = cgrt::normalize(cgrt::reflect(rays[i].direction, his[i].normal));
            Ray rr{pointOn, reflected, cgrt::length(rays[i].direction)};  // :254: t = |direction|, not FLT_MAX
            rr.origin = rr.origin + eps * rr.direction;
            items[i].child = (int)c;
            nrays[c] = rr;
            nparent[c] = (uint32_t)i;
            npixel[c] = pixel[i];
            nhis[c] = noHit;
            nhit[c] = 0;
        }
        levels.push_back(std::move(items));
        rays = std::move(nrays);
        parent = std::move(nparent);
        pixel = std::move(npixel);
        his = std::move(nhis);
        hit = std::move(nhit);
        n = nr;
        if (n) {
            auto t0 = Clock::now();
            bvh.intersectBatch(rays.data(), his.data(), hit.data(), n);
            st.seconds_device += std::chrono::duration<double>(Clock::now() - t0).count();
            st.reflection += n;
        }
    }
    // ---- backward pass: color = directColor + reflectedColor * ks (main.cpp:262), deepest level first ----
    Array<vec3> below;
    for (int level = (int)levels.size() - 1; level >= 0; level--) {
        const Array<LevelItem>& items = levels[level];
        Array<vec3> color(items.size());
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
        for (size_t i = 0; i < items.size(); i++) {
            const LevelItem& it = items[i];
            if (!it.hit)
                color[i] = vec3(0.0f);  // :293
            else if (it.ks.z <= 0.01f)
                color[i] = it.direct;  // :248
            else
                color[i] = it.direct + (it.child >= 0 ? below[it.child] : vec3(0.0f)) * it.ks;
        }
        below = std::move(color);
    }
#pragma omp parallel for schedule(static) num_threads(cgrt::hostTeam())
    for (size_t i = 0; i < npix; i++) {
        const vec3 c = levels.empty() ? vec3(0.0f) : below[i];
        rgb[3 * i] = c.x;
        rgb[3 * i + 1] = c.y;
        rgb[3 * i + 2] = c.z;
    }
    st.seconds_total = std::chrono::duration<double>(Clock::now() - t_begin).count();
    return st;
}

namespace {
// The whole driver on the device(s).  rgb != nullptr: the frame goes into the caller's buffer; screen != nullptr: into the
// Screen (Screen::setPixel's flip, main.cpp:696) -- with one replica straight from the library's pinned frame
// (cgrt_render_mapped: no intermediate copy of the 12-bytes-per-pixel frame), rows copied in bulk (Screen::setFrame).
RenderStats render_on_devices(const Scene& scene, const Trackball& camera, const std::vector<const BoundingVolumeHierarchy*>& bvhs, int W, int H,
                              float* rgb, Screen* screen, int maxLevel, const SoftShadowSampler* sampler) {
    const auto t_begin = Clock::now();
    if (bvhs.empty()) throw std::runtime_error("renderToBufferOnDevices: no BVH replica");
    SoftShadowSampler fallback;
    if (!scene.sphericalLight.empty() && (!sampler || sampler->units.empty() || sampler->samples == 0)) {
        fallback = SoftShadowSampler::gaussian();
        sampler = &fallback;
    }
    std::vector<float> lights, spherical, units;
    for (const PointLight& l : scene.pointLights)
        lights.insert(lights.end(), {l.position.x, l.position.y, l.position.z, l.color.x, l.color.y, l.color.z});
    for (const SphericalLight& l : scene.sphericalLight)
        spherical.insert(spherical.end(), {l.position.x, l.position.y, l.position.z, l.radius, l.color.x, l.color.y, l.color.z});
    CgrtSoftShadows soft{};
    if (!spherical.empty()) {
        for (const vec3& u : sampler->units) units.insert(units.end(), {u.x, u.y, u.z});
        soft.spherical = spherical.data();
        soft.unit_vectors = units.data();
        soft.nspherical = (uint32_t)scene.sphericalLight.size();
        soft.samples = sampler->samples;
        soft.nunits = (uint32_t)sampler->units.size();
        soft.seed = sampler->seed;
        soft.closest_hit = 0;
    }
    CgrtRenderStats cs{};
    const CgrtCamera cam = camera.abi();
    std::vector<CgrtScene*> handles;
    for (const BoundingVolumeHierarchy* b : bvhs) handles.push_back(b->handle());
    const uint32_t L = (uint32_t)scene.pointLights.size();
    const CgrtSoftShadows* sp = spherical.empty() ? nullptr : &soft;
    int rc;
    if (handles.size() == 1 && screen) {
        const float* frame = nullptr;
        rc = cgrt_render_mapped(handles[0], &cam, W, H, lights.data(), L, sp, maxLevel, &frame, &cs);
        if (rc == 0) screen->setFrame(frame);
    } else if (handles.size() == 1) {
        rc = cgrt_render_soft(handles[0], &cam, W, H, lights.data(), L, sp, maxLevel, rgb, &cs);
    } else {
        std::vector<float> tmp;
        if (!rgb) {
            tmp.resize((size_t)W * H * 3);
            rgb = tmp.data();
        }
        rc = cgrt_render_multi(handles.data(), (int)handles.size(), &cam, W, H, lights.data(), L, sp, maxLevel, rgb, &cs);
        if (rc == 0 && screen) screen->setFrame(rgb);
    }
    if (rc != 0) throw std::runtime_error(std::string("cgrt_render: ") + cgrt_last_error());
    RenderStats st;
    st.primary = cs.primary_rays;
    st.shadow = cs.shadow_rays;
    st.reflection = cs.reflection_rays;
    st.softShadow = cs.soft_shadow_rays;
    st.seconds_device = cs.device_ms * 1e-3;
    st.seconds_total = std::chrono::duration<double>(Clock::now() - t_begin).count();
    return st;
}
}  // namespace

RenderStats renderToBufferOnDevices(const Scene& scene, const Trackball& camera, const std::vector<const BoundingVolumeHierarchy*>& bvhs, int W, int H,
                                    float* rgb, int maxLevel, const SoftShadowSampler* sampler) {
    return render_on_devices(scene, camera, bvhs, W, H, rgb, nullptr, maxLevel, sampler);
}

RenderStats renderToBufferOnDevice(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb,
                                   int maxLevel, const SoftShadowSampler* sampler) {
    return render_on_devices(scene, camera, {&bvh}, W, H, rgb, nullptr, maxLevel, sampler);
}

RenderStats renderRayTracingOnDevices(const Scene& scene, const Trackball& camera, const std::vector<const BoundingVolumeHierarchy*>& bvhs,
                                      Screen& screen, int maxLevel, const SoftShadowSampler* sampler) {
    return render_on_devices(scene, camera, bvhs, screen.width(), screen.height(), nullptr, &screen, maxLevel, sampler);
}

RenderStats renderRayTracingOnDevice(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel,
                                     const SoftShadowSampler* sampler) {
    return render_on_devices(scene, camera, {&bvh}, screen.width(), screen.height(), nullptr, &screen, maxLevel, sampler);
}

// ---- the reference's per-pixel recursion, one intersect call per ray (main.cpp:104-135, :160-310, :648-696) ----
namespace {
struct PerRay {
    const Scene& scene;
    const BoundingVolumeHierarchy& bvh;
    const SoftShadowSampler* sampler;
    int maxLevel;
    // pointInShadow, main.cpp:104-135
    bool pointInShadow(const vec3& pointOn, const vec3& fromPosToLight, uint64_t& nshadow) const {
        const float eps = 0.001;
        Ray shadowRay{pointOn, cgrt::normalize(fromPosToLight), std::numeric_limits<float>::max()};
        shadowRay.origin = shadowRay.origin + eps * shadowRay.direction;
        HitInfo tmp;
        nshadow++;
        if (bvh.intersect(shadowRay, tmp)) return !(shadowRay.t + eps >= cgrt::length(fromPosToLight));  // :118-130
        return false;
    }
    // shading, main.cpp:160-235: spherical lights first (soft shadows), then the point lights
    vec3 shading(const Ray& ray, const HitInfo& hitInfo, uint32_t pixel, int level, uint64_t& nshadow, uint64_t& nsoft) const {
        const vec3 pointOn = ray.origin + ray.direction * ray.t;
        vec3 result(0.0f);
        for (size_t l = 0; l < scene.sphericalLight.size(); l++) {
            const SphericalLight& spherical = scene.sphericalLight[l];
            const PointLight light{spherical.position, spherical.color};
            const vec3 fromPosToLight = cgrt::normalize(light.position - pointOn);
            const vec3 diffuse = diffuseOneLight(light, fromPosToLight, hitInfo);
            const vec3 specular = specularOneLight(ray, light, fromPosToLight, hitInfo);
            uint32_t lit = 0;
            for (uint32_t k = 0; k < sampler->samples; k++) {
                const vec3 randomPointOnSphere = spherical.position + spherical.radius * sampler->draw(pixel, (uint32_t)level, (uint32_t)l, k);
                Ray newRay;
                newRay.origin = pointOn + (float)(0.001) * cgrt::normalize(randomPointOnSphere - pointOn);
                newRay.direction = cgrt::normalize(randomPointOnSphere - pointOn);
                newRay.t = cgrt::length(newRay.origin - randomPointOnSphere);
                const float lightT = cgrt::length(newRay.origin - randomPointOnSphere);
                HitInfo hi;
                nsoft++;
                if (!bvh.intersect(newRay, hi) || newRay.t > lightT) lit++;
            }
            const float softShadowCounter = (float)lit / (float)sampler->samples;
            result += diffuse * softShadowCounter;
            result += specular * softShadowCounter;
        }
        for (const PointLight& light : scene.pointLights) {
            const vec3 toLight = light.position - pointOn;
            const vec3 fromPosToLight = cgrt::normalize(toLight);
            if (pointInShadow(pointOn, toLight, nshadow)) continue;
            result += diffuseOneLight(light, fromPosToLight, hitInfo);
            result += specularOneLight(ray, light, fromPosToLight, hitInfo);
        }
        return result;
    }
    // trace + shade, main.cpp:241-295
    vec3 trace(int level, Ray ray, uint32_t pixel, uint64_t& nshadow, uint64_t& nrefl, uint64_t& nsoft) const {
        if (level >= maxLevel) return vec3(0.0f);  // :267
        HitInfo hitInfo;
        if (level > 0) nrefl++;
        if (!bvh.intersect(ray, hitInfo)) return vec3(0.0f);  // :293
        const vec3 color = shading(ray, hitInfo, pixel, level, nshadow, nsoft);
        if (hitInfo.material.ks.z <= 0.01f) return color;  // :246 (the comma operator leaves ks.z)
        if (level + 1 >= maxLevel) return color;            // trace(level + 1) is black: color + 0 * ks
        const float eps = 0.001;
        const vec3 pointOn = ray.origin + ray.direction * ray.t;
        Ray reflected{pointOn, cgrt::normalize(cgrt::reflect(ray.direction, hitInfo.normal)), cgrt::length(ray.direction)};  // :254
        reflected.origin = reflected.origin + eps * reflected.direction;
        const vec3 reflectedColor = trace(level + 1, reflected, pixel, nshadow, nrefl, nsoft);
        return color + reflectedColor * hitInfo.material.ks;  // :262
    }
};
}  // namespace

RenderStats renderToBufferPerRay(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, int W, int H, float* rgb, int maxLevel,
                                 const SoftShadowSampler* sampler, int threads) {
    RenderStats st;
    SoftShadowSampler fallback;
    if (!scene.sphericalLight.empty() && (!sampler || sampler->units.empty() || sampler->samples == 0)) {
        fallback = SoftShadowSampler::gaussian();
        sampler = &fallback;
    }
    const auto t_begin = Clock::now();
    const PerRay drv{scene, bvh, sampler, maxLevel};
    uint64_t nshadow = 0, nrefl = 0, nsoft = 0;
    if (threads <= 0) threads = omp_get_max_threads();
#pragma omp parallel for schedule(dynamic, 1) num_threads(threads) reduction(+ : nshadow, nrefl, nsoft)
    for (int y = 0; y < H; y++) {  // main.cpp:653-656
        for (int x = 0; x < W; x++) {
            const cgrt::vec2 normalizedPixelPos{float(x) / W * 2.0f - 1.0f, float(y) / H * 2.0f - 1.0f};  // :691-693
            const Ray cameraRay = camera.generateRay(normalizedPixelPos);
            const vec3 c = drv.trace(0, cameraRay, (uint32_t)(y * W + x), nshadow, nrefl, nsoft);
            float* p = rgb + 3 * ((size_t)y * W + x);
            p[0] = c.x, p[1] = c.y, p[2] = c.z;
        }
    }
    st.primary = maxLevel >= 1 ? (uint64_t)W * H : 0;
    st.shadow = nshadow;
    st.reflection = nrefl;
    st.softShadow = nsoft;
    st.seconds_total = st.seconds_device = std::chrono::duration<double>(Clock::now() - t_begin).count();
    return st;
}

RenderStats renderRayTracingPerRay(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel,
                                   const SoftShadowSampler* sampler, int threads) {
    const int W = screen.width(), H = screen.height();
    std::vector<float> rgb((size_t)W * H * 3);
    RenderStats st = renderToBufferPerRay(scene, camera, bvh, W, H, rgb.data(), maxLevel, sampler, threads);
    screen.setFrame(rgb.data());  // main.cpp:696 for every pixel
    return st;
}

RenderStats renderRayTracing(const Scene& scene, const Trackball& camera, const BoundingVolumeHierarchy& bvh, Screen& screen, int maxLevel,
                             const SoftShadowSampler* sampler) {
    const int W = screen.width(), H = screen.height();
    Array<float> rgb((size_t)W * H * 3);
    RenderStats st = renderToBuffer(scene, camera, bvh, W, H, rgb.data(), maxLevel, sampler);
    screen.setFrame(rgb.data());  // main.cpp:696 for every pixel
    return st;
}
