// loadMesh / centerAndScaleToUnitMesh (src/mesh.cpp:58-166) without assimp.  See mesh.h for the semantics kept.
#include "mesh.h"

#include <algorithm>
#include <cmath>
#include <fstream>
#include <iostream>
#include <map>
#include <sstream>
#include <string>

namespace {
struct Corner {
    int v, n;
};
struct RunFaces {
    std::string material;
    std::vector<std::vector<Corner>> faces;
};
std::string rest_of_line(std::istringstream& ss) {
    std::string s;
    std::getline(ss, s);
    const size_t a = s.find_first_not_of(" \t\r"), b = s.find_last_not_of(" \t\r");
    return a == std::string::npos ? std::string() : s.substr(a, b - a + 1);
}
std::map<std::string, Material> parse_mtl(const std::filesystem::path& file) {
    std::map<std::string, Material> out;
    std::ifstream f(file);
    std::string line;
    Material* cur = nullptr;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k;
        if (!(ss >> k) || k[0] == '#') continue;
        if (k == "newmtl") {
            Material m;  // assimp ObjFile::Material defaults: Kd 0.6, Ks 0, Ns 0, d 1
            m.kd = cgrt::vec3(0.6f);
            m.ks = cgrt::vec3(0.0f);
            m.shininess = 0.0f;
            m.transparency = 1.0f;
            cur = &(out[rest_of_line(ss)] = m);
        } else if (cur) {
            if (k == "Kd")
                ss >> cur->kd.x >> cur->kd.y >> cur->kd.z;
            else if (k == "Ks")
                ss >> cur->ks.x >> cur->ks.y >> cur->ks.z;
            else if (k == "Ns")
                ss >> cur->shininess;
            else if (k == "d")
                ss >> cur->transparency;
        }
    }
    return out;
}
}  // namespace

std::vector<Mesh> loadMesh(const std::filesystem::path& file, bool centerAndNormalize) {
    if (!std::filesystem::exists(file)) {
        std::cerr << "File " << file << " does not exist." << std::endl;
        throw std::exception();  // mesh.cpp:60-63
    }
    std::vector<cgrt::vec3> V, N;
    std::map<std::string, Material> mtl;
    std::vector<std::vector<RunFaces>> objects(1);
    std::string curMat = "__default__";
    std::ifstream f(file);
    std::string line;
    while (std::getline(f, line)) {
        std::istringstream ss(line);
        std::string k;
        if (!(ss >> k) || k[0] == '#') continue;
        if (k == "v") {
            cgrt::vec3 p;
            ss >> p.x >> p.y >> p.z;
            V.push_back(p);
        } else if (k == "vn") {
            cgrt::vec3 p;
            ss >> p.x >> p.y >> p.z;
            N.push_back(p);
        } else if (k == "mtllib") {
            auto m = parse_mtl(file.parent_path() / rest_of_line(ss));
            mtl.insert(m.begin(), m.end());
        } else if (k == "o" || k == "g") {
            objects.emplace_back();  // assimp maps groups onto objects
        } else if (k == "usemtl") {
            curMat = rest_of_line(ss);
        } else if (k == "f") {
            std::vector<Corner> face;
            std::string tok;
            // An index that is not a number, is 0, or points outside the vertices / normals read so far makes the file
            // invalid here: it is refused the way upstream refuses a file its importer cannot load ("Assimp failed to load mesh
            // file", mesh.cpp:66-71, then a throw).  Whether assimp 5.0.1 refuses, repairs or skips each of these cases is
            // UNPINNED (no fixture upstream, assimp absent): the rule is this repo's -- an unresolvable index never becomes a triangle.
            auto index = [&](const std::string& t, size_t count) -> int {
                size_t used = 0;
                long v = 0;
                try {
                    v = std::stol(t, &used);
                } catch (const std::exception&) {
                    used = 0;
                }
                const long idx = v > 0 ? v - 1 : (long)count + v;
                if (used != t.size() || v == 0 || idx < 0 || idx >= (long)count) {
                    std::cerr << "Assimp failed to load mesh file " << file << std::endl;
                    throw std::exception();
                }
                return (int)idx;
            };
            while (ss >> tok) {
                Corner c{0, -1};
                const size_t s1 = tok.find('/');
                c.v = index(tok.substr(0, s1), V.size());
                if (s1 != std::string::npos) {
                    const size_t s2 = tok.find('/', s1 + 1);
                    if (s2 != std::string::npos && s2 + 1 < tok.size()) c.n = index(tok.substr(s2 + 1), N.size());
                }
                face.push_back(c);
            }
            if (face.size() >= 3) {
                auto& obj = objects.back();
                if (obj.empty() || obj.back().material != curMat) obj.push_back(RunFaces{curMat, {}});
                obj.back().faces.push_back(std::move(face));
            }
        }
    }
    std::vector<Mesh> out;
    // mesh.cpp:77-131 pops sibling nodes off a stack: objects come out in reverse file order
    for (auto it = objects.rbegin(); it != objects.rend(); ++it) {
        for (const RunFaces& run : *it) {
            if (run.faces.empty()) continue;
            Mesh mesh;
            for (const auto& face : run.faces) {
                const uint32_t base = (uint32_t)mesh.vertices.size();
                bool hasN = true;
                for (const Corner& c : face) hasN &= c.n >= 0;
                cgrt::vec3 fn(0.0f);
                if (!hasN) {  // aiProcess_GenNormals: flat face normal
                    const cgrt::vec3 n = cgrt::cross(V[face[1].v] - V[face[0].v], V[face[2].v] - V[face[0].v]);
                    const float l = std::sqrt(cgrt::dot(n, n));
                    if (l > 0) fn = n / l;
                }
                for (const Corner& c : face) mesh.vertices.push_back(Vertex{V[c.v], hasN ? N[c.n] : fn});
                for (uint32_t k = 1; k + 1 < face.size(); k++) mesh.triangles.emplace_back(base, base + k, base + k + 1);  // aiProcess_Triangulate (convex fan)
            }
            auto m = mtl.find(run.material);
            if (m != mtl.end()) {
                mesh.material = m->second;
            } else {
                mesh.material.kd = cgrt::vec3(0.6f);
                mesh.material.shininess = 0.0f;
            }
            out.push_back(std::move(mesh));
        }
    }
    if (centerAndNormalize) centerAndScaleToUnitMesh(out);
    return out;
}

void centerAndScaleToUnitMesh(std::vector<Mesh>& meshes) {  // mesh.cpp:143-166
    cgrt::vec3 sum(0.0f);
    size_t n = 0;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices) {
            sum = sum + v.p;  // std::accumulate, sequential float adds
            n++;
        }
    const cgrt::vec3 center = sum / static_cast<float>(n);
    float maxD = 0.0f;
    for (const Mesh& m : meshes)
        for (const Vertex& v : m.vertices) maxD = std::max(cgrt::length(v.p - center), maxD);
    for (Mesh& m : meshes)
        for (Vertex& v : m.vertices) v.p = (v.p - center) / maxD;
}
