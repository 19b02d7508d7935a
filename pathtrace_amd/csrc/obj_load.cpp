// obj_load.cpp -- OBJ/MTL asset loader: LoadObj (obj_loader.go:19-111) over the vendored g3n decoder
// (obj.go:101-226, :405-939) plus the flattening of DecomposeObjects (renderer.go:293-324), restated
// in C++ behind the C ABI (include/ptmi.h, "assets").  Host only; no GPU call in this file.
//
// What is kept from the reference, on purpose:
//   * only the FIRST object/group of the file is used (obj_loader.go:82: object.Objects[0]);
//     a face before any o/g line opens an object "unnamed<line>" (obj.go:578-583)
//   * "g" is treated like "o" (obj.go:449-452)
//   * a usemtl name the material file does not define is a zero material (obj.go:688-692)
//   * when no material file can be parsed every material becomes the light-grey default (obj.go:209-219)
//   * materials library lookup order: mtllib next to the .obj, then <name>.mtl (obj.go:163-206)
//   * normals are normalised (obj_loader.go:68-74), vertices scaled (:76-78), v flipped to 1-v (:92)
//   * glass / accretion-disk detection by material NAME (tracer.go:49, :95) becomes PtMaterial.flags
// Where the reference would panic the loader returns an error instead: a face that is not a triangle
// (obj_loader.go:83 appends face.Vertices raw), a face corner without a normal (:100-102 indexes with
// invINDEX), a face without usemtl ("internal default" is not in the material map -> nil pointer), a
// corner without vt in a file that has vt lines (:88-96), indices out of range.
// Texture FILES are not decoded here: the names of map_Kd / map_Bump are returned and the host decodes
// them (the Go host with its image package, as LoadObj does at :29-57) into the packed texels that
// ptmi_upload_textures takes.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/ptmi.h"

namespace {

struct Mat {                        // obj.go:63-79
    std::string name;
    int illum = 0;
    float opacity = 0, refraction = 0, shininess = 0;
    float ka[3] = {0, 0, 0}, kd[3] = {0, 0, 0}, ks[3] = {0, 0, 0}, ke[3] = {0, 0, 0};
    std::string map_kd, map_bump;
};

struct Face { int v[3], t[3], n[3]; std::string material; };
constexpr int INV = -1;            // invINDEX (obj.go:92)

struct Decoder {
    std::vector<float> vertices, normals, uvs;
    std::vector<std::string> object_names;
    std::vector<std::vector<Face>> object_faces;
    std::map<std::string, int> mat_index;       // name -> slot in mats, in first-seen order
    std::vector<Mat> mats;
    std::string matlib;
    std::vector<std::string> warnings;
    int cur_obj = -1, cur_mat = -1, line = 1;
    std::string err;
};

std::vector<std::string> fields_of(const std::string &line) {   // strings.Fields
    std::vector<std::string> f;
    std::istringstream ss(line);
    std::string w;
    while (ss >> w) f.push_back(w);
    return f;
}

bool parse_f32(const std::string &s, float &out) {             // strconv.ParseFloat(s, 32)
    if (s.empty()) return false;
    char *end = nullptr;
    out = std::strtof(s.c_str(), &end);
    return end && *end == '\0';
}
bool parse_int(const std::string &s, long &out) {              // strconv.ParseInt(s, 10, 32)
    if (s.empty()) return false;
    char *end = nullptr;
    out = std::strtol(s.c_str(), &end, 10);
    return end && *end == '\0' && out >= INT32_MIN && out <= INT32_MAX;
}

int material_slot(Decoder &d, const std::string &name) {       // obj.go:684-692, :770-777
    auto it = d.mat_index.find(name);
    if (it != d.mat_index.end()) return it->second;
    Mat m; m.name = name;
    d.mats.push_back(m);
    d.mat_index[name] = (int)d.mats.size() - 1;
    return (int)d.mats.size() - 1;
}

void new_object(Decoder &d, const std::string &name) {         // obj.go:490-499
    d.object_names.push_back(name);
    d.object_faces.emplace_back();
    d.cur_obj = (int)d.object_names.size() - 1;
}

bool fail(Decoder &d, const std::string &msg) {
    d.err = msg + " in line:" + std::to_string(d.line);
    return false;
}

bool parse_obj_line(Decoder &d, const std::string &line) {     // obj.go:431-478
    std::vector<std::string> f = fields_of(line);
    if (f.empty() || f[0][0] == '#') return true;
    const std::string &t = f[0];
    if (t == "mtllib") { if (f.size() < 2) return fail(d, "Material library (mtllib) with no fields"); d.matlib = f[1]; return true; }
    if (t == "o" || t == "g") { if (f.size() < 2) return fail(d, "Object line (o) with no fields"); new_object(d, f[1]); return true; }
    if (t == "v" || t == "vn") {
        if (f.size() < 4) return fail(d, t == "v" ? "Less than 3 vertices in 'v' line" : "Less than 3 normals in 'vn' line");
        for (int k = 1; k <= 3; ++k) { float x; if (!parse_f32(f[k], x)) return fail(d, "bad number"); (t == "v" ? d.vertices : d.normals).push_back(x); }
        return true;
    }
    if (t == "vt") {
        if (f.size() < 3) return fail(d, "Less than 2 texture coords. in 'vt' line");
        for (int k = 1; k <= 2; ++k) { float x; if (!parse_f32(f[k], x)) return fail(d, "bad number"); d.uvs.push_back(x); }
        return true;
    }
    if (t == "usemtl") {                                       // obj.go:671-698
        if (f.size() < 2) return fail(d, "Usemtl with no fields");
        if (d.cur_obj < 0) new_object(d, "unnamed" + std::to_string(d.line));
        d.cur_mat = material_slot(d, f[1]);
        return true;
    }
    if (t == "s") return true;                                 // smoothing groups do not reach the renderer
    if (t == "f") {                                            // obj.go:572-668
        if (d.cur_obj < 0) new_object(d, "unnamed" + std::to_string(d.line));
        if (f.size() < 4) return fail(d, "Face line with less 3 fields");
        if (f.size() != 4) return fail(d, "face is not a triangle (the reference appends face.Vertices raw, obj_loader.go:83)");
        Face face;
        face.material = d.cur_mat >= 0 ? d.mats[d.cur_mat].name : "internal default";
        for (int pos = 0; pos < 3; ++pos) {
            std::vector<std::string> vf;
            { std::string part; std::istringstream ss(f[pos + 1]); while (std::getline(ss, part, '/')) vf.push_back(part); if (!f[pos + 1].empty() && f[pos + 1].back() == '/') vf.push_back(""); }
            long val;
            if (vf.empty() || !parse_int(vf[0], val)) return fail(d, "bad face vertex index");
            if (val > 0) face.v[pos] = (int)val - 1;
            else if (val < 0) face.v[pos] = (int)(d.vertices.size() / 3) - 1 + (int)val + 1;
            else return fail(d, "Face vertex index value equal to 0");
            if (vf.size() > 1 && !vf[1].empty()) {
                if (!parse_int(vf[1], val)) return fail(d, "bad face uv index");
                if (val > 0) face.t[pos] = (int)val - 1;
                else if (val < 0) face.t[pos] = (int)(d.uvs.size() / 2) - 1 + (int)val + 1;
                else return fail(d, "Face uv index value equal to 0");
            } else face.t[pos] = INV;
            if (vf.size() >= 3) {
                if (!parse_int(vf[2], val)) return fail(d, "bad face normal index");
                if (val > 0) face.n[pos] = (int)val - 1;
                else if (val < 0) face.n[pos] = (int)(d.normals.size() / 3) - 1 + (int)val + 1;
                else return fail(d, "Face normal index value equal to 0");
            } else face.n[pos] = INV;
        }
        d.object_faces[d.cur_obj].push_back(face);
        return true;
    }
    d.warnings.push_back("obj(" + std::to_string(d.line) + "): field not supported: " + t);
    return true;
}

bool parse_mtl_line(Decoder &d, const std::string &line) {     // obj.go:721-758
    std::vector<std::string> f = fields_of(line);
    if (f.empty() || f[0][0] == '#') return true;
    const std::string &t = f[0];
    if (t == "newmtl") { if (f.size() < 2) return fail(d, "newmtl with no fields"); d.cur_mat = material_slot(d, f[1]); return true; }
    auto known = t == "d" || t == "Ka" || t == "Kd" || t == "Ke" || t == "Ks" || t == "Ni" || t == "Ns" || t == "illum" ||
                 t == "map_Kd" || t == "map_Bump";
    if (!known) { d.warnings.push_back("mtl(" + std::to_string(d.line) + "): field not supported: " + t); return true; }
    if (d.cur_mat < 0) return fail(d, "'" + t + "' before newmtl");     // the reference dereferences a nil matCurrent here
    Mat &m = d.mats[d.cur_mat];
    if (t == "Ka" || t == "Kd" || t == "Ke" || t == "Ks") {
        if (f.size() < 4) return fail(d, "'" + t + "' with less than 3 fields");
        float c[3];
        for (int k = 0; k < 3; ++k) if (!parse_f32(f[k + 1], c[k])) return fail(d, "bad number");
        float *dst = t == "Ka" ? m.ka : t == "Kd" ? m.kd : t == "Ke" ? m.ke : m.ks;
        dst[0] = c[0]; dst[1] = c[1]; dst[2] = c[2];
        return true;
    }
    if (f.size() < 2) return fail(d, "'" + t + "' with no fields");
    if (t == "d" || t == "Ni" || t == "Ns") {
        float x; if (!parse_f32(f[1], x)) return fail(d, "'d' parse float error");
        (t == "d" ? m.opacity : t == "Ni" ? m.refraction : m.shininess) = x;
        return true;
    }
    if (t == "illum") { long v; if (!parse_int(f[1], v) || v < 0) return fail(d, "'d' parse int error"); m.illum = (int)v; return true; }
    if (t == "map_Kd") { m.map_kd = f[1]; return true; }                 // first field (obj.go:917)
    m.map_bump = f.back();                                                // last field  (obj.go:929)
    return true;
}

bool parse_file(Decoder &d, const std::string &path, bool (*fn)(Decoder &, const std::string &)) {   // obj.go:407-429
    std::ifstream in(path, std::ios::binary);
    if (!in) { d.err = "cannot open " + path; return false; }
    d.line = 1;
    std::string line;
    while (std::getline(in, line)) {
        size_t a = line.find_first_not_of("\r\n\t "), b = line.find_last_not_of("\r\n\t ");
        line = a == std::string::npos ? std::string() : line.substr(a, b - a + 1);
        if (!fn(d, line)) return false;
        d.line++;
    }
    return true;
}

std::string dir_of(const std::string &p) {
    size_t s = p.find_last_of('/');
    return s == std::string::npos ? std::string(".") : (s == 0 ? std::string("/") : p.substr(0, s));
}

}  // namespace

struct ptmi_obj {
    std::vector<float> verts, normals, uvs;
    std::vector<uint32_t> tri_idx, tri_material;
    std::vector<PtMaterial> materials;
    std::vector<std::string> names, map_kd, map_bump, warnings;
    std::string mtl_dir;
};

static thread_local std::string g_obj_error;
extern "C" const char *ptmi_obj_last_error(void) { return g_obj_error.c_str(); }

extern "C" int ptmi_obj_load(const char *path, float scale, const float position[3], ptmi_obj **out) {
    if (!out) { g_obj_error = "ptmi_obj_load: out is null"; return PTMI_ERR_INVALID; }
    *out = nullptr;
    if (!path) { g_obj_error = "ptmi_obj_load: path is null"; return PTMI_ERR_INVALID; }
    Decoder d;
    if (!parse_file(d, path, parse_obj_line)) { g_obj_error = "ptmi_obj_load: " + d.err; return PTMI_ERR_INVALID; }
    // material file: mtllib next to the .obj, then <name>.mtl, then the default material (obj.go:163-219)
    const std::string objpath(path), objdir = dir_of(objpath);
    bool mtl_ok = false;
    d.cur_mat = -1;
    const std::vector<Mat> before = d.mats;
    const std::map<std::string, int> before_index = d.mat_index;
    if (!d.matlib.empty()) mtl_ok = parse_file(d, objdir + "/" + d.matlib, parse_mtl_line);
    if (!mtl_ok) {
        d.mats = before; d.mat_index = before_index; d.cur_mat = -1;
        std::string stem = objpath;
        if (stem.size() >= 4 && stem.compare(stem.size() - 4, 4, ".obj") == 0) stem.erase(stem.size() - 4);
        mtl_ok = parse_file(d, stem + ".mtl", parse_mtl_line);
        if (mtl_ok) d.warnings.push_back("mtl: using material file " + stem + ".mtl");
    }
    bool default_materials = false;
    if (!mtl_ok) {
        d.mats = before; d.mat_index = before_index;
        default_materials = true;          // every entry IS defaultMat, whose Name is "" (no glass / disk detection)
        for (Mat &m : d.mats) {                                 // defaultMat (obj.go:82-87)
            std::string n = m.name; m = Mat(); m.name = n;
            m.kd[0] = m.kd[1] = m.kd[2] = 0.7f; m.ka[0] = m.ka[1] = m.ka[2] = 0.7f;
            m.ks[0] = m.ks[1] = m.ks[2] = 0.5f; m.shininess = 30.0f;
        }
        d.warnings.push_back("mtl: unable to parse a material file for obj. using default material instead.");
    }
    if (d.object_faces.empty()) { g_obj_error = "ptmi_obj_load: the file has no object"; return PTMI_ERR_INVALID; }

    ptmi_obj *o = new ptmi_obj();
    o->mtl_dir = objdir;
    o->warnings = d.warnings;
    const float px = position ? position[0] : 0.0f, py = position ? position[1] : 0.0f, pz = position ? position[2] : 0.0f;
    const size_t nv = d.vertices.size() / 3, nn = d.normals.size() / 3, nt = d.uvs.size() / 2;
    o->verts.resize(nv * 3);
    for (size_t i = 0; i < nv; ++i) {                           // obj_loader.go:76-78, then renderer.go:301-303
        o->verts[3 * i] = d.vertices[3 * i] * scale + px;
        o->verts[3 * i + 1] = d.vertices[3 * i + 1] * scale + py;
        o->verts[3 * i + 2] = d.vertices[3 * i + 2] * scale + pz;
    }
    std::vector<float> on(nn * 3);
    for (size_t i = 0; i < nn; ++i) {                           // obj_loader.go:68-74 (vec3.go:60-66)
        const float x = d.normals[3 * i], y = d.normals[3 * i + 1], z = d.normals[3 * i + 2];
        const float len = std::sqrt((x * x + y * y) + z * z);
        if (len == 0.0f) { on[3 * i] = on[3 * i + 1] = on[3 * i + 2] = 0.0f; }
        else { const float inv = 1.0f / len; on[3 * i] = x * inv; on[3 * i + 1] = y * inv; on[3 * i + 2] = z * inv; }
    }
    auto bail = [&](const std::string &m) { g_obj_error = "ptmi_obj_load: " + m; delete o; return PTMI_ERR_INVALID; };
    for (const Face &f : d.object_faces[0]) {                   // obj_loader.go:82-103: Objects[0] only
        auto mi = d.mat_index.find(f.material);
        if (mi == d.mat_index.end()) return bail("face without usemtl (material \"" + f.material + "\")");
        for (int k = 0; k < 3; ++k) {
            if (f.v[k] < 0 || (size_t)f.v[k] >= nv) return bail("face vertex index out of range");
            if (f.n[k] == INV) return bail("face corner without a normal");
            if (f.n[k] < 0 || (size_t)f.n[k] >= nn) return bail("face normal index out of range");
            o->tri_idx.push_back((uint32_t)f.v[k]);
            o->normals.insert(o->normals.end(), on.begin() + 3 * f.n[k], on.begin() + 3 * f.n[k] + 3);
            if (nt > 0) {
                if (f.t[k] == INV) return bail("face corner without vt in a file that has vt lines");
                if (f.t[k] < 0 || (size_t)f.t[k] >= nt) return bail("face uv index out of range");
                o->uvs.push_back(d.uvs[2 * f.t[k]]);
                o->uvs.push_back(1.0f - d.uvs[2 * f.t[k] + 1]);  // :92
            }
        }
        o->tri_material.push_back((uint32_t)mi->second);
    }
    for (const Mat &m : d.mats) {
        PtMaterial pm;
        std::memset(&pm, 0, sizeof pm);
        std::memcpy(pm.kd, m.kd, 12); std::memcpy(pm.ks, m.ks, 12); std::memcpy(pm.ke, m.ke, 12);
        pm.ni = m.refraction; pm.ns = m.shininess;
        if (!default_materials && m.name.compare(0, 5, "Glass") == 0) pm.flags |= PTMI_MAT_GLASS;      // tracer.go:49
        if (!default_materials && m.name == "AccretionDisk") pm.flags |= PTMI_MAT_ACCRETION;           // tracer.go:95
        o->materials.push_back(pm);
        o->names.push_back(m.name); o->map_kd.push_back(m.map_kd); o->map_bump.push_back(m.map_bump);
    }
    *out = o;
    return PTMI_OK;
}

extern "C" void ptmi_obj_free(ptmi_obj *o) { delete o; }

extern "C" int ptmi_obj_sizes(const ptmi_obj *o, uint64_t *nverts, uint64_t *ntris, uint64_t *nmaterials, int32_t *has_uvs,
                              uint64_t *nwarnings) {
    if (!o) { g_obj_error = "ptmi_obj_sizes: null object"; return PTMI_ERR_INVALID; }
    if (nverts) *nverts = o->verts.size() / 3;
    if (ntris) *ntris = o->tri_idx.size() / 3;
    if (nmaterials) *nmaterials = o->materials.size();
    if (has_uvs) *has_uvs = o->uvs.empty() ? 0 : 1;
    if (nwarnings) *nwarnings = o->warnings.size();
    return PTMI_OK;
}

extern "C" int ptmi_obj_get(const ptmi_obj *o, float *verts_xyz, uint32_t *tri_idx, float *corner_normals,
                            uint32_t *tri_material, PtMaterial *materials, float *uvs) {
    if (!o) { g_obj_error = "ptmi_obj_get: null object"; return PTMI_ERR_INVALID; }
    if (verts_xyz && !o->verts.empty()) std::memcpy(verts_xyz, o->verts.data(), o->verts.size() * 4);
    if (tri_idx && !o->tri_idx.empty()) std::memcpy(tri_idx, o->tri_idx.data(), o->tri_idx.size() * 4);
    if (corner_normals && !o->normals.empty()) std::memcpy(corner_normals, o->normals.data(), o->normals.size() * 4);
    if (tri_material && !o->tri_material.empty()) std::memcpy(tri_material, o->tri_material.data(), o->tri_material.size() * 4);
    if (materials && !o->materials.empty()) std::memcpy(materials, o->materials.data(), o->materials.size() * sizeof(PtMaterial));
    if (uvs && !o->uvs.empty()) std::memcpy(uvs, o->uvs.data(), o->uvs.size() * 4);
    return PTMI_OK;
}

// which: 0 material name, 1 map_Kd, 2 map_Bump, 3 warning text (index = warning number), 4 directory of the .obj
extern "C" const char *ptmi_obj_string(const ptmi_obj *o, int which, uint64_t index) {
    if (!o) return "";
    const std::vector<std::string> *v = which == 0 ? &o->names : which == 1 ? &o->map_kd : which == 2 ? &o->map_bump :
                                        which == 3 ? &o->warnings : nullptr;
    if (which == 4) return o->mtl_dir.c_str();
    if (!v || index >= v->size()) return "";
    return (*v)[index].c_str();
}
