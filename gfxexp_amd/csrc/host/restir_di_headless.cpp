// restir_di_headless.cpp -- a windowless main for the renderers behind include/gfxexp_host.h that takes the command line of
// the reference's sample programs (restir_di/restir_di_main.cpp:1-26, parseCommandline :593-878; the path tracing, ReGIR
// and NRC samples parse the same options) and writes images instead of opening a window.
//
//   restir_di_headless -cam-pos -0.753442 0.140257 -0.056083 -cam-yaw 75
//       -name exterior -obj Amazon_Bistro/Exterior/exterior.obj 0.001 trad -brightness 2.0
//       -name rectlight -emittance 5 5 5 -rectangle 0.1 0.1
//       -inst exterior
//       -begin-pos 0.362 0.329 -2.0 -begin-pitch -90 -begin-yaw 150
//       -end-pos -0.719 0.329 -0.442 -end-pitch -90 -end-yaw 30 -inst rectlight
//       -frames 64 -size 1920 1080 -out frame.exr
//
// Scene options (same names, same arity, same accumulate-until-"-inst" state machine as parseCommandline):
//   -cam-pos x y z   -cam-roll|-cam-pitch|-cam-yaw deg   -brightness b   -env-texture file.exr|.pfm
//   -name n   -emittance r g b   -rect-emitter-tex file   -obj path preScale trad|simple_pbr   -rectangle dimX dimZ
//   -begin-pos|-end-pos x y z   -begin-roll|pitch|yaw  -end-roll|pitch|yaw deg   -begin-scale|-end-scale s   -freq f   -time t
//   -inst n
// Headless options (no counterpart; the reference takes these from its GUI):
//   -size W H (1920 1080)   -frames N (1)   -renderer restir-biased|restir-unbiased|rearch-biased|rearch-unbiased|pt|regir|nrc
//   -animate (advance the instance controllers by 1/60 s per frame, :2249-2257)   -accumulate   -bump   -device k
//   -out path (.exr / .pfm: HDR; .bmp / .ppm: tone-mapped SDR)   -dry-run (parse, build the scene on the host, print it, no GPU)
// Neural radiance caching (-renderer nrc; neural_radiance_caching/neural_radiance_caching_main.cpp:755-790, defaults :458-460):
//   -position-encoding tri-wave|hash-grid (hash-grid)   -num-hidden-layers n (2)   -learning-rate lr (1e-2)
//   and, headless: -max-path-length n (5; 0 = unlimited, :1860-1861)   -no-train   -log10-radiance-scale s (0, :2240)
//   -nee lights|regir|restir (lights): next-event estimation from the emitter distributions (the reference), from the ReGIR grid, or --
//        at the first path vertex -- from the pixel's ReSTIR DI reservoir (the two halves of README.md:80-81)
// Textures are read by the host decoders of scene_builder.cpp (PPM / PGM / PFM / BMP / TGA, OpenEXR); DDS / PNG / JPEG assets have to be
// decoded offline (the image has no image libraries).
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>
#include "../../../include/gfxexp.h"
#include "../../../include/gfxexp_host.h"

namespace {

const double kPi = 3.14159265358979323846;

struct Quat {   // rotation quaternion (x, y, z, w); products compose like the reference's qRotate* * ori (:612-640)
    double x = 0, y = 0, z = 0, w = 1;
    bool finite() const { return std::isfinite(x) && std::isfinite(y) && std::isfinite(z) && std::isfinite(w); }
};
Quat mul(const Quat& a, const Quat& b) {
    return { a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
             a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z };
}
Quat axis_angle(int axis, double deg) {
    const double h = 0.5 * deg * kPi / 180.0;
    Quat q; q.w = std::cos(h);
    (axis == 0 ? q.x : axis == 1 ? q.y : q.z) = std::sin(h);
    if (axis != 0) q.x = 0;
    return q;
}
Quat slerp(double t, const Quat& a, Quat b) {
    double c = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
    if (c < 0) { b.x = -b.x; b.y = -b.y; b.z = -b.z; b.w = -b.w; c = -c; }
    if (c > 0.9995) return a;                       // begin == end (the common case of a translating light)
    const double th = std::acos(c), s = std::sin(th);
    const double ka = std::sin((1 - t) * th) / s, kb = std::sin(t * th) / s;
    return { ka * a.x + kb * b.x, ka * a.y + kb * b.y, ka * a.z + kb * b.z, ka * a.w + kb * b.w };
}
void to_matrix(const Quat& q, double m[9]) {
    const double xx = q.x * q.x, yy = q.y * q.y, zz = q.z * q.z, xy = q.x * q.y, yz = q.y * q.z, zx = q.z * q.x;
    const double xw = q.x * q.w, yw = q.y * q.w, zw = q.z * q.w;
    m[0] = 1 - 2 * (yy + zz); m[1] = 2 * (xy - zw);     m[2] = 2 * (zx + yw);
    m[3] = 2 * (xy + zw);     m[4] = 1 - 2 * (xx + zz); m[5] = 2 * (yz - xw);
    m[6] = 2 * (zx - yw);     m[7] = 2 * (yz + xw);     m[8] = 1 - 2 * (xx + yy);
}

struct MeshDesc { bool rectangle = false; std::string path; float preScale = 1.0f; int matConv = GFXH_MATCONV_TRADITIONAL;
                  float dimX = 1, dimZ = 1, emittance[3] = { 0, 0, 0 }; std::string emitterTex; };
struct InstDesc { std::string name; double beginPos[3], endPos[3]; Quat beginOri, endOri; double beginScale, endScale, frequency, time; };

struct Options {
    double camPos[3] = { 0, 0, 0 };
    Quat camOri;
    float brightness = 0.0f;
    std::string envTexture;
    std::map<std::string, MeshDesc> meshes;      // std::map: the reference iterates g_meshInfos in name order (:1125)
    std::vector<InstDesc> insts;
    uint32_t width = 1920, height = 1080, frames = 1;
    int renderer = GFXH_ORIGINAL_RESTIR_BIASED, device = 0;
    bool animate = false, accumulate = false, bump = false, dryRun = false;
    std::string out;
    // neural radiance caching (neural_radiance_caching_main.cpp:458-460)
    bool nrc = false, nrcTrain = true;
    int positionEncoding = GFX_NRC_HASH_GRID;
    uint32_t numHiddenLayers = 2, maxPathLength = 5;
    float learningRate = 1e-2f, log10RadianceScale = 0.0f;
    uint32_t neeSampler = 0;
};

[[noreturn]] void fail(const char* what, const char* arg) {
    std::fprintf(stderr, "restir_di_headless: %s%s%s\n", what, arg ? " " : "", arg ? arg : "");
    std::exit(EXIT_FAILURE);
}

// "-env-texture" (restir_di_main.cpp:1188-1197): a float lat-long image through the scene builder's image loader
void load_env_texture(const std::string& path, std::vector<float>& texels4, uint32_t& w, uint32_t& h) {
    gfxh_scene* tmp = gfxh_scene_create();
    const uint32_t slot = gfxh_scene_load_texture(tmp, path.c_str(), GFX_TEX_RGBA8_SRGB);
    uint32_t format = 0; const void* texels = nullptr;
    if (!slot || gfxh_scene_get_texture(tmp, slot, &w, &h, &format, &texels) || format != GFX_TEX_RGBA32F) fail("-env-texture wants a float image (.exr as the reference reads it, or .pfm):", path.c_str());
    texels4.assign(static_cast<const float*>(texels), static_cast<const float*>(texels) + 4ull * w * h);
    gfxh_scene_destroy(tmp);
}

bool apply_rotation(const char* arg, const char* prefix, double deg, Quat* ori) {
    const std::string a = arg, p = prefix;
    int axis;
    if (a == p + "roll") axis = 2;          // qRotateZ (:615-621)
    else if (a == p + "pitch") axis = 0;    // qRotateX (:623-629)
    else if (a == p + "yaw") axis = 1;      // qRotateY (:631-637)
    else return false;
    *ori = mul(axis_angle(axis, deg), *ori);
    return true;
}

Options parse(int argc, const char* argv[]) {
    Options o;
    std::string name;
    const double nan = std::nan("");
    double beginPos[3] = { 0, 0, 0 }, endPos[3] = { nan, nan, nan };
    Quat beginOri, endOri; endOri.x = endOri.y = endOri.z = endOri.w = nan;
    double beginScale = 1, endScale = nan, frequency = 5, initTime = 0;
    float emittance[3] = { 0, 0, 0 };
    std::string emitterTex;
    auto need = [&](int i, int n) { if (i + n >= argc) fail("option needs more arguments:", argv[i]); };
    auto num = [&](int i) { return std::atof(argv[i]); };
    for (int i = 1; i < argc; ++i) {
        const std::string a = argv[i];
        if (a.empty() || a[0] != '-') continue;                                      // :641-642
        if (a == "-cam-pos") { need(i, 3); for (int k = 0; k < 3; ++k) o.camPos[k] = num(i + 1 + k); i += 3; }
        else if (a == "-cam-roll" || a == "-cam-pitch" || a == "-cam-yaw") { need(i, 1); apply_rotation(argv[i], "-cam-", num(i + 1), &o.camOri); i += 1; }
        else if (a == "-brightness") { need(i, 1); o.brightness = static_cast<float>(std::fmin(std::fmax(num(i + 1), -5.0), 5.0)); i += 1; }   // :657-663
        else if (a == "-env-texture") { need(i, 1); o.envTexture = argv[i + 1]; i += 1; }
        else if (a == "-name") { need(i, 1); name = argv[i + 1]; i += 1; }
        else if (a == "-emittance") {                                                // :681-691
            need(i, 3);
            for (int k = 0; k < 3; ++k) { emittance[k] = static_cast<float>(num(i + 1 + k)); if (!std::isfinite(emittance[k])) fail("invalid value:", argv[i]); }
            i += 3;
        }
        else if (a == "-rect-emitter-tex") { need(i, 1); emitterTex = argv[i + 1]; i += 1; }
        else if (a == "-obj") {                                                      // :701-726
            need(i, 3);
            MeshDesc m; m.path = argv[i + 1]; m.preScale = static_cast<float>(num(i + 2));
            const std::string conv = argv[i + 3];
            if (conv == "trad") m.matConv = GFXH_MATCONV_TRADITIONAL;
            else if (conv == "simple_pbr") m.matConv = GFXH_MATCONV_SIMPLE_PBR;
            else fail("invalid material convention:", argv[i + 3]);
            o.meshes[name] = m;
            i += 3;
        }
        else if (a == "-rectangle") {                                                // :727-745: consumes the pending emittance / texture
            need(i, 2);
            MeshDesc m; m.rectangle = true; m.dimX = static_cast<float>(num(i + 1)); m.dimZ = static_cast<float>(num(i + 2));
            for (int k = 0; k < 3; ++k) m.emittance[k] = emittance[k];
            m.emitterTex = emitterTex;
            o.meshes[name] = m;
            emittance[0] = emittance[1] = emittance[2] = 0; emitterTex.clear();
            i += 2;
        }
        else if (a == "-begin-pos") { need(i, 3); for (int k = 0; k < 3; ++k) beginPos[k] = num(i + 1 + k); i += 3; }
        else if (a == "-begin-roll" || a == "-begin-pitch" || a == "-begin-yaw") { need(i, 1); apply_rotation(argv[i], "-begin-", num(i + 1), &beginOri); i += 1; }
        else if (a == "-begin-scale") { need(i, 1); beginScale = num(i + 1); i += 1; }
        else if (a == "-end-pos") { need(i, 3); for (int k = 0; k < 3; ++k) endPos[k] = num(i + 1 + k); i += 3; }
        else if (a == "-end-roll" || a == "-end-pitch" || a == "-end-yaw") {
            need(i, 1);
            if (!endOri.finite()) endOri = Quat();     // the first -end-* rotation starts from identity (:787-791 computeOrientation on a fresh quaternion)
            apply_rotation(argv[i], "-end-", num(i + 1), &endOri); i += 1;
        }
        else if (a == "-end-scale") { need(i, 1); endScale = num(i + 1); i += 1; }
        else if (a == "-freq") { need(i, 1); frequency = num(i + 1); i += 1; }
        else if (a == "-time") { need(i, 1); initTime = num(i + 1); i += 1; }
        else if (a == "-inst") {                                                     // :828-858: take the pending placement, reset it
            need(i, 1);
            InstDesc d; d.name = argv[i + 1];
            const bool endPosSet = std::isfinite(endPos[0]) && std::isfinite(endPos[1]) && std::isfinite(endPos[2]);
            for (int k = 0; k < 3; ++k) { d.beginPos[k] = beginPos[k]; d.endPos[k] = endPosSet ? endPos[k] : beginPos[k]; }
            d.beginOri = beginOri; d.endOri = endOri.finite() ? endOri : beginOri;
            d.beginScale = beginScale; d.endScale = std::isfinite(endScale) ? endScale : beginScale;
            d.frequency = frequency; d.time = initTime;
            o.insts.push_back(d);
            beginPos[0] = beginPos[1] = beginPos[2] = 0; endPos[0] = endPos[1] = endPos[2] = nan;
            beginOri = Quat(); endOri.x = endOri.y = endOri.z = endOri.w = nan;
            beginScale = 1; endScale = nan; frequency = 5; initTime = 0;
            i += 1;
        }
        // ---- headless
        else if (a == "-size") { need(i, 2); o.width = static_cast<uint32_t>(std::atoi(argv[i + 1])); o.height = static_cast<uint32_t>(std::atoi(argv[i + 2])); i += 2; }
        else if (a == "-frames") { need(i, 1); o.frames = static_cast<uint32_t>(std::atoi(argv[i + 1])); i += 1; }
        else if (a == "-device") { need(i, 1); o.device = std::atoi(argv[i + 1]); i += 1; }
        else if (a == "-out") { need(i, 1); o.out = argv[i + 1]; i += 1; }
        else if (a == "-renderer") {
            need(i, 1);
            const std::string r = argv[i + 1];
            if (r == "restir-biased") o.renderer = GFXH_ORIGINAL_RESTIR_BIASED;
            else if (r == "restir-unbiased") o.renderer = GFXH_ORIGINAL_RESTIR_UNBIASED;
            else if (r == "rearch-biased") o.renderer = GFXH_REARCHITECTED_RESTIR_BIASED;
            else if (r == "rearch-unbiased") o.renderer = GFXH_REARCHITECTED_RESTIR_UNBIASED;
            else if (r == "pt") o.renderer = GFXH_PATH_TRACE_BASELINE;
            else if (r == "regir") o.renderer = GFXH_PATH_TRACE_REGIR;
            else if (r == "nrc") o.nrc = true;
            else fail("unknown renderer:", argv[i + 1]);
            i += 1;
        }
        // ---- neural_radiance_caching_main.cpp:755-790
        else if (a == "-position-encoding") {
            need(i, 1);
            const std::string enc = argv[i + 1];
            if (enc == "tri-wave") o.positionEncoding = GFX_NRC_TRIANGLE_WAVE;
            else if (enc == "hash-grid") o.positionEncoding = GFX_NRC_HASH_GRID;
            else fail("invalid position encoding:", argv[i + 1]);
            i += 1;
        }
        else if (a == "-num-hidden-layers") { need(i, 1); o.numHiddenLayers = static_cast<uint32_t>(std::atoi(argv[i + 1])); i += 1; }
        else if (a == "-learning-rate") {
            need(i, 1);
            o.learningRate = static_cast<float>(std::atof(argv[i + 1]));
            if (!std::isfinite(o.learningRate)) fail("invalid value:", argv[i]);
            i += 1;
        }
        else if (a == "-max-path-length") { need(i, 1); o.maxPathLength = static_cast<uint32_t>(std::atoi(argv[i + 1])); i += 1; }
        else if (a == "-log10-radiance-scale") { need(i, 1); o.log10RadianceScale = static_cast<float>(std::atof(argv[i + 1])); i += 1; }
        else if (a == "-no-train") o.nrcTrain = false;
        else if (a == "-nee") {
            need(i, 1);
            const std::string v = argv[i + 1];
            if (v == "lights") o.neeSampler = 0;
            else if (v == "regir") o.neeSampler = 1;
            else if (v == "restir") o.neeSampler = 2;
            else fail("unknown NEE sampler:", argv[i + 1]);
            i += 1;
        }
        else if (a == "-animate") o.animate = true;
        else if (a == "-accumulate") o.accumulate = true;
        else if (a == "-bump") o.bump = true;
        else if (a == "-dry-run") o.dryRun = true;
        else fail("unknown option:", argv[i]);                                        // :860-863
    }
    if (o.width == 0 || o.height == 0 || o.frames == 0) fail("-size / -frames must be positive", nullptr);
    return o;
}

// InstanceController::updateBody (common_host.h:825-831), in double and rounded once
struct Controller {
    uint32_t instSlot; InstDesc d; double time;
    bool moving() const {
        return d.beginScale != d.endScale || std::memcmp(d.beginPos, d.endPos, sizeof(d.beginPos)) != 0 ||
               d.beginOri.x != d.endOri.x || d.beginOri.y != d.endOri.y || d.beginOri.z != d.endOri.z || d.beginOri.w != d.endOri.w;
    }
    void transform(double dt, float xfm[12], float normalMatrix[9]) {
        time = std::fmod(time + dt, d.frequency);
        const double t = 0.5 - 0.5 * std::cos(2 * kPi * time / d.frequency);
        const double scale = (1 - t) * d.beginScale + t * d.endScale;
        double R[9];
        to_matrix(slerp(t, d.beginOri, d.endOri), R);
        for (int r = 0; r < 3; ++r) {
            for (int c = 0; c < 3; ++c) { xfm[4 * r + c] = static_cast<float>(R[3 * r + c] * scale); normalMatrix[3 * r + c] = static_cast<float>(R[3 * r + c] / scale); }
            xfm[4 * r + 3] = static_cast<float>((1 - t) * d.beginPos[r] + t * d.endPos[r]);
        }
    }
};

bool ends_with(const std::string& s, const char* ext) {
    const size_t n = std::strlen(ext);
    return s.size() >= n && s.compare(s.size() - n, n, ext) == 0;
}

} // namespace

int main(int argc, const char* argv[]) {
    Options o = parse(argc, argv);
    gfxh_scene* scene = gfxh_scene_create();
    // meshes in name order (:1125-1147); the pre-scale of an OBJ is part of its group transform (:1133-1134), which this host
    // API folds into the instance transform
    std::map<std::string, uint32_t> groupOf;
    std::map<std::string, float> preScaleOf;
    for (const auto& kv : o.meshes) {
        const MeshDesc& m = kv.second;
        uint32_t group;
        if (m.rectangle) group = gfxh_scene_add_rectangle_textured(scene, m.dimX, m.dimZ, m.emittance, m.emitterTex.c_str());
        else group = gfxh_scene_load_obj_conv(scene, m.path.c_str(), m.matConv);
        if (group == 0xFFFFFFFFu) fail(gfxh_last_error(), nullptr);
        groupOf[kv.first] = group; preScaleOf[kv.first] = m.rectangle ? 1.0f : m.preScale;
    }
    std::vector<Controller> controllers;
    for (const InstDesc& d : o.insts) {                                               // :1149-1176
        if (!groupOf.count(d.name)) fail("-inst names an unknown mesh:", d.name.c_str());
        Controller c{ 0, d, d.time };
        // createInstance(instXfm = begin placement) (:1155-1157); the controller's first update comes with the first animated frame
        InstDesc first = d; first.endScale = d.beginScale; first.endOri = d.beginOri; std::memcpy(first.endPos, d.beginPos, sizeof(first.endPos));
        Controller placement{ 0, first, 0.0 };
        float xfm[12], nm[9];
        placement.transform(0.0, xfm, nm);
        const float pre = preScaleOf[d.name];
        for (int r = 0; r < 3; ++r) for (int col = 0; col < 3; ++col) xfm[4 * r + col] *= pre;
        c.instSlot = gfxh_scene_add_instance(scene, groupOf[d.name], xfm);
        if (c.moving()) controllers.push_back(c);
    }
    uint32_t counts[5];
    gfxh_scene_counts(scene, counts);
    float bounds[6] = { 0, 0, 0, 0, 0, 0 };
    if (counts[3]) gfxh_scene_bounds(scene, bounds);
    double camM[9];
    to_matrix(o.camOri, camM);
    std::printf("{\"materials\": %u, \"geometries\": %u, \"groups\": %u, \"instances\": %u, \"triangles\": %u, \"textures\": %u, \"animated_instances\": %zu,\n"
                " \"bounds\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g], \"camera_position\": [%.9g, %.9g, %.9g],\n"
                " \"camera_orientation\": [%.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g, %.9g], \"renderer\": %d, \"size\": [%u, %u], \"frames\": %u",
                counts[0], counts[1], counts[2], counts[3], counts[4], gfxh_scene_num_textures(scene), controllers.size(),
                bounds[0], bounds[1], bounds[2], bounds[3], bounds[4], bounds[5], o.camPos[0], o.camPos[1], o.camPos[2],
                camM[0], camM[1], camM[2], camM[3], camM[4], camM[5], camM[6], camM[7], camM[8], o.nrc ? -1 : o.renderer, o.width, o.height, o.frames);
    if (o.nrc) std::printf(",\n \"nrc\": {\"position_encoding\": \"%s\", \"num_hidden_layers\": %u, \"learning_rate\": %.9g, \"max_path_length\": %u, \"train\": %s, \"nee\": \"%s\"}",
                           o.positionEncoding == GFX_NRC_HASH_GRID ? "hash-grid" : "tri-wave", o.numHiddenLayers, o.learningRate, o.maxPathLength, o.nrcTrain ? "true" : "false",
                           o.neeSampler == 1 ? "regir" : o.neeSampler == 2 ? "restir" : "lights");
    std::printf(",\n \"instance_transforms\": [");
    for (uint32_t i = 0; i < counts[3]; ++i) {
        uint32_t group; float xfm[12];
        gfxh_scene_get_instance(scene, i, &group, xfm);
        std::printf("%s[%u", i ? ", " : "", group);
        for (int k = 0; k < 12; ++k) std::printf(", %.9g", xfm[k]);
        std::printf("]");
    }
    std::printf("]");
    if (o.dryRun) { std::printf("}\n"); gfxh_scene_destroy(scene); return 0; }
    if (counts[3] == 0) fail("the scene has no instances (-inst)", nullptr);

    gfx_ctx* ctx = nullptr;
    if (gfx_ctx_create(o.device, &ctx)) fail("gfx_ctx_create:", gfx_last_error(nullptr));
    if (gfxh_scene_upload(scene, ctx)) fail("gfxh_scene_upload:", gfxh_last_error());
    for (const Controller& c : controllers) gfx_instance_set_dynamic(ctx, c.instSlot, 1);
    auto animate_instances = [&]() {                                                  // :2249-2264
        for (Controller& c : controllers) {
            float xfm[12], nm[9];
            c.transform(1.0 / 60.0, xfm, nm);
            const float pre = preScaleOf[c.d.name];
            for (int r = 0; r < 3; ++r) for (int col = 0; col < 3; ++col) xfm[4 * r + col] *= pre;
            if (gfx_instance_set_transform_and_normal_matrix(ctx, c.instSlot, xfm, nm)) fail("gfx_instance_set_transform:", gfx_last_error(ctx));
        }
    };
    std::vector<float> rgba(4ull * o.width * o.height);
    gfxh_restir* renderer = nullptr;
    gfxh_nrc* nrc = nullptr;
    if (o.nrc) {
        // neural_radiance_caching_main.cpp: NeuralRadianceCache::initialize(g_positionEncoding, g_numHiddenLayers, g_learningRate)
        // (:1198), the frame loop :2225-2370 behind gfxh_nrc_render_frame
        gfxh_nrc_config cfg;
        gfxh_nrc_default_config(&cfg, o.width, o.height);
        cfg.positionEncoding = o.positionEncoding; cfg.numHiddenLayers = o.numHiddenLayers; cfg.learningRate = o.learningRate;
        cfg.maxPathLength = o.maxPathLength; cfg.train = o.nrcTrain ? 1u : 0u;
        cfg.radianceScale = std::pow(10.0f, o.log10RadianceScale);                    // :2240
        cfg.enableAccumulation = o.accumulate ? 1u : 0u;
        cfg.enableBumpMapping = o.bump ? 1u : 0u;
        cfg.neeSampler = o.neeSampler;
        for (int k = 0; k < 3; ++k) cfg.camera.position[k] = static_cast<float>(o.camPos[k]);
        for (int k = 0; k < 9; ++k) cfg.camera.orientation[k] = static_cast<float>(camM[k]);
        for (int k = 0; k < 3; ++k) { cfg.sceneAabbMin[k] = bounds[k]; cfg.sceneAabbMax[k] = bounds[3 + k]; }   // scene.initialSceneAabb (:1139)
        if (gfxh_nrc_create(ctx, &cfg, &nrc)) fail("gfxh_nrc_create:", gfxh_nrc_last_error());
        if (!o.envTexture.empty()) {
            std::vector<float> env; uint32_t w = 0, h = 0;
            load_env_texture(o.envTexture, env, w, h);
            if (gfxh_nrc_set_env(nrc, env.data(), w, h, 1.0f, 0.0f)) fail("gfxh_nrc_set_env:", gfxh_nrc_last_error());
        }
        float loss = 0.0f;
        for (uint32_t frame = 0; frame < o.frames; ++frame) {
            if (o.animate && frame > 0 && !controllers.empty()) {
                animate_instances();
                if (gfxh_nrc_rebuild_accel(nrc, nullptr)) fail("gfxh_nrc_rebuild_accel:", gfxh_nrc_last_error());
            }
            // the loss is read back (a host wait for the training stream) on the last frame only: on the others the four
            // training steps stay overlapped with the next frame
            if (gfxh_nrc_render_frame(nrc, nullptr, frame + 1 == o.frames ? &loss : nullptr)) fail("gfxh_nrc_render_frame:", gfxh_nrc_last_error());
        }
        uint32_t numTrainingData = 0, tileSize[2] = { 0, 0 }, numQueries = 0;
        gfxh_nrc_stats(nrc, &numTrainingData, tileSize, &numQueries);
        (void)gfxh_nrc_network(nrc);                                                  // joins the training stream: `loss` is final
        std::printf(",\n \"nrc_last_frame\": {\"training_records\": %u, \"tile_size\": [%u, %u], \"inference_queries\": %u, \"loss\": %.9g}",
                    numTrainingData, tileSize[0], tileSize[1], numQueries, loss);
        if (gfx_read_device(ctx, gfxh_nrc_beauty_buffer(nrc), rgba.data(), rgba.size() * sizeof(float))) fail("gfx_read_device:", gfx_last_error(ctx));
    }
    else {
    gfxh_restir_config cfg;
    gfxh_restir_default_config(&cfg, o.width, o.height, o.renderer);
    for (int k = 0; k < 3; ++k) cfg.camera.position[k] = static_cast<float>(o.camPos[k]);
    for (int k = 0; k < 9; ++k) cfg.camera.orientation[k] = static_cast<float>(camM[k]);
    cfg.enableAccumulation = o.accumulate ? 1u : 0u;
    cfg.enableBumpMapping = o.bump ? 1u : 0u;
    for (int k = 0; k < 3; ++k) { cfg.regirAabbMin[k] = bounds[k]; cfg.regirAabbMax[k] = bounds[3 + k]; }
    if (gfxh_restir_create(ctx, &cfg, &renderer)) fail("gfxh_restir_create:", gfxh_restir_last_error());
    if (!o.envTexture.empty()) {
        std::vector<float> env; uint32_t w = 0, h = 0;
        load_env_texture(o.envTexture, env, w, h);
        if (gfxh_restir_set_env(renderer, env.data(), w, h, 1.0f, 0.0f)) fail("gfxh_restir_set_env:", gfxh_restir_last_error());
    }
    for (uint32_t frame = 0; frame < o.frames; ++frame) {
        if (o.animate && frame > 0 && !controllers.empty()) {
            animate_instances();
            if (gfxh_restir_rebuild_accel(renderer, nullptr)) fail("gfxh_restir_rebuild_accel:", gfxh_restir_last_error());
        }
        if (gfxh_restir_render_frame(renderer, nullptr)) fail("gfxh_restir_render_frame:", gfxh_restir_last_error());
    }
    if (gfx_read_device(ctx, gfxh_restir_beauty_buffer(renderer), rgba.data(), rgba.size() * sizeof(float))) fail("gfx_read_device:", gfx_last_error(ctx));
    }
    double sum[3] = { 0, 0, 0 };
    for (size_t p = 0; p < static_cast<size_t>(o.width) * o.height; ++p) for (int k = 0; k < 3; ++k) sum[k] += rgba[4 * p + k];
    const double n = static_cast<double>(o.width) * o.height;
    std::printf(",\n \"mean_rgb\": [%.9g, %.9g, %.9g]", sum[0] / n, sum[1] / n, sum[2] / n);
    if (!o.out.empty()) {
        const float brightnessScale = std::pow(10.0f, o.brightness);                  // :2206 / saveImage callers
        int rc;
        if (ends_with(o.out, ".exr") || ends_with(o.out, ".pfm")) rc = gfxh_save_image_hdr(o.out.c_str(), o.width, o.height, brightnessScale, rgba.data(), 0);
        else { gfxh_sdr_config sdr = { 1.0f, brightnessScale, 1u, 1u, 0u }; rc = gfxh_save_image_sdr(o.out.c_str(), o.width, o.height, rgba.data(), &sdr); }
        if (rc) fail("saving the image:", gfxh_last_error());
        std::printf(", \"out\": \"%s\"", o.out.c_str());
    }
    std::printf("}\n");
    if (renderer) gfxh_restir_destroy(renderer);
    if (nrc) gfxh_nrc_destroy(nrc);
    gfxh_scene_destroy(scene);
    gfx_ctx_destroy(ctx);
    return 0;
}
