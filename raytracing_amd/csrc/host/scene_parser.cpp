// scene_parser.cpp -- the reference's line-oriented scene format (src/parser.rs:104-199) and
// the derivation of the camera tangents (src/parser.rs:53-67).
//
// Grammar: one statement per line, tokens separated by SINGLE spaces (the reference splits on
// ' ', so a doubled space yields an empty token that fails to parse if it is read, and a
// trailing space yields an unused empty token).  Only the first token selects the statement;
// operands beyond those a statement reads are ignored.
#include "../../../include/rtx_host.h"

#include <cerrno>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <new>
#include <string>
#include <vector>

namespace {

// fixed storage: reporting an error must not allocate (it may be reporting an allocation failure)
thread_local char g_error[512] = "";

int fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_error, sizeof(g_error), fmt, ap);
    va_end(ap);
    return code;
}

struct ParseError {
    std::string what;
};

// Rust's `str::parse::<f32>`: optional sign, then "inf" / "infinity" / "nan" (any case) or a
// decimal with optional fraction and exponent; nothing else (no blanks, no hex).  The value is
// the correctly rounded binary32, which is what strtof returns.
float parse_f32(const std::string &tok)
{
    const char *s = tok.c_str();
    size_t i = 0, n = tok.size();
    if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
    auto ieq = [&](const char *w) {
        size_t k = 0;
        while (w[k] && i + k < n && std::tolower((unsigned char)s[i + k]) == w[k]) ++k;
        return w[k] == 0 && i + k == n;
    };
    bool ok = false;
    if (ieq("inf") || ieq("infinity") || ieq("nan")) {
        ok = true;
    } else {
        size_t digits = 0;
        while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++digits; }
        if (i < n && s[i] == '.') {
            ++i;
            while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++digits; }
        }
        if (digits > 0) {
            if (i < n && (s[i] == 'e' || s[i] == 'E')) {
                ++i;
                if (i < n && (s[i] == '+' || s[i] == '-')) ++i;
                size_t ed = 0;
                while (i < n && std::isdigit((unsigned char)s[i])) { ++i; ++ed; }
                ok = ed > 0 && i == n;
            } else {
                ok = i == n;
            }
        }
    }
    if (!ok) throw ParseError{"not a number: '" + tok + "'"};
    return std::strtof(s, nullptr);
}

// Rust's `str::parse::<usize>`: optional '+', decimal digits.
uint32_t parse_usize(const std::string &tok)
{
    size_t i = 0, n = tok.size();
    if (i < n && tok[i] == '+') ++i;
    if (i == n) throw ParseError{"not an integer: '" + tok + "'"};
    uint64_t v = 0;
    for (; i < n; ++i) {
        if (!std::isdigit((unsigned char)tok[i])) throw ParseError{"not an integer: '" + tok + "'"};
        v = v * 10 + (uint64_t)(tok[i] - '0');
        if (v > 0xffffffffull) throw ParseError{"integer too large: '" + tok + "'"};
    }
    return (uint32_t)v;
}

const std::string &operand(const std::vector<std::string> &tokens, size_t i)
{
    if (i >= tokens.size()) throw ParseError{"missing operand " + std::to_string(i)};
    return tokens[i];
}
void parse_vec3(const std::vector<std::string> &tokens, float out[3])
{
    for (size_t k = 0; k < 3; ++k) out[k] = parse_f32(operand(tokens, 1 + k));
}

struct Builder {
    bool have_dims = false, have_depth = false, have_samples = false, have_bg = false;
    bool have_pos = false, have_axis[3] = {false, false, false}, have_fov = false;
    uint32_t width = 0, height = 0, ray_depth = 0, n_samples = 0;
    float bg[3], cam_pos[3], axis[3][3], fov_x = 0;
    std::vector<rtx_prim> prims;

    rtx_prim &last()
    {
        if (prims.empty()) throw ParseError{"modifier before any primitive"};
        return prims.back();
    }
    // Object::new (src/objects/object.rs:19-27) + PositionedFigure::new (src/objects/figures.rs:25-33)
    void push(uint32_t kind, const float param[3])
    {
        rtx_prim p;
        std::memset(&p, 0, sizeof(p));
        p.kind = kind;
        std::memcpy(p.param, param, sizeof(p.param));
        p.rotation[3] = 1.0f;
        p.material = RTX_DIFFUSE;
        p.ior = 1.0f;
        prims.push_back(p);
    }

    void statement(const std::vector<std::string> &t)
    {
        const std::string &kw = t[0];
        float v[3];
        if (kw == "DIMENSIONS") {
            width = parse_usize(operand(t, 1));
            height = parse_usize(operand(t, 2));
            have_dims = true;
        } else if (kw == "RAY_DEPTH") {
            ray_depth = parse_usize(operand(t, 1));
            have_depth = true;
        } else if (kw == "SAMPLES") {
            n_samples = parse_usize(operand(t, 1));
            have_samples = true;
        } else if (kw == "BG_COLOR") {
            parse_vec3(t, bg);
            have_bg = true;
        } else if (kw == "CAMERA_POSITION") {
            parse_vec3(t, cam_pos);
            have_pos = true;
        } else if (kw == "CAMERA_RIGHT") {
            parse_vec3(t, axis[0]);
            have_axis[0] = true;
        } else if (kw == "CAMERA_UP") {
            parse_vec3(t, axis[1]);
            have_axis[1] = true;
        } else if (kw == "CAMERA_FORWARD") {
            parse_vec3(t, axis[2]);
            have_axis[2] = true;
        } else if (kw == "CAMERA_FOV_X") {
            fov_x = parse_f32(operand(t, 1));
            have_fov = true;
        } else if (kw == "NEW_PRIMITIVE") {
            // src/parser.rs:139: no effect
        } else if (kw == "PLANE") {
            parse_vec3(t, v);
            push(RTX_PLANE, v);
        } else if (kw == "ELLIPSOID") {
            parse_vec3(t, v);
            push(RTX_ELLIPSOID, v);
        } else if (kw == "BOX") {
            parse_vec3(t, v);
            push(RTX_BOX, v);
        } else if (kw == "POSITION") {
            parse_vec3(t, v);
            std::memcpy(last().position, v, sizeof(v));
        } else if (kw == "ROTATION") {
            // src/parser.rs:209-216: UnitQuaternion::from_quaternion divides by the norm;
            // nalgebra adds a 4-vector's squares as (x*x + z*z) + (y*y + w*w)
            float q[4];
            for (size_t k = 0; k < 4; ++k) q[k] = parse_f32(operand(t, 1 + k));
            rtx_prim &p = last();
            float n = std::sqrt((q[0] * q[0] + q[2] * q[2]) + (q[1] * q[1] + q[3] * q[3]));
            for (int k = 0; k < 4; ++k) p.rotation[k] = q[k] / n;
        } else if (kw == "COLOR") {
            parse_vec3(t, v);
            std::memcpy(last().color, v, sizeof(v));
        } else if (kw == "EMISSION") {
            parse_vec3(t, v);
            std::memcpy(last().emission, v, sizeof(v));
        } else if (kw == "METALLIC") {
            last().material = RTX_METALLIC;
        } else if (kw == "DIELECTRIC") {
            rtx_prim &p = last();
            p.material = RTX_DIELECTRIC;
            p.ior = 1.0f;  // src/parser.rs:185
        } else if (kw == "IOR") {
            float ior = parse_f32(operand(t, 1));
            rtx_prim &p = last();
            if (p.material == RTX_DIELECTRIC) p.ior = ior;  // src/parser.rs:190
        }
        // anything else: src/parser.rs:194
    }
};

int finish(Builder &b, rtx_host_scene *out)
{
    // the Option::unwrap calls of src/parser.rs:51-66, 92-95
    if (!b.have_dims) return fail(RTX_ERR_INVALID_SCENE, "missing DIMENSIONS");
    if (!b.have_fov) return fail(RTX_ERR_INVALID_SCENE, "missing CAMERA_FOV_X");
    for (int k = 0; k < 3; ++k)
        if (!b.have_axis[k]) return fail(RTX_ERR_INVALID_SCENE, "missing CAMERA_RIGHT / CAMERA_UP / CAMERA_FORWARD");
    if (!b.have_pos) return fail(RTX_ERR_INVALID_SCENE, "missing CAMERA_POSITION");
    if (!b.have_depth) return fail(RTX_ERR_INVALID_SCENE, "missing RAY_DEPTH");
    if (!b.have_samples) return fail(RTX_ERR_INVALID_SCENE, "missing SAMPLES");
    if (!b.have_bg) return fail(RTX_ERR_INVALID_SCENE, "missing BG_COLOR");

    std::memset(out, 0, sizeof(*out));
    rtx_scene &s = out->scene;
    s.width = b.width;
    s.height = b.height;
    s.ray_depth = b.ray_depth;
    s.n_samples = b.n_samples;
    std::memcpy(s.background, b.bg, sizeof(s.background));
    std::memcpy(s.cam_pos, b.cam_pos, sizeof(s.cam_pos));
    for (int c = 0; c < 3; ++c)
        for (int r = 0; r < 3; ++r) s.cam_axis[3 * c + r] = b.axis[c][r];  // columns right, up, forward
    // src/parser.rs:53-55
    s.tg_fov_x = std::tan(b.fov_x / 2.0f);
    float aspect = (float)b.height / (float)b.width;
    s.tg_fov_y = aspect * s.tg_fov_x;
    s.n_prims = (uint32_t)b.prims.size();
    out->fov_x = b.fov_x;
    out->prims = nullptr;
    if (!b.prims.empty()) {
        out->prims = static_cast<rtx_prim *>(std::malloc(sizeof(rtx_prim) * b.prims.size()));
        if (!out->prims) return fail(RTX_ERR_NO_MEMORY, "out of memory");
        std::memcpy(out->prims, b.prims.data(), sizeof(rtx_prim) * b.prims.size());
    }
    s.prims = out->prims;
    return RTX_OK;
}

int parse_stream(std::istream &in, rtx_host_scene *out)
{
    Builder b;
    std::string line;
    size_t lineno = 0;
    while (std::getline(in, line)) {
        ++lineno;
        if (!line.empty() && line.back() == '\r') line.pop_back();  // BufRead::lines strips "\r\n"
        std::vector<std::string> tokens;
        size_t start = 0;
        while (true) {
            size_t pos = line.find(' ', start);
            if (pos == std::string::npos) {
                tokens.push_back(line.substr(start));
                break;
            }
            tokens.push_back(line.substr(start, pos - start));
            start = pos + 1;
        }
        try {
            b.statement(tokens);
        } catch (const ParseError &e) {
            return fail(RTX_ERR_INVALID_SCENE, "line %zu: %s", lineno, e.what.c_str());
        }
    }
    return finish(b, out);
}

}  // namespace

extern "C" {

const char *rtx_host_last_error(void) { return g_error; }

// no exception crosses the C boundary: running out of memory is a status code
int rtx_host_parse_file(const char *path, rtx_host_scene *out)
{
    if (!path || !out) return fail(RTX_ERR_INVALID_ARG, "null argument");
    try {
        std::ifstream f(path);
        if (!f) return fail(RTX_ERR_INVALID_SCENE, "cannot open %s", path);
        return parse_stream(f, out);
    } catch (const std::bad_alloc &) {
        return fail(RTX_ERR_NO_MEMORY, "out of memory while parsing %s", path);
    } catch (...) {
        return fail(RTX_ERR_INVALID_SCENE, "unexpected exception while parsing %s", path);
    }
}

int rtx_host_parse_text(const char *text, rtx_host_scene *out)
{
    if (!text || !out) return fail(RTX_ERR_INVALID_ARG, "null argument");
    try {
        std::istringstream s{std::string(text)};
        return parse_stream(s, out);
    } catch (const std::bad_alloc &) {
        return fail(RTX_ERR_NO_MEMORY, "out of memory while parsing the scene text");
    } catch (...) {
        return fail(RTX_ERR_INVALID_SCENE, "unexpected exception while parsing the scene text");
    }
}

void rtx_host_free_scene(rtx_host_scene *scene)
{
    if (!scene) return;
    std::free(scene->prims);
    scene->prims = nullptr;
    scene->scene.prims = nullptr;
    scene->scene.n_prims = 0;
}

}  // extern "C"
