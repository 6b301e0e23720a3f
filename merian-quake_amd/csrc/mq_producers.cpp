// mq_producers.cpp -- per-frame geometry producers: what the reference's
// QuakeNode::update_dynamic_geo (src/game/quake_node.cpp:896-983) collects every frame into ONE non-opaque geometry,
//
//   particles        add_particles     src/game/quake_helpers.cpp:50-216   (a jittered tetrahedron per particle)
//   alias models     add_geo_alias     src/game/quake_helpers.cpp:218-359  (MDL poses, lerped, with previous positions)
//   brush entities   add_geo_brush     src/game/quake_helpers.cpp:362-469  (BSP submodels under the entity transform)
//   sprites          add_geo_sprite    src/game/quake_helpers.cpp:471-626  (two back-to-back quads)
//
// restated on PLAIN inputs: the reference reads quakespasm's live structures (entity_t, aliashdr_t, particle_t,
// msprite_t; its fork is an empty submodule here), this file takes the same quantities through small C structs and
// reads the on-disk formats itself (id Software's MDL "IDPO" version 6 and SPR "IDSP" version 1).
// Definitions where the reference leans on absent code: merian::XORShift32::next_double() = the renderer's xorshift32
// (13, 17, 5), (state >> 8) * 2^-24; glm::rotate = Rodrigues' rotation; quakespasm's 162-entry vertex-normal table is
// not reproduced -- the shaders do not read alias vertex normals (raytrace.glsl:280-286), the n0 / n1 / n2 fields carry
// the encoded geometric normal of the triangle instead (n1 != 0xffffffff is what marks "not a brush model").
#include "mq_host.h"

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <exception>
#include <string>

namespace {

struct V3 { float x, y, z; };
inline V3 operator+(V3 a, V3 b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V3 operator-(V3 a, V3 b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V3 operator*(V3 a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline float dot(V3 a, V3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
inline V3 cross(V3 a, V3 b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline V3 normalize(V3 a) { float l = std::sqrt(dot(a, a)); return l > 0.0f ? a * (1.0f / l) : a; }

uint16_t f2h(float f) { // round to nearest even, as merian::float_to_half
    uint32_t x; memcpy(&x, &f, 4);
    uint32_t sign = (x >> 16) & 0x8000u, ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
    if (ax < 0x33000001u) return (uint16_t)sign;
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift = e < -14 ? 13 + (-14 - e) : 13;
    uint32_t hexp = e < -14 ? 0u : (uint32_t)(e + 15);
    uint32_t hm = m >> shift, rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
    if (rem > half || (rem == half && (hm & 1u))) hm++;
    return (uint16_t)(sign | (hexp == 0 ? hm : ((hexp - 1) << 10) + hm));
}
uint32_t encode_normal(V3 n) { // the renderer's octahedral 2 x snorm16 codec (mq_device.h encode_normal)
    float l1 = std::fabs(n.x) + std::fabs(n.y) + std::fabs(n.z);
    float px = n.x / l1, py = n.y / l1;
    if (n.z < 0.0f) { float ox = (1.0f - std::fabs(py)) * (px >= 0.0f ? 1.0f : -1.0f), oy = (1.0f - std::fabs(px)) * (py >= 0.0f ? 1.0f : -1.0f); px = ox; py = oy; }
    auto q = [](float v) { v = std::min(1.0f, std::max(-1.0f, v)); return (int)std::floor(v * 32767.0f + 0.5f); };
    return ((uint32_t)q(px) & 0xffffu) | (((uint32_t)q(py) & 0xffffu) << 16);
}
struct XorShift { uint32_t s; double next() { s ^= s << 13; s ^= s >> 17; s ^= s << 5; return (double)(s >> 8) * (1.0 / 16777216.0); } };

// AngleVectors of quakespasm's mathlib (pitch, yaw, roll in degrees)
void angle_vectors(const float a[3], V3& fwd, V3& right, V3& up) {
    const float d2r = 3.14159265358979323846f / 180.0f;
    float sy = std::sin(a[1] * d2r), cy = std::cos(a[1] * d2r), sp = std::sin(a[0] * d2r), cp = std::cos(a[0] * d2r), sr = std::sin(a[2] * d2r), cr = std::cos(a[2] * d2r);
    fwd = {cp * cy, cp * sy, -sp};
    right = {(-1 * sr * sp * cy + -1 * cr * -sy), (-1 * sr * sp * sy + -1 * cr * cy), -1 * sr * cp};
    up = {(cr * sp * cy + -sr * -sy), (cr * sp * sy + -sr * cy), cr * cp};
}
struct M34 { V3 c0, c1, c2, t; V3 apply(V3 p) const { return ((c0 * p.x + c1 * p.y) + c2 * p.z) + t; } };
M34 entity_matrix(const float origin[3], const float angles[3]) { // columns forward, -right, up; translation origin (quake_helpers.cpp:266-269,372-376)
    V3 f, r, u; angle_vectors(angles, f, r, u);
    return {f, r * -1.0f, u, {origin[0], origin[1], origin[2]}};
}
V3 rodrigues(V3 v, V3 axis, float angle) { // glm::rotate(identity, angle, axis) * v
    float c = std::cos(angle), s = std::sin(angle);
    return (v * c + cross(axis, v) * s) + axis * (dot(axis, v) * (1.0f - c));
}

void push_vtx(MqHostGeo& g, V3 p, V3 q) { g.vtx.push_back(p.x); g.vtx.push_back(p.y); g.vtx.push_back(p.z); g.prev_vtx.push_back(q.x); g.prev_vtx.push_back(q.y); g.prev_vtx.push_back(q.z); }
mq_ext make_ext(uint16_t texnum_alpha, uint16_t fb_flags, uint32_t n0, uint32_t n1, uint32_t n2, float s0, float t0, float s1, float t1, float s2, float t2) {
    mq_ext e; e.texnum_alpha = texnum_alpha; e.texnum_fb_flags = fb_flags; e.n0_gloss_norm = n0; e.n1_brush = n1; e.n2 = n2;
    e.st[0] = f2h(s0); e.st[1] = f2h(t0); e.st[2] = f2h(s1); e.st[3] = f2h(t1); e.st[4] = f2h(s2); e.st[5] = f2h(t2);
    return e;
}
uint16_t texnum_alpha(uint32_t texnum, bool has_alpha) { return (uint16_t)(std::min<uint32_t>(texnum, MQ_MAX_GLTEXTURES - 1) | ((has_alpha ? 0u : 15u) << 12)); } // make_texnum_alpha, quake_helpers.cpp:26-48

bool read_file(const char* path, std::vector<uint8_t>& out) {
    FILE* f = fopen(path, "rb");
    if (!f) return false;
    fseek(f, 0, SEEK_END); long n = ftell(f); fseek(f, 0, SEEK_SET);
    if (n < 0) { fclose(f); return false; }
    out.resize((size_t)n);
    size_t got = n ? fread(out.data(), 1, (size_t)n, f) : 0;
    fclose(f);
    return got == (size_t)n;
}
template <class T> bool rd(const std::vector<uint8_t>& f, size_t& at, T& v) { if (at + sizeof(T) > f.size()) return false; memcpy(&v, f.data() + at, sizeof(T)); at += sizeof(T); return true; }

// 8-bit indexed pixels -> RGBA8 texture (+ fullbright mask texture if any texel index >= 224), as the BSP loader does for miptex
#define MQ_MAX_PICTURE_DIM 4096 // skins and sprite frames: far above anything id Software's formats hold, far below a size_t wrap
void upload_indexed(mq_ctx* ctx, const uint8_t* px, uint32_t w, uint32_t h, const uint8_t pal[768], bool transparent255, uint32_t texnum, uint32_t* fb_texnum, uint32_t* next_tex) {
    MqHostTex& t = mq_ctx_tex(ctx, texnum); t.w = w; t.h = h; t.flags = MQ_TEX_SRGB | MQ_TEX_MIPMAP; t.px.resize((size_t)w * h * 4);
    bool any_fb = false;
    for (size_t k = 0; k < (size_t)w * h; k++) {
        const uint8_t ci = px[k]; uint8_t* d = &t.px[4 * k];
        d[0] = pal[3 * ci]; d[1] = pal[3 * ci + 1]; d[2] = pal[3 * ci + 2]; d[3] = (transparent255 && ci == 255) ? 0 : 255;
        if (ci >= 224 && !(transparent255 && ci == 255)) any_fb = true;
    }
    if (fb_texnum) *fb_texnum = 0;
    if (any_fb && fb_texnum && *next_tex < MQ_MAX_GLTEXTURES) {
        *fb_texnum = (*next_tex)++;
        MqHostTex& f = mq_ctx_tex(ctx, *fb_texnum); f.w = w; f.h = h; f.flags = MQ_TEX_SRGB | MQ_TEX_MIPMAP; f.px.assign((size_t)w * h * 4, 0);
        for (size_t k = 0; k < (size_t)w * h; k++) { const uint8_t ci = px[k]; if (ci >= 224 && !(transparent255 && ci == 255)) { uint8_t* d = &f.px[4 * k]; d[0] = pal[3 * ci]; d[1] = pal[3 * ci + 1]; d[2] = pal[3 * ci + 2]; d[3] = 255; } }
    }
}

} // namespace

bool mq_read_palette(const char* palette_path, uint8_t pal[768], std::string& err) {
    for (int i = 0; i < 256; i++) pal[3 * i] = pal[3 * i + 1] = pal[3 * i + 2] = (uint8_t)i; // grey ramp stand-in
    if (!palette_path || !*palette_path) return true;
    std::vector<uint8_t> p;
    if (!read_file(palette_path, p) || p.size() < 768) { err = std::string("cannot read palette ") + palette_path; return false; }
    memcpy(pal, p.data(), 768);
    return true;
}

extern "C" {

int mq_dyn_begin(mq_ctx* ctx) {
    if (!ctx) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    P.pending = MqHostGeo(); P.collecting = true;
    return MQ_OK;
}

int mq_dyn_end(mq_ctx* ctx, int slot) { // the one per-frame geometry: not opaque (alpha tests apply), not static (quake_node.cpp:969-981)
    if (!ctx) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    if (!P.collecting) return mq_ctx_fail(ctx, MQ_ESTATE, "mq_dyn_end without mq_dyn_begin");
    P.collecting = false;
    MqHostGeo& g = P.pending;
    return mq_scene_set_geometry(ctx, slot, g.vtx.data(), g.prev_vtx.data(), (uint32_t)(g.vtx.size() / 3), g.idx.data(), g.ext.data(), g.n_tri(), 0u);
}

// add_particles, quake_helpers.cpp:50-216
int mq_dyn_add_particles(mq_ctx* ctx, const mq_particle* parts, uint32_t n, const mq_view* view, uint32_t texnum_blood, uint32_t texnum_explosion, double cl_time, double prev_cl_time) {
    if (!ctx || (!parts && n) || !view) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    if (!P.collecting) return mq_ctx_fail(ctx, MQ_ESTATE, "mq_dyn_add_particles outside mq_dyn_begin / mq_dyn_end");
    MqHostGeo& g = P.pending;
    static const V3 voff[4] = {{0.0f, 1.0f, 0.0f}, {-0.5f, -0.5f, 0.87f}, {-0.5f, -0.5f, -0.87f}, {1.0f, -0.5f, 0.0f}};
    const V3 vpn = {view->forward[0], view->forward[1], view->forward[2]}, r_origin = {view->origin[0], view->origin[1], view->origin[2]};
    // every particle is a tetrahedron of its own (4 vertices, 4 triangles) with its own random stream: written in place, in parallel
    const size_t v_at = g.vtx.size(), i_at = g.idx.size(), e_at = g.ext.size();
    g.vtx.resize(v_at + 12 * (size_t)n); g.prev_vtx.resize(v_at + 12 * (size_t)n); g.idx.resize(i_at + 12 * (size_t)n); g.ext.resize(e_at + 4 * (size_t)n);
    mq_parallel_for(n, 512, [&](size_t p_begin, size_t p_end) {
    for (size_t pi = p_begin; pi < p_end; pi++) {
        const mq_particle& p = parts[pi];
        const V3 org = {p.org[0], p.org[1], p.org[2]}, prev_org = {p.prev_org[0], p.prev_org[1], p.prev_org[2]}, vel = {p.vel[0], p.vel[1], p.vel[2]};
        float scale = dot(org - r_origin, vpn); // from r_part.c
        scale = scale < 20.0f ? 1.0f + 0.08f : 1.0f + scale * 0.004f;
        scale *= 0.5f;
        uint8_t cb[4] = {(uint8_t)(p.color_rgba & 0xff), (uint8_t)((p.color_rgba >> 8) & 0xff), (uint8_t)((p.color_rgba >> 16) & 0xff), (uint8_t)(p.color_rgba >> 24)};
        XorShift xr{p.seed ? p.seed : 1u};
        uint32_t texnum = 0, texnum_fb = 0; // heuristics for blood, fire, explosions (:95-113)
        if (cb[1] == 0 && cb[2] == 0 && cb[0] > 10) texnum = texnum_blood;
        else if (p.type == MQ_PT_EXPLODE2) { texnum = texnum_fb = texnum_explosion; scale *= 2.0f; }
        else if (p.type == MQ_PT_FIRE && (cb[0] != cb[1] || cb[1] != cb[2] || cb[0] != cb[2])) { texnum = texnum_fb = texnum_explosion; scale *= 2.0f; }
        else if (0.299 * cb[0] + 0.587 * cb[1] + 0.114 * cb[2] > 200) { texnum = texnum_fb = texnum_explosion; scale *= 2.0f; }
        V3 vert[4], prev_vert[4];
        const float speed = std::sqrt(dot(vel, vel));
        for (int l = 0; l < 3; l++) { // (the reference overwrites the vertices three times: the last pass counts, every pass draws random numbers)
            const float particle_offset = (float)(2 * (xr.next() - 0.5) + 2 * (xr.next() - 0.5));
            const double rand_angle = xr.next();
            const double ax = xr.next(), ay = xr.next(), az = xr.next();
            const V3 axis = normalize(V3{(float)ax, (float)ay, (float)az});
            const float ang = (float)((rand_angle + cl_time * 0.001 * speed) * 2 * M_PI), prev_ang = (float)((rand_angle + prev_cl_time * 0.001 * speed) * 2 * M_PI);
            // only the last pass's vertices survive: the first two only advance the random stream (3 draws per vertex); the rotation's
            // sine and cosine are per pass, not per vertex (the values rodrigues() computes: same expressions, same bits)
            if (l < 2) { for (int k = 0; k < 12; k++) (void)xr.next(); continue; }
            const float ca = std::cos(ang), sa = std::sin(ang), cp = std::cos(prev_ang), sp = std::sin(prev_ang);
            auto rot = [&](V3 v, float c, float s) { return (v * c + cross(axis, v) * s) + axis * (dot(axis, v) * (1.0f - c)); };
            for (int k = 0; k < 4; k++) {
                const float vertex_offset = (float)(0.5 * ((xr.next() - 0.5) + (xr.next() - 0.5)));
                const float rand_offset_scale = (float)xr.next();
                const V3 local = (voff[k] * scale) * (1.0f + rand_offset_scale) + V3{vertex_offset, vertex_offset, vertex_offset};
                vert[k] = (org + V3{particle_offset, particle_offset, particle_offset}) + rot(local, ca, sa);
                prev_vert[k] = (prev_org + V3{particle_offset, particle_offset, particle_offset}) + rot(local, cp, sp);
            }
        }
        const uint32_t base = (uint32_t)(v_at / 3 + 4 * pi);
        for (int k = 0; k < 4; k++) {
            float* v = &g.vtx[v_at + 12 * pi + 3 * k]; float* q = &g.prev_vtx[v_at + 12 * pi + 3 * k];
            v[0] = vert[k].x; v[1] = vert[k].y; v[2] = vert[k].z; q[0] = prev_vert[k].x; q[1] = prev_vert[k].y; q[2] = prev_vert[k].z;
        }
        static const uint32_t tet[12] = {0, 1, 2, 0, 2, 3, 0, 3, 1, 1, 3, 2};
        for (int k = 0; k < 4; k++) {
            const uint32_t i0 = base + tet[3 * k], i1 = base + tet[3 * k + 1], i2 = base + tet[3 * k + 2];
            uint32_t* ix = &g.idx[i_at + 12 * pi + 3 * k]; ix[0] = i0; ix[1] = i1; ix[2] = i2;
            mq_ext& ex = g.ext[e_at + 4 * pi + k];
            if (texnum) { // texture patch
                const V3 a = vert[tet[3 * k]], b = vert[tet[3 * k + 1]], c = vert[tet[3 * k + 2]];
                const uint32_t enc = encode_normal(normalize(cross(c - a, b - a)));
                ex = make_ext((uint16_t)texnum, (uint16_t)texnum_fb, enc, enc, enc, 0, 1, 0, 0, 1, 0);
            } else { // solid colour (:193-211)
                for (int i = 0; i < 3; i++) cb[0] = (uint8_t)std::min(255.0, std::max(0.0, cb[0] * (1 + xr.next() * 0.1 - 0.05)));
                const uint32_t c = (uint32_t)cb[0] | ((uint32_t)cb[1] << 8) | ((uint32_t)cb[2] << 16) | ((uint32_t)cb[3] << 24);
                const uint32_t c_orig = p.color_rgba;
                const uint32_t c_fb = (0.299 * cb[0] + 0.587 * cb[1] + 0.114 * cb[2] > 150) ? c : 0u; // bright colours are probably emitting
                (void)c_orig;
                ex = make_ext(0, (uint16_t)(MQ_MAT_FLAGS_SOLID << 12), c, c_fb, 0, 0, 1, 0, 0, 1, 0);
            }
        }
    }
    });
    return MQ_OK;
}

// add_geo_sprite, quake_helpers.cpp:471-626
int mq_dyn_add_sprite(mq_ctx* ctx, int sprite_model, const mq_sprite_instance* inst, const mq_view* view) {
    if (!ctx || !inst || !view) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    if (!P.collecting) return mq_ctx_fail(ctx, MQ_ESTATE, "mq_dyn_add_sprite outside mq_dyn_begin / mq_dyn_end");
    if (sprite_model < 0 || (size_t)sprite_model >= P.sprites.size()) return mq_ctx_fail(ctx, MQ_EINVAL, "unknown sprite model");
    const MqSpriteModel& m = P.sprites[(size_t)sprite_model];
    if (m.frames.empty()) return MQ_OK;
    const MqSpriteFrame& fr = m.frames[(size_t)std::min<int>(std::max(inst->frame, 0), (int)m.frames.size() - 1)];
    if (!fr.texnum) return MQ_OK; // "if (!frame->gltexture) return"
    MqHostGeo& g = P.pending;
    const V3 vpn = {view->forward[0], view->forward[1], view->forward[2]}, vright = {view->right[0], view->right[1], view->right[2]}, vup = {view->up[0], view->up[1], view->up[2]};
    const V3 r_origin = {view->origin[0], view->origin[1], view->origin[2]}, origin = {inst->origin[0], inst->origin[1], inst->origin[2]}, prev_origin = {inst->prev_origin[0], inst->prev_origin[1], inst->prev_origin[2]};
    V3 s_up, s_right;
    switch (m.type) {
    case 0: s_up = {0, 0, 1}; s_right = normalize(cross(vpn, s_up)); break;                                            // SPR_VP_PARALLEL_UPRIGHT
    case 1: { V3 f = origin - r_origin; f.z = 0; f = normalize(f); s_right = {f.y, -f.x, 0}; s_up = {0, 0, 1}; break; } // SPR_FACING_UPRIGHT
    case 2: s_up = vup; s_right = vright; break;                                                                      // SPR_VP_PARALLEL
    case 3: { V3 f, r, u; angle_vectors(inst->angles, f, r, u); s_up = u; s_right = r; break; }                        // SPR_ORIENTED
    case 4: { const float a = inst->angles[2] * (3.14159265358979323846f / 180.0f), sr = std::sin(a), cr = std::cos(a); // SPR_VP_PARALLEL_ORIENTED
              s_right = vright * cr + vup * sr; s_up = vright * -sr + vup * cr; break; }
    default: return MQ_OK;
    }
    s_up = normalize(s_up); s_right = normalize(s_right);
    const float scale = inst->scale > 0.0f ? inst->scale : 1.0f;
    for (int k = 0; k < 2; k++) { // two quads, back to back
        const float sg = k == 0 ? 1.0f : -1.0f;
        const V3 v0 = (s_up * fr.down + s_right * (sg * fr.left)) * scale, v1 = (s_up * fr.up + s_right * (sg * fr.left)) * scale;
        const V3 v2 = (s_up * fr.up + s_right * (sg * fr.right)) * scale, v3 = (s_up * fr.down + s_right * (sg * fr.right)) * scale;
        const uint32_t base = (uint32_t)(g.vtx.size() / 3);
        push_vtx(g, v0 + origin, v0 + prev_origin); push_vtx(g, v1 + origin, v1 + prev_origin); push_vtx(g, v2 + origin, v2 + prev_origin); push_vtx(g, v3 + origin, v3 + prev_origin);
        g.idx.push_back(base); g.idx.push_back(base + 1); g.idx.push_back(base + 2);
        g.idx.push_back(base); g.idx.push_back(base + 2); g.idx.push_back(base + 3);
        const uint32_t enc = encode_normal(normalize(cross(v2 - v0, v1 - v0)));
        const uint16_t tn = texnum_alpha(fr.texnum, fr.alpha);
        const uint16_t fl = (uint16_t)(MQ_MAT_FLAGS_SPRITE << 12); // a sprite always emits
        g.ext.push_back(make_ext(tn, fl, enc, enc, enc, 0, fr.tmax, 0, 0, fr.smax, 0));
        g.ext.push_back(make_ext(tn, fl, enc, enc, enc, 0, fr.tmax, fr.smax, 0, fr.smax, fr.tmax));
    }
    return MQ_OK;
}

// add_geo_alias, quake_helpers.cpp:218-359
namespace {
// One alias-model entity into caller-provided slices (nv = m.vertindex.size() vertices, m.indexes.size() indices, a third as many extra-data records):
// the body of add_geo_alias, quake_helpers.cpp:218-359.  `base`: number of the slice's first vertex in the slot.
void alias_write(const MqAliasModel& m, const mq_alias_instance* in, float* vtx, float* prev, uint32_t* idx, mq_ext* ext, uint32_t base) {
    const V3 fov = {1.0f, in->fovscale > 0.0f ? in->fovscale : 1.0f, in->fovscale > 0.0f ? in->fovscale : 1.0f}; // view model: makes the gun fov independent (:244-246)
    const float ang[3] = {-in->angles[0], in->angles[1], in->angles[2]}, pang[3] = {-in->prev_angles[0], in->prev_angles[1], in->prev_angles[2]}; // "lerpdata.angles[0] *= -1"; the stored previous angles are already negated the same way
    const M34 mm = entity_matrix(in->origin, ang), pm = entity_matrix(in->prev_origin, pang);
    const V3 so = {m.scale_origin[0] * fov.x, m.scale_origin[1] * fov.y, m.scale_origin[2] * fov.z}, sc = {m.scale[0] * fov.x, m.scale[1] * fov.y, m.scale[2] * fov.z};
    const uint32_t nvbo = (uint32_t)m.vertindex.size();
    for (uint32_t v = 0; v < nvbo; v++) {
        const uint8_t* a = &m.trivertexes[4 * ((size_t)m.numverts * (size_t)in->pose1 + m.vertindex[v])];
        const uint8_t* b = &m.trivertexes[4 * ((size_t)m.numverts * (size_t)in->pose2 + m.vertindex[v])];
        auto lerp = [&](float t) { return V3{(float)a[0] * (1.0f - t) + (float)b[0] * t, (float)a[1] * (1.0f - t) + (float)b[1] * t, (float)a[2] * (1.0f - t) + (float)b[2] * t}; };
        auto model = [&](V3 p) { return V3{p.x * sc.x + so.x, p.y * sc.y + so.y, p.z * sc.z + so.z}; };
        const V3 p = mm.apply(model(lerp(in->blend))), q = pm.apply(model(lerp(in->prev_blend)));
        vtx[3 * v] = p.x; vtx[3 * v + 1] = p.y; vtx[3 * v + 2] = p.z; prev[3 * v] = q.x; prev[3 * v + 1] = q.y; prev[3 * v + 2] = q.z;
    }
    for (size_t i = 0; i < m.indexes.size(); i++) idx[i] = base + m.indexes[i];
    const size_t nskins = m.skin_texnum.size();
    const size_t sk = nskins ? (size_t)std::min<int>(std::max(in->skin, 0), (int)nskins - 1) : 0;
    for (size_t t = 0; t < m.indexes.size() / 3; t++) {
        const uint16_t i0 = m.indexes[3 * t], i1 = m.indexes[3 * t + 1], i2 = m.indexes[3 * t + 2];
        uint32_t n0, n1, n2;
        if (nskins && m.skin_norm_texnum[sk]) { n0 = (m.skin_gloss_texnum[sk] & 0xffffu) | (m.skin_norm_texnum[sk] << 16); n1 = 0xffffffffu; n2 = 0; } // pack_uint32(gloss, norm): marks "use the normal map"
        else {
            const float* p0 = &vtx[3 * (size_t)i0]; const float* p1 = &vtx[3 * (size_t)i1]; const float* p2 = &vtx[3 * (size_t)i2];
            const V3 a = {p0[0], p0[1], p0[2]}, b = {p1[0], p1[1], p1[2]}, c = {p2[0], p2[1], p2[2]};
            n0 = n1 = n2 = encode_normal(normalize(cross(c - a, b - a)));
            if (n1 == 0xffffffffu) n1 = n0 = n2 = 0xfffffffeu; // never the brush-model marker
        }
        const float iw = 1.0f / (float)m.skinwidth, ih = 1.0f / (float)m.skinheight;
        ext[t] = make_ext(texnum_alpha(nskins ? m.skin_texnum[sk] : 0u, false), (uint16_t)(nskins ? m.skin_fb_texnum[sk] : 0u), n0, n1, n2,
                          (m.st[2 * i0] + 0.5f) * iw, (m.st[2 * i0 + 1] + 0.5f) * ih, (m.st[2 * i1] + 0.5f) * iw, (m.st[2 * i1 + 1] + 0.5f) * ih, (m.st[2 * i2] + 0.5f) * iw, (m.st[2 * i2 + 1] + 0.5f) * ih);
    }
}
} // namespace

int mq_dyn_add_alias(mq_ctx* ctx, int alias_model, const mq_alias_instance* in) { return mq_dyn_add_alias_batch(ctx, &alias_model, in, 1); }

// n entities at once, on the worker pool (the reference runs add_geo_alias under a parallel_for over the visible entities, quake_node.cpp:904-938):
// the same triangles, in the same order, as n calls of mq_dyn_add_alias
int mq_dyn_add_alias_batch(mq_ctx* ctx, const int* alias_models, const mq_alias_instance* in, uint32_t n) {
    if (!ctx || (n && (!alias_models || !in))) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    if (!P.collecting) return mq_ctx_fail(ctx, MQ_ESTATE, "mq_dyn_add_alias outside mq_dyn_begin / mq_dyn_end");
    for (uint32_t e = 0; e < n; e++) if (alias_models[e] < 0 || (size_t)alias_models[e] >= P.alias.size()) return mq_ctx_fail(ctx, MQ_EINVAL, "unknown alias model");
    MqHostGeo& g = P.pending;
    std::vector<size_t> v_at(n + 1), i_at(n + 1);
    v_at[0] = g.vtx.size() / 3; i_at[0] = g.idx.size();
    for (uint32_t e = 0; e < n; e++) {
        const MqAliasModel& m = P.alias[(size_t)alias_models[e]];
        const bool skip = in[e].pose1 < 0 || in[e].pose2 < 0 || (uint32_t)in[e].pose1 >= m.numposes || (uint32_t)in[e].pose2 >= m.numposes; // "if (f < 0 || f >= hdr->numposes) return"
        v_at[e + 1] = v_at[e] + (skip ? 0 : m.vertindex.size()); i_at[e + 1] = i_at[e] + (skip ? 0 : m.indexes.size());
    }
    g.vtx.resize(3 * v_at[n]); g.prev_vtx.resize(3 * v_at[n]); g.idx.resize(i_at[n]); g.ext.resize(i_at[n] / 3);
    mq_parallel_for(n, 1, [&](size_t b, size_t e1) {
        for (size_t e = b; e < e1; e++) if (v_at[e + 1] != v_at[e] || i_at[e + 1] != i_at[e])
            alias_write(P.alias[(size_t)alias_models[e]], &in[e], &g.vtx[3 * v_at[e]], &g.prev_vtx[3 * v_at[e]], &g.idx[i_at[e]], &g.ext[i_at[e] / 3], (uint32_t)v_at[e]);
    });
    return MQ_OK;
}

// add_geo_brush for an entity's brush model (doors, platforms, ...), quake_helpers.cpp:362-469: the submodel's
// triangles under the entity's transform, previous positions under the previous one
int mq_dyn_add_brush_model(mq_ctx* ctx, int bsp_model, const float origin[3], const float angles[3], const float prev_origin[3], const float prev_angles[3]) {
    if (!ctx || !origin || !angles || !prev_origin || !prev_angles) return MQ_EINVAL;
    MqProducerState& P = mq_ctx_producers(ctx);
    if (!P.collecting) return mq_ctx_fail(ctx, MQ_ESTATE, "mq_dyn_add_brush_model outside mq_dyn_begin / mq_dyn_end");
    if (bsp_model < 1 || (size_t)bsp_model >= P.bsp_models.size()) return mq_ctx_fail(ctx, MQ_EINVAL, "unknown brush model (models 1.. of the loaded BSP)");
    const MqHostGeo& src = P.bsp_models[(size_t)bsp_model];
    MqHostGeo& g = P.pending;
    const float a[3] = {-angles[0], angles[1], angles[2]}, pa[3] = {-prev_angles[0], prev_angles[1], prev_angles[2]};
    const M34 mm = entity_matrix(origin, a), pm = entity_matrix(prev_origin, pa);
    const uint32_t base = (uint32_t)(g.vtx.size() / 3);
    for (size_t v = 0; v < src.vtx.size() / 3; v++) { const V3 p = {src.vtx[3 * v], src.vtx[3 * v + 1], src.vtx[3 * v + 2]}; push_vtx(g, mm.apply(p), pm.apply(p)); }
    for (uint32_t i : src.idx) g.idx.push_back(base + i);
    g.ext.insert(g.ext.end(), src.ext.begin(), src.ext.end());
    return MQ_OK;
}

int mq_bsp_model_count(const mq_ctx* ctx) { return ctx ? (int)mq_ctx_producers(const_cast<mq_ctx*>(ctx)).bsp_models.size() : 0; }

// ---- id Software MDL ("IDPO", version 6) -> MqAliasModel + skin textures ------------------------------------------
static int load_mdl_impl(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* model_out, uint32_t* next_texnum_out) {
    if (!ctx || !path || !model_out) return MQ_EINVAL;
    std::vector<uint8_t> f; std::string err;
    if (!read_file(path, f)) return mq_ctx_fail(ctx, MQ_EIO, std::string("cannot read ") + path);
    uint8_t pal[768];
    if (!mq_read_palette(palette_path, pal, err)) return mq_ctx_fail(ctx, MQ_EIO, err);
    size_t at = 0;
    struct Hdr { int32_t ident, version; float scale[3], scale_origin[3], boundingradius, eye[3]; int32_t numskins, skinwidth, skinheight, numverts, numtris, numframes, synctype, flags; float size; } h;
    if (!rd(f, at, h) || h.ident != 0x4f504449 || h.version != 6) return mq_ctx_fail(ctx, MQ_EIO, "not an IDPO version 6 model");
    if (h.numskins < 1 || h.skinwidth < 1 || h.skinheight < 1 || h.numverts < 1 || h.numtris < 1 || h.numframes < 1 || h.numverts > 65535 || h.numtris > 65535) return mq_ctx_fail(ctx, MQ_EIO, "bad model header");
    if (h.skinwidth > MQ_MAX_PICTURE_DIM || h.skinheight > MQ_MAX_PICTURE_DIM) return mq_ctx_fail(ctx, MQ_EIO, "skin larger than 4096 x 4096"); // (the products below stay far from wrapping size_t)
    MqAliasModel m;
    memcpy(m.scale, h.scale, 12); memcpy(m.scale_origin, h.scale_origin, 12);
    m.skinwidth = (uint32_t)h.skinwidth; m.skinheight = (uint32_t)h.skinheight; m.numverts = (uint32_t)h.numverts;
    uint32_t next_tex = first_texnum;
    const size_t skin_px = (size_t)h.skinwidth * h.skinheight;
    for (int s = 0; s < h.numskins; s++) { // single skins and skin groups (the first picture of a group is used)
        int32_t group; if (!rd(f, at, group)) return mq_ctx_fail(ctx, MQ_EIO, "truncated skins");
        int32_t npics = 1;
        if (group) { if (!rd(f, at, npics) || npics < 1 || (size_t)npics > f.size() / skin_px) return mq_ctx_fail(ctx, MQ_EIO, "bad skin group"); at += 4 * (size_t)npics; } // (a group cannot hold more pictures than the file has bytes for)
        if (at > f.size() || skin_px * (size_t)npics > f.size() - at) return mq_ctx_fail(ctx, MQ_EIO, "truncated skin");
        if (next_tex + 2 >= MQ_MAX_GLTEXTURES) return mq_ctx_fail(ctx, MQ_EINVAL, "too many textures");
        const uint32_t tn = next_tex++; uint32_t fb = 0;
        upload_indexed(ctx, f.data() + at, (uint32_t)h.skinwidth, (uint32_t)h.skinheight, pal, false, tn, &fb, &next_tex);
        m.skin_texnum.push_back(tn); m.skin_fb_texnum.push_back(fb); m.skin_norm_texnum.push_back(0); m.skin_gloss_texnum.push_back(0);
        at += skin_px * (size_t)npics;
    }
    struct StVert { int32_t onseam, s, t; }; struct Tri { int32_t facesfront, v[3]; };
    std::vector<StVert> stv((size_t)h.numverts); std::vector<Tri> tris((size_t)h.numtris);
    for (auto& v : stv) if (!rd(f, at, v)) return mq_ctx_fail(ctx, MQ_EIO, "truncated texture coordinates");
    for (auto& t : tris) { if (!rd(f, at, t)) return mq_ctx_fail(ctx, MQ_EIO, "truncated triangles"); for (int k = 0; k < 3; k++) if (t.v[k] < 0 || t.v[k] >= h.numverts) return mq_ctx_fail(ctx, MQ_EIO, "triangle vertex out of range"); }
    for (int fr = 0; fr < h.numframes; fr++) { // every pose of every frame (a frame group contributes all of its poses, as quakespasm's posenum numbering)
        int32_t type; if (!rd(f, at, type)) return mq_ctx_fail(ctx, MQ_EIO, "truncated frames");
        int32_t nposes = 1;
        if (type) { if (!rd(f, at, nposes) || nposes < 1) return mq_ctx_fail(ctx, MQ_EIO, "bad frame group"); at += 8 + 4 * (size_t)nposes; } // group min / max, intervals
        for (int p = 0; p < nposes; p++) {
            at += 8 + 16; // bboxmin, bboxmax (trivertx each), name[16]
            if (at + 4 * (size_t)h.numverts > f.size()) return mq_ctx_fail(ctx, MQ_EIO, "truncated pose");
            m.trivertexes.insert(m.trivertexes.end(), f.data() + at, f.data() + at + 4 * (size_t)h.numverts);
            at += 4 * (size_t)h.numverts; m.numposes++;
        }
    }
    // GL_MakeAliasModelDisplayLists_VBO: one VBO vertex per distinct (vertex, s, t); back-facing triangles take seam vertices from the right half of the skin
    for (const Tri& t : tris) for (int k = 0; k < 3; k++) {
        const int vi = t.v[k]; float s = (float)stv[(size_t)vi].s; const float tt = (float)stv[(size_t)vi].t;
        if (!t.facesfront && stv[(size_t)vi].onseam) s += (float)(h.skinwidth / 2);
        size_t found = m.vertindex.size();
        for (size_t v = 0; v < m.vertindex.size(); v++) if (m.vertindex[v] == vi && m.st[2 * v] == s && m.st[2 * v + 1] == tt) { found = v; break; }
        if (found == m.vertindex.size()) { m.vertindex.push_back((uint16_t)vi); m.st.push_back(s); m.st.push_back(tt); }
        m.indexes.push_back((uint16_t)found);
    }
    MqProducerState& P = mq_ctx_producers(ctx);
    P.alias.push_back(std::move(m));
    *model_out = (int)P.alias.size() - 1;
    if (next_texnum_out) *next_texnum_out = next_tex;
    return MQ_OK;
}

// ---- id Software SPR ("IDSP", version 1) -> MqSpriteModel + frame textures -------------------------------------------
static int load_spr_impl(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* model_out, uint32_t* next_texnum_out) {
    if (!ctx || !path || !model_out) return MQ_EINVAL;
    std::vector<uint8_t> f; std::string err;
    if (!read_file(path, f)) return mq_ctx_fail(ctx, MQ_EIO, std::string("cannot read ") + path);
    uint8_t pal[768];
    if (!mq_read_palette(palette_path, pal, err)) return mq_ctx_fail(ctx, MQ_EIO, err);
    size_t at = 0;
    struct Hdr { int32_t ident, version, type; float boundingradius; int32_t width, height, numframes; float beamlength; int32_t synctype; } h;
    if (!rd(f, at, h) || h.ident != 0x50534449 || h.version != 1 || h.numframes < 1) return mq_ctx_fail(ctx, MQ_EIO, "not an IDSP version 1 sprite");
    MqSpriteModel m; m.type = h.type;
    uint32_t next_tex = first_texnum;
    for (int fr = 0; fr < h.numframes; fr++) {
        int32_t group; if (!rd(f, at, group)) return mq_ctx_fail(ctx, MQ_EIO, "truncated sprite");
        int32_t n = 1;
        if (group) { if (!rd(f, at, n) || n < 1) return mq_ctx_fail(ctx, MQ_EIO, "bad sprite group"); at += 4 * (size_t)n; }
        for (int k = 0; k < n; k++) { // (the first picture of a group is the frame; the others are read past)
            struct Fr { int32_t origin[2], width, height; } fh;
            if (!rd(f, at, fh) || fh.width < 1 || fh.height < 1 || fh.width > MQ_MAX_PICTURE_DIM || fh.height > MQ_MAX_PICTURE_DIM || (size_t)fh.width * fh.height > f.size() - at) return mq_ctx_fail(ctx, MQ_EIO, "truncated sprite frame");
            if (k == 0) {
                if (next_tex + 1 >= MQ_MAX_GLTEXTURES) return mq_ctx_fail(ctx, MQ_EINVAL, "too many textures");
                MqSpriteFrame sf; sf.up = (float)fh.origin[1]; sf.down = (float)((int64_t)fh.origin[1] - fh.height); sf.left = (float)fh.origin[0]; sf.right = (float)((int64_t)fh.width + fh.origin[0]); // (64-bit: the origin of a damaged file may be anything)
                sf.smax = 1.0f; sf.tmax = 1.0f; sf.texnum = next_tex++; sf.alpha = true; // index 255 is transparent: TEXPREF_ALPHA
                upload_indexed(ctx, f.data() + at, (uint32_t)fh.width, (uint32_t)fh.height, pal, true, sf.texnum, nullptr, &next_tex);
                m.frames.push_back(sf);
            }
            at += (size_t)fh.width * fh.height;
        }
    }
    MqProducerState& P = mq_ctx_producers(ctx);
    P.sprites.push_back(std::move(m));
    *model_out = (int)P.sprites.size() - 1;
    if (next_texnum_out) *next_texnum_out = next_tex;
    return MQ_OK;
}
// no C++ exception crosses the C ABI: an allocation a damaged file provokes (std::length_error, std::bad_alloc) is an I/O error
int mq_load_mdl(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* model_out, uint32_t* next_texnum_out) {
    try { return load_mdl_impl(ctx, path, palette_path, first_texnum, model_out, next_texnum_out); }
    catch (const std::exception& e) { return mq_ctx_fail(ctx, MQ_EIO, std::string("model file: ") + e.what()); }
}
int mq_load_spr(mq_ctx* ctx, const char* path, const char* palette_path, uint32_t first_texnum, int* model_out, uint32_t* next_texnum_out) {
    try { return load_spr_impl(ctx, path, palette_path, first_texnum, model_out, next_texnum_out); }
    catch (const std::exception& e) { return mq_ctx_fail(ctx, MQ_EIO, std::string("sprite file: ") + e.what()); }
}

} // extern "C"

// ---- per-frame uniform: QuakeNode::process, quake_node.cpp:768-824 ---------------------------------------------------
extern "C" int mq_uniform_update(mq_uniform* u, const mq_frame_state* in) {
    if (!u || !in) return MQ_EINVAL;
    uint32_t flags = 0; // :770-779 (one flags byte, three padding bytes)
    if (in->render && in->has_player) flags = (in->weapon == 1 ? (uint32_t)MQ_PLAYER_FLAGS_TORCH : 0u) | (in->waterlevel >= 3 ? (uint32_t)MQ_PLAYER_FLAGS_UNDERWATER : 0u);
    u->player = flags;
    u->frame = in->frame;
    memcpy(u->prev_cam_x, u->cam_x, 16); memcpy(u->prev_cam_w, u->cam_w, 16); memcpy(u->prev_cam_u, u->cam_u, 16); // :781-783
    V3 f, r, up; angle_vectors(in->viewangles, f, r, up);                                                      // :784-786
    u->cam_w[0] = f.x; u->cam_w[1] = f.y; u->cam_w[2] = f.z;
    u->cam_u[0] = up.x; u->cam_u[1] = up.y; u->cam_u[2] = up.z;
    u->cam_x[0] = in->vieworg[0]; u->cam_x[1] = in->vieworg[1]; u->cam_x[2] = in->vieworg[2]; u->cam_x[3] = 1.0f; // :787
    uint16_t sky[6]; for (uint16_t& t : sky) t = in->notexture;                                                 // :788-799
    if (in->render && in->sky_mode == 1) for (int i = 0; i < 6; i++) sky[i] = in->sky[i];
    else if (in->render && in->sky_mode == 2) { sky[0] = in->sky[0]; sky[1] = in->sky[1]; sky[2] = 0xffffu; }
    u->sky_rt_bk = (uint32_t)sky[0] | ((uint32_t)sky[1] << 16); u->sky_lf_ft = (uint32_t)sky[2] | ((uint32_t)sky[3] << 16); u->sky_up_dn = (uint32_t)sky[4] | ((uint32_t)sky[5] << 16);
    if (in->mu_overwrite) {                                                                                    // :801-805
        u->cam_x[3] = in->mu_t;
        u->prev_cam_x[3] = in->mu_s_div_mu_t[0] * in->mu_t; u->prev_cam_w[3] = in->mu_s_div_mu_t[1] * in->mu_t; u->prev_cam_u[3] = in->mu_s_div_mu_t[2] * in->mu_t;
    } else {                                                                                                   // :806-816
        u->cam_x[3] = std::pow(in->fog_density, 2.f) * 0.1f;
        u->prev_cam_x[3] = std::pow(in->fog_color[0], 1.f / 1.2f) * u->cam_x[3];
        u->prev_cam_w[3] = std::pow(in->fog_color[1], 1.f / 1.2f) * u->cam_x[3];
        u->prev_cam_u[3] = std::pow(in->fog_color[2], 1.f / 1.2f) * u->cam_x[3];
    }
    const float time_diff = (float)(in->cl_time - (double)u->cl_time);                                         // :817-823 (cl.time is a double, the uniform a float)
    u->cam_w[3] = time_diff > 0 ? time_diff : 1.0f;
    u->cl_time = (float)in->cl_time;
    return MQ_OK;
}
extern "C" int mq_constants_fov(mq_constants* k, float fov_x_degrees) { // :762-765
    if (!k) return MQ_EINVAL;
    k->fov = fov_x_degrees;
    k->fov_tan_alpha_half = std::tan(fov_x_degrees * (3.14159265358979323846f / 180.0f) / 2);
    return MQ_OK;
}
