// mq_synth.cpp -- seeded synthetic "BSP-like" scenes.
//
// No Quake .bsp/.pak is available to this build, so every BASELINE config has a runnable stand-in
// with the same data contract the reference's QuakeNode emits (src/game/quake_node.cpp:847-1012):
// world-space triangle soup + per-triangle VertexExtraData (src/game/quake_helpers.cpp:426-461),
// RGBA8 textures, worldspawn-style sun, classic two-layer sky, fog.  Layout: a maze of
// axis-aligned rooms (indoor with ceilings, outdoor under a sky brush) joined by door tunnels,
// with octagonal pillars, crates, emissive (fullbright) tiles, alpha-tested grates (non-opaque
// geometry slot, exercises the any-hit path) and a few moving boxes (dynamic slot with prev_vtx).
#include "mq_host.h"

#include <cmath>
#include <cstring>

namespace {

struct Rng {
    uint64_t s;
    explicit Rng(uint64_t seed) : s(seed * 0x9E3779B97F4A7C15ull + 0x1234567ull) { next(); next(); }
    uint64_t next() { s ^= s >> 12; s ^= s << 25; s ^= s >> 27; return s * 0x2545F4914F6CDD1Dull; }
    float uni() { return (float)(next() >> 40) * (1.0f / 16777216.0f); }
    int range(int n) { return (int)(next() % (uint64_t)n); }
};

struct V { float x, y, z; };
inline V operator+(V a, V b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
inline V operator-(V a, V b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
inline V operator*(V a, float s) { return {a.x * s, a.y * s, a.z * s}; }
inline V cross(V a, V b) { return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x}; }
inline float dot(V a, V b) { return a.x * b.x + a.y * b.y + a.z * b.z; }

uint16_t f2h_host(float f) { // round-to-nearest-even float -> half (merian::float_to_half)
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

enum { TEX_FB = 17, TEX_SKY_BACK = 18, TEX_SKY_FRONT = 19, TEX_GRATE = 20, TEX_NORMAL = 21, TEX_GLOSS = 22 };

struct Mesh {
    MqHostGeo* g;
    void quad(V a, V b, V c, V d, V n, const float st[8], const mq_ext& proto, const V* vel = nullptr) {
        // wind so that cross(v2-v0, v1-v0) (the reference's geometric normal) points along n
        float sts[8]; memcpy(sts, st, sizeof sts);
        if (dot(cross(c - a, b - a), n) < 0.0f) { std::swap(b, d); std::swap(sts[2], sts[6]); std::swap(sts[3], sts[7]); }
        uint32_t base = (uint32_t)(g->vtx.size() / 3);
        V vs[4] = {a, b, c, d};
        for (auto& v : vs) {
            g->vtx.push_back(v.x); g->vtx.push_back(v.y); g->vtx.push_back(v.z);
            V p = vel ? v - *vel : v;
            g->prev_vtx.push_back(p.x); g->prev_vtx.push_back(p.y); g->prev_vtx.push_back(p.z);
        }
        const uint32_t tri[2][3] = {{0, 1, 2}, {0, 2, 3}};
        for (auto& t : tri) {
            g->idx.push_back(base + t[0]); g->idx.push_back(base + t[1]); g->idx.push_back(base + t[2]);
            mq_ext e = proto;
            for (int k = 0; k < 3; k++) { e.st[2 * k] = f2h_host(sts[2 * t[k]]); e.st[2 * k + 1] = f2h_host(sts[2 * t[k] + 1]); }
            g->ext.push_back(e);
        }
    }
};

struct Gen {
    Rng rng;
    Mesh world, alpha, dyn;
    float q; // tessellation size
    float light_frac_ceiling, light_frac_wall;
    explicit Gen(uint64_t seed) : rng(seed), q(32.0f), light_frac_ceiling(0.06f), light_frac_wall(0.015f) {}

    mq_ext material(int tex, bool emissive, bool fancy) {
        mq_ext e; memset(&e, 0, sizeof e);
        e.texnum_alpha = (uint16_t)(tex | (15u << 12)); // opaque (quake_helpers.cpp:26-48)
        e.texnum_fb_flags = emissive ? (uint16_t)TEX_FB : 0;
        e.n1_brush = 0xffffffffu;                        // brush model marker (quake_helpers.cpp:431)
        e.n0_gloss_norm = fancy ? ((uint32_t)TEX_GLOSS | ((uint32_t)TEX_NORMAL << 16)) : 0u; // pack_uint32(gloss, norm)
        return e;
    }

    // tessellated rectangle: origin o, edges eu/ev (axis aligned), inward normal n
    void rect(Mesh& m, V o, V eu, V ev, V n, int tex, float light_frac, V st_origin, bool fancy, int flags_override = -1) {
        float lu = std::sqrt(dot(eu, eu)), lv = std::sqrt(dot(ev, ev));
        if (lu < 1e-3f || lv < 1e-3f) return;
        int nu = std::max(1, (int)std::ceil(lu / q - 1e-4f)), nv = std::max(1, (int)std::ceil(lv / q - 1e-4f));
        V du = eu * (1.0f / lu), dv = ev * (1.0f / lv);
        float s0 = dot(o - st_origin, du) / 64.0f, t0 = dot(o - st_origin, dv) / 64.0f;
        const int lone = light_frac < 0.0f ? rng.range(nu * nv) : -1; // light_frac < 0: exactly one emissive tile on this face
        for (int j = 0; j < nv; j++) for (int i = 0; i < nu; i++) {
            float u0 = lu * i / nu, u1 = lu * (i + 1) / nu, v0 = lv * j / nv, v1 = lv * (j + 1) / nv;
            V a = o + du * u0 + dv * v0, b = o + du * u1 + dv * v0, c = o + du * u1 + dv * v1, d = o + du * u0 + dv * v1;
            float st[8] = {s0 + u0 / 64.0f, t0 + v0 / 64.0f, s0 + u1 / 64.0f, t0 + v0 / 64.0f, s0 + u1 / 64.0f, t0 + v1 / 64.0f, s0 + u0 / 64.0f, t0 + v1 / 64.0f};
            bool em = lone >= 0 ? (j * nu + i == lone) : (light_frac > 0.0f && rng.uni() < light_frac);
            mq_ext e = material(tex, em, fancy);
            if (flags_override >= 0) e.texnum_fb_flags = (uint16_t)((e.texnum_fb_flags & 0xfffu) | ((uint32_t)flags_override << 12));
            m.quad(a, b, c, d, n, st, e);
        }
    }

    // wall in the plane (fixed axis) with an optional door hole centred on it
    void wall(V o, V eu, float height, V n, int tex, bool door, float door_w, float door_h, V st_origin, bool fancy) {
        float lu = std::sqrt(dot(eu, eu));
        V du = eu * (1.0f / lu), up = {0, 0, 1};
        if (!door) { rect(world, o, eu, up * height, n, tex, light_frac_wall, st_origin, fancy); return; }
        float a = 0.5f * (lu - door_w), b = a + door_w;
        rect(world, o, du * a, up * height, n, tex, light_frac_wall, st_origin, fancy);
        rect(world, o + du * b, du * (lu - b), up * height, n, tex, light_frac_wall, st_origin, fancy);
        rect(world, o + du * a + up * door_h, du * door_w, up * (height - door_h), n, tex, light_frac_wall, st_origin, fancy);
    }

    void box(Mesh& m, V lo, V hi, int tex, bool bottom, V st_origin, const V* vel = nullptr) {
        V c[8]; for (int i = 0; i < 8; i++) c[i] = {(i & 1) ? hi.x : lo.x, (i & 2) ? hi.y : lo.y, (i & 4) ? hi.z : lo.z};
        const int f[6][4] = {{0, 2, 6, 4}, {1, 3, 7, 5}, {0, 1, 5, 4}, {2, 3, 7, 6}, {0, 1, 3, 2}, {4, 5, 7, 6}};
        const V n[6] = {{-1, 0, 0}, {1, 0, 0}, {0, -1, 0}, {0, 1, 0}, {0, 0, -1}, {0, 0, 1}};
        for (int k = 0; k < 6; k++) {
            if (k == 4 && !bottom) continue;
            V a = c[f[k][0]], b = c[f[k][1]], cc = c[f[k][2]], d = c[f[k][3]];
            V eu = b - a, ev = d - a;
            float lu = std::sqrt(dot(eu, eu)), lv = std::sqrt(dot(ev, ev));
            float s0 = dot(a - st_origin, eu) / (lu * 64.0f), t0 = dot(a - st_origin, ev) / (lv * 64.0f);
            float st[8] = {s0, t0, s0 + lu / 64.0f, t0, s0 + lu / 64.0f, t0 + lv / 64.0f, s0, t0 + lv / 64.0f};
            mq_ext e = material(tex, false, false);
            m.quad(a, b, cc, d, n[k], st, e, vel);
        }
    }

    void pillar(V base, float radius, float height, int tex, V st_origin) {
        const int sides = 8;
        int nz = std::max(1, (int)std::ceil(height / (2.0f * q)));
        for (int k = 0; k < sides; k++) {
            float a0 = 6.2831853f * k / sides, a1 = 6.2831853f * (k + 1) / sides;
            V p0 = {base.x + radius * std::cos(a0), base.y + radius * std::sin(a0), base.z};
            V p1 = {base.x + radius * std::cos(a1), base.y + radius * std::sin(a1), base.z};
            V n = {std::cos(0.5f * (a0 + a1)), std::sin(0.5f * (a0 + a1)), 0};
            float w = std::sqrt(dot(p1 - p0, p1 - p0));
            for (int j = 0; j < nz; j++) {
                float z0 = height * j / nz, z1 = height * (j + 1) / nz;
                V a = {p0.x, p0.y, base.z + z0}, b = {p1.x, p1.y, base.z + z0}, c = {p1.x, p1.y, base.z + z1}, d = {p0.x, p0.y, base.z + z1};
                float s0 = k * w / 64.0f;
                float st[8] = {s0, z0 / 64.0f, s0 + w / 64.0f, z0 / 64.0f, s0 + w / 64.0f, z1 / 64.0f, s0, z1 / 64.0f};
                world.quad(a, b, c, d, n, st, material(tex, false, false));
            }
        }
        (void)st_origin;
    }
};

void make_textures(mq_ctx* ctx, Rng& rng) {
    for (int t = 1; t <= 16; t++) { // albedo: uniform random in [32,224] with a per-texture tint
        MqHostTex& tx = mq_ctx_tex(ctx, (uint32_t)t);
        tx.w = tx.h = 64; tx.flags = MQ_TEX_SRGB | MQ_TEX_MIPMAP | ((t & 1) ? MQ_TEX_LINEAR : 0u); // world textures carry TEXPREF_MIPMAP
        tx.px.resize(64 * 64 * 4);
        float tint[3] = {0.6f + 0.4f * rng.uni(), 0.6f + 0.4f * rng.uni(), 0.6f + 0.4f * rng.uni()};
        for (int i = 0; i < 64 * 64; i++) {
            int v = 32 + rng.range(193);
            for (int c = 0; c < 3; c++) tx.px[4 * i + c] = (uint8_t)(v * tint[c]);
            tx.px[4 * i + 3] = 255;
        }
    }
    { // fullbright: values in [0.5, 1]
        MqHostTex& tx = mq_ctx_tex(ctx, TEX_FB); tx.w = tx.h = 64; tx.flags = MQ_TEX_SRGB | MQ_TEX_MIPMAP; tx.px.resize(64 * 64 * 4);
        for (int i = 0; i < 64 * 64; i++) { int v = 128 + rng.range(128); tx.px[4 * i] = (uint8_t)v; tx.px[4 * i + 1] = (uint8_t)(v * 0.9f); tx.px[4 * i + 2] = (uint8_t)(v * 0.7f); tx.px[4 * i + 3] = 255; }
    }
    for (int layer = 0; layer < 2; layer++) { // classic two-layer sky (back solid, front with alpha holes)
        MqHostTex& tx = mq_ctx_tex(ctx, layer ? TEX_SKY_FRONT : TEX_SKY_BACK); tx.w = tx.h = 128; tx.flags = MQ_TEX_SRGB | MQ_TEX_LINEAR; tx.px.resize(128 * 128 * 4);
        for (int y = 0; y < 128; y++) for (int x = 0; x < 128; x++) {
            float f = 0.5f + 0.25f * std::sin(x * 0.196f + layer) + 0.25f * std::sin(y * 0.147f + 2 * layer);
            uint8_t* p = &tx.px[4 * (y * 128 + x)];
            p[0] = (uint8_t)(60 + 60 * f); p[1] = (uint8_t)(80 + 70 * f); p[2] = (uint8_t)(120 + 90 * f);
            p[3] = layer ? (uint8_t)(f > 0.55f ? 200 : 0) : 255;
        }
    }
    { // grate: alpha-tested bars
        MqHostTex& tx = mq_ctx_tex(ctx, TEX_GRATE); tx.w = tx.h = 64; tx.flags = MQ_TEX_SRGB; tx.px.resize(64 * 64 * 4);
        for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) {
            bool bar = (x % 16) < 4 || (y % 16) < 4;
            uint8_t* p = &tx.px[4 * (y * 64 + x)];
            p[0] = 90; p[1] = 80; p[2] = 70; p[3] = bar ? 255 : 0;
        }
    }
    { // tangent-space normal map (linear)
        MqHostTex& tx = mq_ctx_tex(ctx, TEX_NORMAL); tx.w = tx.h = 64; tx.flags = MQ_TEX_LINEAR; tx.px.resize(64 * 64 * 4);
        for (int y = 0; y < 64; y++) for (int x = 0; x < 64; x++) {
            float nx = 0.25f * std::sin(x * 0.3927f), ny = 0.25f * std::sin(y * 0.3927f), nz = std::sqrt(1.0f - nx * nx - ny * ny);
            uint8_t* p = &tx.px[4 * (y * 64 + x)];
            p[0] = (uint8_t)(127.5f + 127.5f * nx); p[1] = (uint8_t)(127.5f + 127.5f * ny); p[2] = (uint8_t)(127.5f + 127.5f * nz); p[3] = 255;
        }
    }
    { // gloss (roughness) map (linear)
        MqHostTex& tx = mq_ctx_tex(ctx, TEX_GLOSS); tx.w = tx.h = 64; tx.flags = MQ_TEX_LINEAR; tx.px.resize(64 * 64 * 4);
        for (int i = 0; i < 64 * 64; i++) { uint8_t v = (uint8_t)(60 + rng.range(120)); tx.px[4 * i] = tx.px[4 * i + 1] = tx.px[4 * i + 2] = v; tx.px[4 * i + 3] = 255; }
    }
}

} // namespace

bool mq_synth_generate(mq_ctx* ctx, const char* name, uint32_t seed, std::string& err) {
    int G; float q; float outdoor_frac; float sun_k; float mu_t = 0.0f;
    bool lamps = false;
    bool material_zoo = false; // panels of every material class of raytrace.glsl:95-119,198-204,246-311 in every room
    if (!strcmp(name, "synth_materials")) { G = 2; q = 128.0f; outdoor_frac = 0.25f; sun_k = 3.0f; mu_t = 1e-3f; material_zoo = true; }
    else if (!strcmp(name, "synth_start")) { G = 4; q = 32.0f; outdoor_frac = 0.0f; sun_k = 0.0f; }
    else if (!strcmp(name, "synth_tiny")) { G = 2; q = 128.0f; outdoor_frac = 0.25f; sun_k = 3.0f; }
    else if (!strcmp(name, "synth_lamps")) { G = 3; q = 48.0f; outdoor_frac = 0.0f; sun_k = 0.0f; lamps = true; } // indoor, ONE small ceiling light per room: the lighting guiding is made for
    else if (!strcmp(name, "synth_tiny_fog")) { G = 2; q = 128.0f; outdoor_frac = 0.25f; sun_k = 3.0f; mu_t = 2e-3f; }
    else if (!strcmp(name, "synth_start_fog")) { G = 4; q = 32.0f; outdoor_frac = 0.2f; sun_k = 4.0f; mu_t = 2e-3f; }
    else if (!strcmp(name, "synth_sepulcher")) { G = 8; q = 16.0f; outdoor_frac = 0.4f; sun_k = 4.0f; }
    else if (!strcmp(name, "synth_tears")) { G = 8; q = 16.0f; outdoor_frac = 0.4f; sun_k = 6.0f; mu_t = 2e-3f; }
    else if (!strcmp(name, "synth_azad")) { G = 11; q = 16.0f; outdoor_frac = 0.5f; sun_k = 4.0f; }
    else { err = std::string("unknown synthetic scene: ") + name; return false; }

    mq_ctx_clear_scene(ctx);
    Gen gen(seed ? seed : 1u);
    gen.q = q;
    gen.world.g = &mq_ctx_geo(ctx, 0); gen.alpha.g = &mq_ctx_geo(ctx, 1); gen.dyn.g = &mq_ctx_geo(ctx, 2);
    gen.world.g->flags = MQ_GEO_OPAQUE | MQ_GEO_STATIC; // selector 1, quake_node.cpp:863-871
    gen.alpha.g->flags = MQ_GEO_STATIC;                  // selector 2 (alpha tested), quake_node.cpp:884-892
    gen.dyn.g->flags = MQ_GEO_OPAQUE;                    // per-frame geometry, quake_node.cpp:896-983
    if (lamps) { gen.light_frac_ceiling = -1.0f; gen.light_frac_wall = 0.0f; }
    Rng& rng = gen.rng;
    make_textures(ctx, rng);

    const float S = 512.0f, T = 16.0f, door_w = 128.0f, door_h = 128.0f;
    // maze: random spanning tree + extra doors.  door_e[c] = door on +x wall, door_n[c] = door on +y wall
    std::vector<uint8_t> door_e(G * G, 0), door_n(G * G, 0), vis(G * G, 0), outdoor(G * G, 0);
    std::vector<float> height(G * G);
    std::vector<int> tour; // Euler tour of the spanning tree (camera path)
    {
        std::vector<int> stack; stack.push_back(0); vis[0] = 1; tour.push_back(0);
        while (!stack.empty()) {
            int c = stack.back(); int cx = c % G, cy = c / G;
            int nb[4], nn = 0;
            if (cx + 1 < G && !vis[c + 1]) nb[nn++] = c + 1;
            if (cx > 0 && !vis[c - 1]) nb[nn++] = c - 1;
            if (cy + 1 < G && !vis[c + G]) nb[nn++] = c + G;
            if (cy > 0 && !vis[c - G]) nb[nn++] = c - G;
            if (!nn) { stack.pop_back(); if (!stack.empty()) tour.push_back(stack.back()); continue; }
            int t = nb[rng.range(nn)];
            if (t == c + 1) door_e[c] = 1; else if (t == c - 1) door_e[t] = 1; else if (t == c + G) door_n[c] = 1; else door_n[t] = 1;
            vis[t] = 1; stack.push_back(t); tour.push_back(t);
        }
        for (int c = 0; c < G * G; c++) {
            if (c % G + 1 < G && rng.uni() < 0.3f) door_e[c] = 1;
            if (c / G + 1 < G && rng.uni() < 0.3f) door_n[c] = 1;
            outdoor[c] = rng.uni() < outdoor_frac;
            height[c] = outdoor[c] ? 640.0f : 192.0f + 64.0f * rng.range(3);
        }
    }
    for (int c = 0; c < G * G; c++) {
        int cx = c % G, cy = c / G;
        V org = {cx * S, cy * S, 0.0f};
        float x0 = org.x + T / 2, x1 = org.x + S - T / 2, y0 = org.y + T / 2, y1 = org.y + S - T / 2, H = height[c];
        int tex_floor = 1 + rng.range(16), tex_wall = 1 + rng.range(16), tex_ceil = 1 + rng.range(16);
        bool fancy = rng.uni() < 0.15f;
        gen.rect(gen.world, {x0, y0, 0}, {x1 - x0, 0, 0}, {0, y1 - y0, 0}, {0, 0, 1}, tex_floor, 0.0f, org, fancy);
        if (outdoor[c]) gen.rect(gen.world, {x0, y0, H}, {x1 - x0, 0, 0}, {0, y1 - y0, 0}, {0, 0, -1}, tex_ceil, 0.0f, org, false, MQ_MAT_FLAGS_SKY);
        else gen.rect(gen.world, {x0, y0, H}, {x1 - x0, 0, 0}, {0, y1 - y0, 0}, {0, 0, -1}, tex_ceil, gen.light_frac_ceiling, org, false);
        bool de = cx + 1 < G && door_e[c], dw = cx > 0 && door_e[c - 1], dn = cy + 1 < G && door_n[c], ds = cy > 0 && door_n[c - G];
        gen.wall({x1, y0, 0}, {0, y1 - y0, 0}, H, {-1, 0, 0}, tex_wall, de, door_w, door_h, org, fancy);
        gen.wall({x0, y0, 0}, {0, y1 - y0, 0}, H, {1, 0, 0}, tex_wall, dw, door_w, door_h, org, fancy);
        gen.wall({x0, y1, 0}, {x1 - x0, 0, 0}, H, {0, -1, 0}, tex_wall, dn, door_w, door_h, org, fancy);
        gen.wall({x0, y0, 0}, {x1 - x0, 0, 0}, H, {0, 1, 0}, tex_wall, ds, door_w, door_h, org, fancy);
        // door tunnels (owned by the cell on the low side)
        if (de) {
            float ya = org.y + 0.5f * (S - door_w), yb = ya + door_w, xa = x1, xb = x1 + T;
            gen.rect(gen.world, {xa, ya, 0}, {xb - xa, 0, 0}, {0, yb - ya, 0}, {0, 0, 1}, tex_floor, 0.0f, org, false);
            gen.rect(gen.world, {xa, ya, door_h}, {xb - xa, 0, 0}, {0, yb - ya, 0}, {0, 0, -1}, tex_wall, 0.0f, org, false);
            gen.rect(gen.world, {xa, ya, 0}, {xb - xa, 0, 0}, {0, 0, door_h}, {0, 1, 0}, tex_wall, 0.0f, org, false);
            gen.rect(gen.world, {xa, yb, 0}, {xb - xa, 0, 0}, {0, 0, door_h}, {0, -1, 0}, tex_wall, 0.0f, org, false);
            if (rng.uni() < 0.25f) { // grate across the tunnel, both facings, alpha tested
                mq_ext e; memset(&e, 0, sizeof e); e.texnum_alpha = (uint16_t)TEX_GRATE; e.n1_brush = 0xffffffffu; // alpha nibble 0 = use texture alpha
                float xm = 0.5f * (xa + xb);
                float st[8] = {0, 0, 2, 0, 2, 2, 0, 2};
                gen.alpha.quad({xm, ya, 0}, {xm, yb, 0}, {xm, yb, door_h}, {xm, ya, door_h}, {-1, 0, 0}, st, e);
                gen.alpha.quad({xm, ya, 0}, {xm, yb, 0}, {xm, yb, door_h}, {xm, ya, door_h}, {1, 0, 0}, st, e);
            }
        }
        if (dn) {
            float xa = org.x + 0.5f * (S - door_w), xb = xa + door_w, ya = y1, yb = y1 + T;
            gen.rect(gen.world, {xa, ya, 0}, {xb - xa, 0, 0}, {0, yb - ya, 0}, {0, 0, 1}, tex_floor, 0.0f, org, false);
            gen.rect(gen.world, {xa, ya, door_h}, {xb - xa, 0, 0}, {0, yb - ya, 0}, {0, 0, -1}, tex_wall, 0.0f, org, false);
            gen.rect(gen.world, {xa, ya, 0}, {0, yb - ya, 0}, {0, 0, door_h}, {1, 0, 0}, tex_wall, 0.0f, org, false);
            gen.rect(gen.world, {xb, ya, 0}, {0, yb - ya, 0}, {0, 0, door_h}, {-1, 0, 0}, tex_wall, 0.0f, org, false);
        }
        // pillars at the quarter points, crates in the corners
        int np = rng.range(5);
        for (int k = 0; k < np; k++) {
            float pxo = (k & 1) ? 128.0f : -128.0f, pyo = (k & 2) ? 128.0f : -128.0f;
            gen.pillar({org.x + S / 2 + pxo, org.y + S / 2 + pyo, 0}, 24.0f + 8.0f * rng.range(3), std::min(H, 640.0f), 1 + rng.range(16), org);
        }
        int ncr = rng.range(3);
        for (int k = 0; k < ncr; k++) {
            float sz = 32.0f + 16.0f * rng.range(3);
            float bx = (k & 1) ? x1 - sz - 8 : x0 + 8, by = rng.uni() < 0.5f ? y1 - sz - 8 : y0 + 8;
            gen.box(gen.world, {bx, by, 0}, {bx + sz, by + sz, sz}, 1 + rng.range(16), false, org);
        }
        if (material_zoo) { // free-standing double-sided panels, one per material class, off the camera path
            struct Panel { int flags; int alpha; bool alpha_geo; int kind; }; // kind 0 textured brush, 1 solid particle colours, 2 alias-style (encoded vertex normals)
            const Panel panels[] = {
                {MQ_MAT_FLAGS_LAVA, 15, false, 0}, {MQ_MAT_FLAGS_SLIME, 15, false, 0}, {MQ_MAT_FLAGS_TELE, 15, false, 0}, {MQ_MAT_FLAGS_WATER, 15, false, 0},
                {MQ_MAT_FLAGS_WATERFALL, 15, false, 0}, {MQ_MAT_FLAGS_SPRITE, 15, false, 0}, {MQ_MAT_FLAGS_SOLID, 15, false, 1}, {0, 15, false, 2},
                {0, 3, true, 0},                      // vertex alpha (3 - 1) / 14 < 0.666: rays pass (raytrace.glsl:108-110)
                {0, 12, true, 0},                     // vertex alpha (12 - 1) / 14 >= 0.666: rays stop
                {0, 0, true, 0},                      // alpha from the texture (the grate)
                {MQ_MAT_FLAGS_WATER, 3, true, 0},     // liquid in the alpha-tested set: always confirmed (flags in [1, 6], :104-106)
            };
            const int np = (int)(sizeof panels / sizeof panels[0]);
            for (int k = 0; k < np; k++) {
                const Panel& pn = panels[k];
                float px0 = org.x + 48.0f + 34.0f * k, py = org.y + 112.0f + 36.0f * (k & 1), z0 = 8.0f, z1 = 88.0f, w = 30.0f;
                mq_ext e; memset(&e, 0, sizeof e);
                int tex = pn.alpha == 0 ? (int)TEX_GRATE : 1 + (k % 16);
                e.texnum_alpha = (uint16_t)(tex | ((uint32_t)pn.alpha << 12));
                e.texnum_fb_flags = (uint16_t)((uint32_t)pn.flags << 12);
                if (pn.kind == 0) e.n1_brush = 0xffffffffu;
                else if (pn.kind == 1) { e.n0_gloss_norm = 0x00c08040u + 0x010101u * (uint32_t)k; e.n1_brush = 0x00204080u; } // albedo / emission colour bytes (:275-278)
                else { e.n0_gloss_norm = 0x12345678u; e.n1_brush = 0x23456789u; e.n2 = 0x3456789au; }                          // encoded vertex normals (unused by the shader, :279-285)
                float st[8] = {0, 0, 0.5f, 0, 0.5f, 1.25f, 0, 1.25f};
                Mesh& m = pn.alpha_geo ? gen.alpha : gen.world;
                m.quad({px0, py, z0}, {px0 + w, py, z0}, {px0 + w, py, z1}, {px0, py, z1}, {0, -1, 0}, st, e);
                m.quad({px0, py, z0}, {px0 + w, py, z0}, {px0 + w, py, z1}, {px0, py, z1}, {0, 1, 0}, st, e);
            }
        }
        if (rng.uni() < 0.2f) { // a "moving" box: dynamic slot, prev_vtx offset by its per-frame velocity
            V vel = {2.0f * (rng.uni() - 0.5f), 2.0f * (rng.uni() - 0.5f), 0.0f};
            float bx = org.x + S / 2 + 56 + 40 * rng.uni(), by = org.y + S / 2 + 56 + 40 * rng.uni(); // off the camera path
            gen.box(gen.dyn, {bx, by, 40}, {bx + 40, by + 40, 80}, 1 + rng.range(16), true, org, &vel);
        }
    }
    mq_constants& cst = mq_ctx_constants(ctx);
    float sd = 1.0f / std::sqrt(3.0f);
    cst.sun_direction[0] = cst.sun_direction[1] = cst.sun_direction[2] = sd; // quake_node.cpp:297,312
    cst.sun_color[0] = cst.sun_color[1] = cst.sun_color[2] = sun_k;
    cst.fov = 90.0f; cst.fov_tan_alpha_half = 1.0f; // tan(rad(90)/2), quake_node.cpp:765
    cst.volume_max_t = 10000.0f;                    // default_config.json:396
    MqSynthInfo& si = mq_ctx_synth(ctx);
    si.valid = true; si.eye_height = 56.0f; si.speed = 0.02f; si.mu_t = mu_t;
    float fog[3] = {0.55f, 0.6f, 0.7f};
    for (int k = 0; k < 3; k++) si.mu_s[k] = std::pow(fog[k], 1.0f / 1.2f) * mu_t; // quake_node.cpp:806-814
    si.path.clear();
    for (size_t i = 0; i < tour.size(); i++) {
        int c = tour[i];
        si.path.push_back((c % G) * S + S / 2); si.path.push_back((c / G) * S + S / 2); si.path.push_back(si.eye_height);
    }
    if (si.path.size() < 6) { si.path.push_back(S / 2 + 1); si.path.push_back(S / 2); si.path.push_back(si.eye_height); }
    return true;
}
