// mq_bsp.cpp -- Quake BSP29 / BSP2 world-model ingestion.
//
// Restates, for a .bsp file on disk, what the reference obtains from its quakespasm fork:
//   brush faces -> fan triangles + VertexExtraData   src/game/quake_helpers.cpp:362-469
//   opaque / alpha-tested geometry split              src/game/quake_node.cpp:847-894
//   texture upload (sRGB RGBA8, fullbright mask)      src/game/quake_node.cpp:683-704
//   worldspawn sun keys                               src/game/quake_node.cpp:231-313
// The on-disk layouts are id Software's BSP29 and the BSP2 extension (32-bit face/edge indices).
// Model 0 (the world) becomes the static geometry; the other brush models (doors, platforms, ...) are kept in model
// space for mq_dyn_add_brush_model (mq_producers.cpp), which places them per frame as add_geo_brush does for an entity.
// External normal / gloss maps (`<texture>_norm.tga`, `<texture>_gloss.tga` in a textures/ directory beside or above the
// map, as the reference's quakespasm fork loads them: t->norm / t->gloss, quake_helpers.cpp:428-431) are uploaded
// linear (quake_node.hpp:93-95) and referenced from VertexExtraData::n0_gloss_norm.
#include "mq_host.h"

#include <cmath>
#include <cstdio>
#include <cstring>
#include <map>

namespace {

struct Lump { int32_t ofs, len; };
struct TexInfo { float vecs[2][4]; int32_t miptex, flags; };
struct Model { float mins[3], maxs[3], origin[3]; int32_t headnode[4], visleafs, firstface, numfaces; };
struct MipHdr { char name[16]; uint32_t width, height, offsets[4]; };

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

uint16_t f2h_host(float f) {
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

// uncompressed / RLE true-colour TGA (types 2, 10; 24 or 32 bits) -> RGBA8, top row first
bool read_tga(const char* path, uint32_t& w, uint32_t& h, std::vector<uint8_t>& rgba) {
    std::vector<uint8_t> f;
    if (!read_file(path, f) || f.size() < 18) return false;
    const uint8_t idlen = f[0], cmap = f[1], type = f[2], bpp = f[16], desc = f[17];
    w = f[12] | (f[13] << 8); h = f[14] | (f[15] << 8);
    if (cmap || (type != 2 && type != 10) || (bpp != 24 && bpp != 32) || !w || !h || w > 8192 || h > 8192) return false;
    const size_t bytes = bpp / 8, n = (size_t)w * h;
    size_t at = 18 + idlen;
    std::vector<uint8_t> px(n * 4);
    auto put = [&](size_t i, const uint8_t* p) { px[4 * i] = p[2]; px[4 * i + 1] = p[1]; px[4 * i + 2] = p[0]; px[4 * i + 3] = bytes == 4 ? p[3] : 255; };
    if (type == 2) { if (at + n * bytes > f.size()) return false; for (size_t i = 0; i < n; i++) put(i, &f[at + i * bytes]); }
    else for (size_t i = 0; i < n;) {
        if (at >= f.size()) return false;
        const uint8_t c = f[at++]; const size_t run = (size_t)(c & 0x7f) + 1;
        if (c & 0x80) { if (at + bytes > f.size()) return false; for (size_t k = 0; k < run && i < n; k++) put(i++, &f[at]); at += bytes; }
        else { if (at + run * bytes > f.size()) return false; for (size_t k = 0; k < run && i < n; k++) put(i++, &f[at + k * bytes]); at += run * bytes; }
    }
    rgba.resize(n * 4);
    const bool top_first = (desc & 0x20) != 0;
    for (uint32_t y = 0; y < h; y++) memcpy(&rgba[4 * (size_t)y * w], &px[4 * (size_t)(top_first ? y : h - 1 - y) * w], 4 * (size_t)w);
    return true;
}

// first { ... } block of the entity lump -> key/value map (COM_Parse semantics, quake_node.cpp:241-264)
std::vector<std::map<std::string, std::string>> parse_entities(const char* s, size_t n) {
    std::vector<std::map<std::string, std::string>> ents;
    size_t i = 0;
    auto token = [&](std::string& t) -> bool {
        t.clear();
        while (i < n && (s[i] == ' ' || s[i] == '\n' || s[i] == '\t' || s[i] == '\r' || s[i] == 0)) i++;
        if (i >= n) return false;
        if (s[i] == '"') { i++; while (i < n && s[i] != '"') t.push_back(s[i++]); if (i < n) i++; return true; }
        if (s[i] == '{' || s[i] == '}') { t.push_back(s[i++]); return true; }
        const size_t start = i;
        while (i < n && (unsigned char)s[i] > ' ' && s[i] != '{' && s[i] != '}') t.push_back(s[i++]);
        if (i == start) i++; // a control byte outside a quoted string (damaged lump): skip it, always advance
        return true;
    };
    std::string t;
    while (token(t)) {
        if (t != "{") continue;
        std::map<std::string, std::string> e;
        for (;;) {
            std::string k, v;
            if (!token(k) || k == "}") break;
            if (!token(v)) break;
            if (!k.empty() && k[0] == '_') k = k.substr(1);
            while (!k.empty() && k.back() == ' ') k.pop_back();
            e[k] = v;
        }
        ents.push_back(e);
    }
    return ents;
}

} // namespace

bool mq_bsp_load(mq_ctx* ctx, const char* bsp_path, const char* palette_path, std::string& err) {
    std::vector<uint8_t> file;
    if (!read_file(bsp_path, file)) { err = std::string("cannot read ") + bsp_path; return false; }
    if (file.size() < 4 + 15 * 8) { err = "file too small for a BSP header"; return false; }
    int32_t version; memcpy(&version, file.data(), 4);
    bool bsp2 = !memcmp(file.data(), "BSP2", 4);
    if (!bsp2 && version != 29) { err = "unsupported BSP version (need 29 or BSP2)"; return false; }
    Lump lumps[15]; memcpy(lumps, file.data() + 4, sizeof lumps);
    for (auto& l : lumps) if (l.ofs < 0 || l.len < 0 || (size_t)l.ofs + (size_t)l.len > file.size()) { err = "lump out of range"; return false; }
    auto lump_ptr = [&](int i) { return file.data() + lumps[i].ofs; };

    uint8_t pal[768];
    for (int i = 0; i < 256; i++) pal[3 * i] = pal[3 * i + 1] = pal[3 * i + 2] = (uint8_t)i; // grey ramp stand-in
    if (palette_path && *palette_path) {
        std::vector<uint8_t> p;
        if (!read_file(palette_path, p) || p.size() < 768) { err = std::string("cannot read palette ") + palette_path; return false; }
        memcpy(pal, p.data(), 768);
    }

    mq_ctx_clear_scene(ctx);
    // ---- textures (lump 2) ---------------------------------------------------------------------
    struct TexMeta { std::string name; uint32_t w = 0, h = 0; uint32_t texnum = 0, fb = 0, norm = 0, gloss = 0; bool alpha = false, sky = false, turb = false; int turb_flag = 0; };
    std::string dir = bsp_path; { size_t sl = dir.find_last_of('/'); dir = sl == std::string::npos ? std::string(".") : dir.substr(0, sl); }
    uint32_t next_tex = 1;
    auto external = [&](const std::string& name, const char* suffix) -> uint32_t { // <name><suffix>.tga under textures/ beside or above the map; '*' of liquid names is '#' on disk
        std::string base = name; for (char& ch : base) if (ch == '*') ch = '#';
        for (const char* sub : {"/textures/", "/../textures/"}) {
            uint32_t w = 0, h = 0; std::vector<uint8_t> px;
            if (!read_tga((dir + sub + base + suffix + ".tga").c_str(), w, h, px) || next_tex + 1 >= MQ_MAX_GLTEXTURES) continue;
            const uint32_t tn = next_tex++;
            MqHostTex& t = mq_ctx_tex(ctx, tn); t.w = w; t.h = h; t.flags = MQ_TEX_LINEAR | MQ_TEX_MIPMAP; t.px = px; // not sRGB: quake_node.hpp:93-95
            return tn;
        }
        return 0u;
    };
    std::vector<TexMeta> metas;
    uint32_t sky_back = 0xffffu, sky_front = 0xffffu;
    if (lumps[2].len >= 4) {
        const uint8_t* base = lump_ptr(2);
        int32_t nmip; memcpy(&nmip, base, 4);
        if (nmip < 0 || (size_t)nmip * 4 + 4 > (size_t)lumps[2].len) { err = "bad miptex directory"; return false; }
        metas.resize((size_t)nmip);
        for (int i = 0; i < nmip; i++) {
            int32_t ofs; memcpy(&ofs, base + 4 + 4 * i, 4);
            TexMeta& m = metas[(size_t)i];
            if (ofs < 0 || (size_t)ofs + sizeof(MipHdr) > (size_t)lumps[2].len) continue; // missing texture (external)
            MipHdr h; memcpy(&h, base + ofs, sizeof h);
            char nm[17]; memcpy(nm, h.name, 16); nm[16] = 0;
            m.name = nm; m.w = h.width; m.h = h.height;
            if (!m.w || !m.h || m.w > 4096 || m.h > 4096) { m.w = m.h = 0; continue; }
            size_t px = (size_t)m.w * m.h;
            bool have_px = h.offsets[0] && (size_t)ofs + h.offsets[0] + px <= (size_t)lumps[2].len;
            const uint8_t* src = have_px ? base + ofs + h.offsets[0] : nullptr;
            m.alpha = m.name[0] == '{';
            m.sky = !strncasecmp(m.name.c_str(), "sky", 3);
            m.turb = m.name[0] == '*';
            if (m.turb) {
                if (!strncasecmp(m.name.c_str() + 1, "lava", 4)) m.turb_flag = 1;       // MAT_FLAGS_LAVA
                else if (!strncasecmp(m.name.c_str() + 1, "slime", 5)) m.turb_flag = 2; // MAT_FLAGS_SLIME
                else if (!strncasecmp(m.name.c_str() + 1, "tele", 4)) m.turb_flag = 3;  // MAT_FLAGS_TELE
                else m.turb_flag = 4;                                                   // MAT_FLAGS_WATER
            }
            if (next_tex + 2 >= MQ_MAX_GLTEXTURES) { err = "too many textures"; return false; }
            if (m.sky && src && m.w == 2 * m.h) { // classic sky: right half back layer, left half front layer (index 0 transparent)
                uint32_t hw = m.w / 2;
                for (int layer = 0; layer < 2; layer++) {
                    uint32_t tn = next_tex++;
                    MqHostTex& t = mq_ctx_tex(ctx, tn); t.w = hw; t.h = m.h; t.flags = MQ_TEX_SRGB | MQ_TEX_LINEAR; t.px.resize((size_t)hw * m.h * 4);
                    for (uint32_t y = 0; y < m.h; y++) for (uint32_t x = 0; x < hw; x++) {
                        uint8_t ci = src[(size_t)y * m.w + x + (layer == 0 ? hw : 0)];
                        uint8_t* d = &t.px[4 * ((size_t)y * hw + x)];
                        d[0] = pal[3 * ci]; d[1] = pal[3 * ci + 1]; d[2] = pal[3 * ci + 2]; d[3] = (layer == 1 && ci == 0) ? 0 : 255;
                    }
                    if (layer == 0) sky_back = tn; else sky_front = tn;
                }
                m.texnum = sky_back;
                continue;
            }
            m.texnum = next_tex++;
            MqHostTex& t = mq_ctx_tex(ctx, m.texnum); t.w = m.w; t.h = m.h; t.flags = MQ_TEX_SRGB | (m.turb ? 0u : MQ_TEX_MIPMAP); t.px.resize(px * 4); // brush textures carry TEXPREF_MIPMAP, warped liquids do not
            bool any_fb = false;
            for (size_t k = 0; k < px; k++) {
                uint8_t ci = src ? src[k] : (uint8_t)(((k / m.w) ^ (k % m.w)) & 8 ? 96 : 160);
                uint8_t* d = &t.px[4 * k];
                d[0] = pal[3 * ci]; d[1] = pal[3 * ci + 1]; d[2] = pal[3 * ci + 2];
                d[3] = (m.alpha && ci == 255) ? 0 : 255;
                if (ci >= 224 && !(m.alpha && ci == 255)) any_fb = true;
            }
            if (!m.turb && !m.sky) { m.norm = external(m.name, "_norm"); m.gloss = external(m.name, "_gloss"); }
            if (any_fb && src && !m.turb) { // fullbright mask texture: non-fullbright texels are black
                m.fb = next_tex++;
                MqHostTex& f = mq_ctx_tex(ctx, m.fb); f.w = m.w; f.h = m.h; f.flags = MQ_TEX_SRGB | MQ_TEX_MIPMAP; f.px.assign(px * 4, 0);
                for (size_t k = 0; k < px; k++) {
                    uint8_t ci = src[k];
                    if (ci >= 224 && !(m.alpha && ci == 255)) { uint8_t* d = &f.px[4 * k]; d[0] = pal[3 * ci]; d[1] = pal[3 * ci + 1]; d[2] = pal[3 * ci + 2]; d[3] = 255; }
                }
            }
        }
    }
    // ---- geometry ------------------------------------------------------------------------------
    // typed lumps are copied out: a lump may start at any byte offset of the file
    const size_t nverts = (size_t)lumps[3].len / 12, ntexinfo = (size_t)lumps[6].len / sizeof(TexInfo), nsurfedges = (size_t)lumps[13].len / 4;
    std::vector<float> verts_v(3 * nverts + 1); std::vector<TexInfo> texinfo_v(ntexinfo + 1); std::vector<int32_t> surfedges_v(nsurfedges + 1);
    memcpy(verts_v.data(), lump_ptr(3), 12 * nverts); memcpy((void*)texinfo_v.data(), lump_ptr(6), sizeof(TexInfo) * ntexinfo); memcpy(surfedges_v.data(), lump_ptr(13), 4 * nsurfedges);
    const float* verts = verts_v.data(); const TexInfo* texinfo = texinfo_v.data(); const int32_t* surfedges = surfedges_v.data();
    size_t edge_sz = bsp2 ? 8 : 4, nedges = (size_t)lumps[12].len / edge_sz;
    size_t face_sz = bsp2 ? 28 : 20, nfaces = (size_t)lumps[7].len / face_sz;
    if ((size_t)lumps[14].len < sizeof(Model)) { err = "no models"; return false; }
    Model world; memcpy(&world, lump_ptr(14), sizeof world);
    if (world.firstface < 0 || world.numfaces < 0 || (size_t)world.firstface + (size_t)world.numfaces > nfaces) { err = "world faces out of range"; return false; }
    MqHostGeo& opaque = mq_ctx_geo(ctx, 0); MqHostGeo& alpha = mq_ctx_geo(ctx, 1);
    opaque.flags = MQ_GEO_OPAQUE | MQ_GEO_STATIC; alpha.flags = MQ_GEO_STATIC;
    auto edge_vert = [&](int32_t lindex) -> int64_t {
        size_t e = (size_t)(lindex >= 0 ? lindex : -lindex);
        if (e >= nedges) return -1;
        const uint8_t* p = lump_ptr(12) + e * edge_sz;
        uint32_t v0, v1;
        if (bsp2) { memcpy(&v0, p, 4); memcpy(&v1, p + 4, 4); } else { uint16_t a, b; memcpy(&a, p, 2); memcpy(&b, p + 2, 2); v0 = a; v1 = b; }
        return lindex >= 0 ? v0 : v1;
    };
    auto emit_face = [&](int32_t face, MqHostGeo& opaque, MqHostGeo& alpha) {
        const uint8_t* fp = lump_ptr(7) + (size_t)face * face_sz;
        int32_t firstedge, numedges, ti;
        if (bsp2) { memcpy(&firstedge, fp + 8, 4); memcpy(&numedges, fp + 12, 4); memcpy(&ti, fp + 16, 4); }
        else { int16_t ne, t16; memcpy(&firstedge, fp + 4, 4); memcpy(&ne, fp + 8, 2); memcpy(&t16, fp + 10, 2); numedges = ne; ti = t16; }
        if (numedges < 3 || firstedge < 0 || (size_t)firstedge + (size_t)numedges > nsurfedges || ti < 0 || (size_t)ti >= ntexinfo) return;
        const TexInfo& tx = texinfo[ti];
        if (tx.miptex < 0 || (size_t)tx.miptex >= metas.size()) return;
        const TexMeta& m = metas[(size_t)tx.miptex];
        if (!strcasecmp(m.name.c_str(), "skip")) return; // quake_helpers.cpp:391
        MqHostGeo& g = m.alpha ? alpha : opaque;
        uint32_t base = (uint32_t)(g.vtx.size() / 3);
        std::vector<float> st((size_t)numedges * 2);
        bool ok = true;
        for (int32_t k = 0; k < numedges; k++) {
            int64_t vi = edge_vert(surfedges[firstedge + k]);
            if (vi < 0 || (size_t)vi >= nverts) { ok = false; break; }
            const float* v = verts + 3 * vi;
            for (int a = 0; a < 3; a++) { g.vtx.push_back(v[a]); g.prev_vtx.push_back(v[a]); }
            float s = v[0] * tx.vecs[0][0] + v[1] * tx.vecs[0][1] + v[2] * tx.vecs[0][2] + tx.vecs[0][3];
            float t = v[0] * tx.vecs[1][0] + v[1] * tx.vecs[1][1] + v[2] * tx.vecs[1][2] + tx.vecs[1][3];
            st[2 * k] = m.w ? s / (float)m.w : 0.0f; st[2 * k + 1] = m.h ? t / (float)m.h : 0.0f;
        }
        if (!ok) { g.vtx.resize(3 * (size_t)base); g.prev_vtx.resize(3 * (size_t)base); return; }
        for (int32_t k = 2; k < numedges; k++) { // fan, quake_helpers.cpp:419-423
            g.idx.push_back(base); g.idx.push_back(base + (uint32_t)k - 1); g.idx.push_back(base + (uint32_t)k);
            mq_ext e; memset(&e, 0, sizeof e);
            e.n1_brush = 0xffffffffu;
            e.n0_gloss_norm = (m.gloss & 0xffffu) | (m.norm << 16); // pack_uint32(gloss, norm), quake_helpers.cpp:428-431
            e.st[0] = f2h_host(st[0]); e.st[1] = f2h_host(st[1]);
            e.st[2] = f2h_host(st[2 * (k - 1)]); e.st[3] = f2h_host(st[2 * (k - 1) + 1]);
            e.st[4] = f2h_host(st[2 * k]); e.st[5] = f2h_host(st[2 * k + 1]);
            uint32_t flags = 0;
            if (m.texnum) {
                e.texnum_alpha = (uint16_t)(std::min<uint32_t>(m.texnum, MQ_MAX_GLTEXTURES - 1) | ((m.alpha ? 0u : 15u) << 12)); // quake_helpers.cpp:26-48
                e.texnum_fb_flags = (uint16_t)m.fb;
                if (m.turb) flags = (uint32_t)m.turb_flag;
                if (strstr(m.name.c_str(), "wfall")) flags = MQ_MAT_FLAGS_WATERFALL;
            }
            if (m.sky) flags = MQ_MAT_FLAGS_SKY;
            e.texnum_fb_flags = (uint16_t)((e.texnum_fb_flags & 0xfffu) | (flags << 12));
            g.ext.push_back(e);
        }
    };
    for (int32_t fi = 0; fi < world.numfaces; fi++) emit_face(world.firstface + fi, opaque, alpha);
    { // the other brush models, in model space, for the per-frame producer
        MqProducerState& P = mq_ctx_producers(ctx);
        const size_t nmodels = (size_t)lumps[14].len / sizeof(Model);
        P.bsp_models.assign(nmodels, MqHostGeo());
        for (size_t mi = 1; mi < nmodels; mi++) {
            Model sub; memcpy(&sub, lump_ptr(14) + mi * sizeof(Model), sizeof sub);
            if (sub.firstface < 0 || sub.numfaces < 0 || (size_t)sub.firstface + (size_t)sub.numfaces > nfaces) continue;
            for (int32_t fi = 0; fi < sub.numfaces; fi++) emit_face(sub.firstface + fi, P.bsp_models[mi], P.bsp_models[mi]);
        }
    }
    // ---- worldspawn + player start ---------------------------------------------------------------
    mq_constants& cst = mq_ctx_constants(ctx);
    cst.fov = 90.0f; cst.fov_tan_alpha_half = 1.0f; cst.volume_max_t = 1000.0f;
    float sun_col[3] = {0, 0, 0}, sun_dir[3] = {1, 1, 1};
    auto ents = parse_entities((const char*)lump_ptr(0), (size_t)lumps[0].len);
    MqSynthInfo& si = mq_ctx_synth(ctx);
    si = MqSynthInfo();
    if (!ents.empty()) {
        auto& ws = ents[0];
        auto lum = [](const float* c) { return c[0] * 0.299f + c[1] * 0.587f + c[2] * 0.114f; };
        for (const char* k : {"sunlight", "sunlight2", "sunlight3"}) {
            if (!ws.count(k)) continue;
            float col[3] = {1, 1, 1};
            std::string ck = std::string(k) + "_color";
            if (ws.count(ck)) sscanf(ws[ck].c_str(), "%f %f %f", &col[0], &col[1], &col[2]);
            float inten = (float)atoi(ws[k].c_str());
            for (float& c : col) c = c * inten / 4000.0f;
            if (lum(col) > lum(sun_col)) memcpy(sun_col, col, sizeof col);
        }
        if (ws.count("sun_mangle")) {
            float yaw = 0, pitch = 0, roll = 0;
            sscanf(ws["sun_mangle"].c_str(), "%f %f %f", &yaw, &pitch, &roll);
            yaw -= 180.0f; // quake_node.cpp:293
            float cy = std::cos(yaw * 0.01745329252f), sy = std::sin(yaw * 0.01745329252f), cp = std::cos(pitch * 0.01745329252f), sp = std::sin(pitch * 0.01745329252f);
            sun_dir[0] = cp * cy; sun_dir[1] = cp * sy; sun_dir[2] = -sp; // AngleVectors forward
        }
        if (ws.count("sky") && ws["sky"] == "stormydays_") { sun_dir[0] = 1; sun_dir[1] = -1; sun_dir[2] = 1; sun_col[0] = 6.6f; sun_col[1] = 6.0f; sun_col[2] = 5.4f; } // quake_node.cpp:301-305
    }
    float mx = std::max(sun_col[0], std::max(sun_col[1], sun_col[2]));
    if (mx > 20.0f) for (float& c : sun_col) c = c / mx * 20.0f; // MAX_SUN_COLOR, config.h:19
    float dl = std::sqrt(sun_dir[0] * sun_dir[0] + sun_dir[1] * sun_dir[1] + sun_dir[2] * sun_dir[2]);
    for (int k = 0; k < 3; k++) { cst.sun_color[k] = sun_col[k]; cst.sun_direction[k] = sun_dir[k] / dl; }
    for (auto& e : ents) {
        if (!e.count("classname") || e["classname"] != "info_player_start" || !e.count("origin")) continue;
        float o[3] = {0, 0, 0}, ang = 0;
        sscanf(e["origin"].c_str(), "%f %f %f", &o[0], &o[1], &o[2]);
        if (e.count("angle")) ang = (float)atof(e["angle"].c_str());
        si.valid = true; si.eye_height = 22.0f; si.speed = 0.0f;
        float dx = std::cos(ang * 0.01745329252f), dy = std::sin(ang * 0.01745329252f);
        for (int k = -1; k <= 2; k++) { si.path.push_back(o[0] + dx * (float)k); si.path.push_back(o[1] + dy * (float)k); si.path.push_back(o[2] + 22.0f); }
        break;
    }
    si.sky_rt_bk = (sky_back & 0xffffu) | ((sky_front & 0xffffu) << 16);
    return true;
}
