/*
 * orc_math.h -- deterministic scalar math + the shading helpers the reference takes from the
 * absent `merian-shaders` headers.  ORACLE ONLY (test infrastructure, see mq_oracle.h).
 *
 * Every function uses only IEEE-754 correctly-rounded +,-,*,/,sqrt and integer ops in a fixed
 * order (compile with -ffp-contract=off), so a device implementation that performs the same
 * operations in the same order is bit-identical.  No libm transcendental is called.
 */
#ifndef ORC_MATH_H
#define ORC_MATH_H

#include <math.h>
#include <stdint.h>
#include <string.h>

typedef struct { float x, y, z; } v3;

static inline uint32_t f2u(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static inline float u2f(uint32_t u) { float f; memcpy(&f, &u, 4); return f; }

static inline v3 V3(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 vadd(v3 a, v3 b) { return V3(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 vsub(v3 a, v3 b) { return V3(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 vmul(v3 a, v3 b) { return V3(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 vscale(v3 a, float s) { return V3(a.x * s, a.y * s, a.z * s); }
static inline float vdot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 vcross(v3 a, v3 b) {
    return V3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
static inline float vlen(v3 a) { return sqrtf(vdot(a, a)); }
static inline v3 vnormalize(v3 a) { float inv = 1.0f / sqrtf(vdot(a, a)); return vscale(a, inv); }
static inline v3 vneg(v3 a) { return V3(-a.x, -a.y, -a.z); }
/* IEEE maxNum/minNum semantics (a NaN operand is ignored), as GPU v_max_f32/v_min_f32 */
static inline float omax(float a, float b) { return fmaxf(a, b); }
static inline float omin(float a, float b) { return fminf(a, b); }
static inline float oclamp(float x, float lo, float hi) { return omin(omax(x, lo), hi); }
static inline float omix(float a, float b, float t) { return a * (1.0f - t) + b * t; }

/* ---- transcendental replacements ------------------------------------------------------------- */

/* 2^x. |rel err| ~ 2e-7.  x < -126 -> 0, x >= 128 -> +inf. */
static inline float orc_exp2(float x) {
    if (!(x >= -126.0f)) return (x != x) ? x : 0.0f;
    if (x >= 128.0f) return INFINITY;
    float n = floorf(x + 0.5f);
    float f = x - n;
    float y = f * 0.693147182464599609375f;
    float p = 1.0f / 5040.0f;
    p = p * y + 1.0f / 720.0f;
    p = p * y + 1.0f / 120.0f;
    p = p * y + 1.0f / 24.0f;
    p = p * y + 1.0f / 6.0f;
    p = p * y + 0.5f;
    p = p * y + 1.0f;
    p = p * y + 1.0f;
    int e = (int)n;
    /* n can be 128 when x in [127.5,128): split the scale to stay finite-exact */
    if (e > 127) return (p * 2.0f) * u2f((uint32_t)(e - 1 + 127) << 23);
    return p * u2f((uint32_t)(e + 127) << 23);
}

/* log2(x) for x > 0 (x <= 0 -> -inf, NaN -> NaN). */
static inline float orc_log2(float x) {
    if (x != x) return x;
    if (!(x > 0.0f)) return -INFINITY;
    if (x == INFINITY) return x;
    float bias = 0.0f;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; bias = -24.0f; }
    uint32_t b = f2u(x);
    int e = (int)(b >> 23) - 127;
    float m = u2f((b & 0x007fffffu) | 0x3f800000u);
    if (m > 1.41421354f) { m = m * 0.5f; e += 1; }
    float f = m - 1.0f;
    float s = f / (2.0f + f);
    float z = s * s;
    float p = 1.0f / 9.0f;
    p = p * z + 1.0f / 7.0f;
    p = p * z + 0.2f;
    p = p * z + 1.0f / 3.0f;
    p = p * z + 1.0f;
    float ln_m = 2.0f * s * p;
    return ((float)e + bias) + ln_m * 1.44269502162933349609375f;
}
static inline float orc_exp(float x) { return orc_exp2(x * 1.44269502162933349609375f); }
static inline float orc_log(float x) { return orc_log2(x) * 0.693147182464599609375f; }
/* x^y for x >= 0 */
static inline float orc_pow(float x, float y) {
    if (x == 0.0f) return (y == 0.0f) ? 1.0f : 0.0f;
    return orc_exp2(y * orc_log2(x));
}

/* (cos, sin)(2*pi*u) for any finite u (u is reduced to [0,1) first) */
static inline void orc_sincos2pi(float u, float* c_out, float* s_out) {
    u = u - floorf(u);
    float q = floorf(u * 4.0f + 0.5f);
    float r = u - q * 0.25f;
    float y = r * 6.283185482025146484375f;
    float y2 = y * y;
    float sp = 1.0f / 362880.0f;
    sp = sp * y2 - 1.0f / 5040.0f;
    sp = sp * y2 + 1.0f / 120.0f;
    sp = sp * y2 - 1.0f / 6.0f;
    sp = sp * y2 + 1.0f;
    float s = sp * y;
    float cp = 1.0f / 40320.0f;
    cp = cp * y2 - 1.0f / 720.0f;
    cp = cp * y2 + 1.0f / 24.0f;
    cp = cp * y2 - 0.5f;
    float c = cp * y2 + 1.0f;
    int qi = ((int)q) & 3;
    if (qi == 0) { *c_out = c; *s_out = s; }
    else if (qi == 1) { *c_out = -s; *s_out = c; }
    else if (qi == 2) { *c_out = -c; *s_out = -s; }
    else { *c_out = s; *s_out = -c; }
}
static inline float orc_sin(float x) {
    float c, s;
    orc_sincos2pi(x * 0.15915493667125701904296875f, &c, &s);
    return s;
}

/* ---- IEEE half <-> float, round-to-nearest-even, overflow -> inf, denormals kept ------------- */
static inline uint16_t orc_f2h(float f) {
    uint32_t x = f2u(f);
    uint32_t sign = (x >> 16) & 0x8000u;
    uint32_t ax = x & 0x7fffffffu;
    if (ax >= 0x7f800000u) return (uint16_t)(sign | (ax > 0x7f800000u ? 0x7e00u : 0x7c00u));
    if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u); /* rounds to >= 65520 -> inf */
    if (ax < 0x33000001u) return (uint16_t)sign;              /* <= 2^-25 -> 0 (ties to even) */
    int e = (int)(ax >> 23) - 127;
    uint32_t m = (ax & 0x7fffffu) | 0x800000u;
    int shift;
    uint32_t hexp;
    if (e < -14) { shift = 13 + (-14 - e); hexp = 0; }
    else { shift = 13; hexp = (uint32_t)(e + 15); }
    uint32_t half_m = m >> shift;
    uint32_t rem = m & ((1u << shift) - 1u);
    uint32_t halfway = 1u << (shift - 1);
    if (rem > halfway || (rem == halfway && (half_m & 1u))) half_m += 1;
    uint32_t h;
    if (hexp == 0) h = half_m;                         /* denormal (may carry into exp 1) */
    else h = ((hexp - 1) << 10) + half_m;              /* half_m has the implicit bit at 0x400 */
    return (uint16_t)(sign | h);
}
static inline float orc_h2f(uint16_t h) {
    uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
    uint32_t e = (h >> 10) & 0x1fu;
    uint32_t m = h & 0x3ffu;
    if (e == 0) {
        if (m == 0) return u2f(sign);
        float v = (float)m * 5.9604644775390625e-8f; /* m * 2^-24, exact */
        return (sign ? -v : v);
    }
    if (e == 31) return u2f(sign | 0x7f800000u | (m << 13));
    return u2f(sign | ((e + 112u) << 23) | (m << 13));
}
/* round a float to the nearest half and back ("stored in a float16_t variable") */
static inline float orc_rh(float f) { return orc_h2f(orc_f2h(f)); }
static inline v3 orc_rh3(v3 a) { return V3(orc_rh(a.x), orc_rh(a.y), orc_rh(a.z)); }

/* ---- RNG (merian-shaders/random.glsl, DEFINED here) ------------------------------------------ */
/* Jarzynski & Olano PCG-4D with the 16-bit xorshift; returns the x lane (never 0). */
static inline uint32_t orc_pcg4d16(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    (void)w;
    return x ? x : 0x9e3779b9u;
}
/* Marsaglia xorshift32 (13,17,5) -> uniform in [0,1) with 24 random bits */
static inline float orc_xorshift(uint32_t* s) {
    uint32_t v = *s;
    v ^= v << 13; v ^= v >> 17; v ^= v << 5;
    *s = v;
    return (float)(v >> 8) * 5.9604644775390625e-8f;
}

/* ---- 32-bit octahedral unit-vector codec (merian-shaders/normal_encode.glsl, DEFINED) -------- */
static inline float orc_sgn(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
static inline uint32_t orc_encode_normal(v3 n) {
    float inv = 1.0f / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    float px = n.x * inv, py = n.y * inv;
    if (n.z < 0.0f) {
        float tx = (1.0f - fabsf(py)) * orc_sgn(px);
        float ty = (1.0f - fabsf(px)) * orc_sgn(py);
        px = tx; py = ty;
    }
    /* snorm16 per axis: 0 and +-1 are exact, so axis-aligned normals survive the codec bit for bit */
    int qx = (int)floorf(oclamp(px, -1.0f, 1.0f) * 32767.0f + 0.5f);
    int qy = (int)floorf(oclamp(py, -1.0f, 1.0f) * 32767.0f + 0.5f);
    return ((uint32_t)qx & 0xffffu) | (((uint32_t)qy & 0xffffu) << 16);
}
static inline v3 orc_decode_normal(uint32_t e) {
    float px = (float)(int16_t)(e & 0xffffu) * (1.0f / 32767.0f);
    float py = (float)(int16_t)(e >> 16) * (1.0f / 32767.0f);
    float pz = 1.0f - fabsf(px) - fabsf(py);
    if (pz < 0.0f) {
        float tx = (1.0f - fabsf(py)) * orc_sgn(px);
        float ty = (1.0f - fabsf(px)) * orc_sgn(py);
        px = tx; py = ty;
    }
    return vnormalize(V3(px, py, pz));
}

/* orthonormal frame around n (Duff et al. 2017) */
static inline void orc_make_frame(v3 n, v3* t, v3* b) {
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float bb = n.x * n.y * a;
    *t = V3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    *b = V3(bb, sign + n.y * n.y * a, -n.y);
}

static inline float orc_luminance(v3 c) { return c.x * 0.299f + c.y * 0.587f + c.z * 0.114f; }

/* ---- von Mises-Fisher (merian-shaders/von_mises_fisher.glsl, DEFINED) ------------------------ */
#define ORC_INV_4PI 0.079577468335628509521484375f
#define ORC_INV_PI 0.3183098733425140380859375f
#define ORC_INV_2PI 0.15915493667125701904296875f
static inline float orc_vmf_pdf(v3 w, v3 mu, float kappa) {
    if (!(kappa > 1e-4f)) return ORC_INV_4PI;
    float e2k = orc_exp(-2.0f * kappa);
    return kappa * ORC_INV_2PI / (1.0f - e2k) * orc_exp(kappa * (vdot(mu, w) - 1.0f));
}
static inline v3 orc_vmf_sample(v3 mu, float kappa, float xi0, float xi1) {
    float wz;
    if (!(kappa > 1e-4f)) wz = 1.0f - 2.0f * xi0;
    else {
        float e2k = orc_exp(-2.0f * kappa);
        wz = 1.0f + orc_log(omax(xi0 + (1.0f - xi0) * e2k, 1e-37f)) / kappa;
    }
    wz = oclamp(wz, -1.0f, 1.0f);
    float sr = sqrtf(omax(1.0f - wz * wz, 0.0f));
    float c, s;
    orc_sincos2pi(xi1, &c, &s);
    v3 t, b;
    orc_make_frame(mu, &t, &b);
    return vadd(vadd(vscale(t, sr * c), vscale(b, sr * s)), vscale(mu, wz));
}

/* ---- BSDF: 50/50 mix of Lambert and GGX (merian-shaders/bsdf_ggx.glsl, DEFINED) --------------
 * wi is the direction of travel of the incoming ray (points INTO the surface), so the view vector
 * is v = -wi.  alpha = roughness^2. */
static inline float orc_roughness_to_alpha(float r) { return r * r; }
static inline float orc_ggx_D(float ndoth, float alpha) {
    float a2 = alpha * alpha;
    float d = ndoth * ndoth * (a2 - 1.0f) + 1.0f;
    return a2 * ORC_INV_PI / (d * d);
}
static inline float orc_ggx_G1(float ndotx, float alpha) {
    float a2 = alpha * alpha;
    return 2.0f * ndotx / (ndotx + sqrtf(a2 + (1.0f - a2) * ndotx * ndotx));
}
static inline v3 orc_bsdf_sample(v3 wi, v3 n, float alpha, float xi0, float xi1, float xi2) {
    v3 t, b;
    orc_make_frame(n, &t, &b);
    float c, s;
    orc_sincos2pi(xi1, &c, &s);
    if (xi2 < 0.5f) { /* cosine-weighted hemisphere */
        float r = sqrtf(xi0);
        float z = sqrtf(omax(1.0f - xi0, 0.0f));
        return vadd(vadd(vscale(t, r * c), vscale(b, r * s)), vscale(n, z));
    }
    float a2 = alpha * alpha;
    float ct2 = (1.0f - xi0) / (1.0f + (a2 - 1.0f) * xi0);
    float ct = sqrtf(ct2);
    float st = sqrtf(omax(1.0f - ct2, 0.0f));
    v3 h = vadd(vadd(vscale(t, st * c), vscale(b, st * s)), vscale(n, ct));
    float d = vdot(wi, h);
    return vsub(wi, vscale(h, 2.0f * d)); /* reflect(wi, h) */
}
static inline float orc_bsdf_pdf(v3 wi, v3 wo, v3 n, float alpha) {
    float ndoto = vdot(n, wo);
    if (!(ndoto > 0.0f)) return 0.0f;
    v3 v = vneg(wi);
    v3 hs = vadd(v, wo);
    float hl = vlen(hs);
    float pd = 0.5f * ndoto * ORC_INV_PI;
    if (!(hl > 1e-12f)) return pd;
    v3 h = vscale(hs, 1.0f / hl);
    float ndoth = vdot(n, h), vdoth = vdot(v, h);
    if (!(ndoth > 0.0f) || !(vdoth > 0.0f)) return pd;
    return pd + 0.5f * orc_ggx_D(ndoth, alpha) * ndoth / (4.0f * vdoth);
}
/* ((1-F(n.v))/pi + F(v.h) D G / (4 (n.v)(n.o))) * (n.o), Schlick F with F0; albedo excluded */
static inline float orc_bsdf_times_wodotn(v3 wi, v3 wo, v3 n, float alpha, float F0) {
    float ndoto = vdot(n, wo);
    v3 v = vneg(wi);
    float ndotv = vdot(n, v);
    if (!(ndoto > 0.0f) || !(ndotv > 0.0f)) return 0.0f;
    v3 hs = vadd(v, wo);
    float hl = vlen(hs);
    if (!(hl > 1e-12f)) return ndoto * ORC_INV_PI;
    v3 h = vscale(hs, 1.0f / hl);
    float ndoth = omax(vdot(n, h), 0.0f), vdoth = omax(vdot(v, h), 0.0f);
    float m = 1.0f - vdoth;
    float m2 = m * m;
    float F = F0 + (1.0f - F0) * (m2 * m2 * m);
    float mv = 1.0f - ndotv;
    float mv2 = mv * mv;
    float Fv = F0 + (1.0f - F0) * (mv2 * mv2 * mv); /* view-dependent Fresnel gates the diffuse lobe */
    float spec = F * orc_ggx_D(ndoth, alpha) * orc_ggx_G1(ndotv, alpha) * orc_ggx_G1(ndoto, alpha) /
                 (4.0f * ndotv * ndoto);
    return ((1.0f - Fv) * ORC_INV_PI + spec) * ndoto;
}

/* ---- hash grid (merian-shaders/grid.glsl + hash.glsl, DEFINED) ------------------------------- */
static inline uint32_t orc_hash_u32(uint32_t x) { /* lowbias32 */
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
static inline uint32_t orc_hash2_u32(uint32_t x) { /* murmur3 finalizer */
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
typedef struct { int32_t x, y, z; } i3;
static inline i3 orc_grid_idx_interpolate(v3 pos, float width, float xi) {
    i3 r;
    float inv = 1.0f / width;
    r.x = (int32_t)floorf(pos.x * inv + xi);
    r.y = (int32_t)floorf(pos.y * inv + xi);
    r.z = (int32_t)floorf(pos.z * inv + xi);
    return r;
}
/* floor(-log2(1 - xi)) for xi = k * 2^-24 (mc.glsl:70), evaluated exactly in integers:
 * 1 - xi = m * 2^-24 with m = 2^24 - k in [1, 2^24] */
static inline uint32_t orc_level_jitter(float xi) {
    uint32_t m = 16777216u - (uint32_t)(xi * 16777216.0f);
    uint32_t fl = 31u - (uint32_t)__builtin_clz(m); /* floor(log2 m) */
    return (m & (m - 1u)) ? 23u - fl : 24u - fl;
}
/* map a 32-bit hash to [0, size): multiply-high range reduction */
static inline uint32_t orc_reduce(uint32_t h, uint32_t size) { return (uint32_t)(((uint64_t)h * (uint64_t)size) >> 32); }
/* dominant axis + sign -> 0..5 */
static inline uint32_t orc_normal_face(v3 n) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax >= ay && ax >= az) return n.x < 0.0f ? 1u : 0u;
    if (ay >= az) return n.y < 0.0f ? 3u : 2u;
    return n.z < 0.0f ? 5u : 4u;
}
static inline uint32_t orc_hash3(i3 c, uint32_t salt) {
    return orc_hash_u32((uint32_t)c.x + orc_hash_u32((uint32_t)c.y + orc_hash_u32((uint32_t)c.z + salt)));
}
static inline uint32_t orc_hash_grid(i3 c, uint32_t size) { return orc_reduce(orc_hash3(c, 0x51ed270bu), size); }
static inline uint32_t orc_hash_grid_normal_level(i3 c, v3 n, uint32_t level, uint32_t size) {
    return orc_reduce(orc_hash3(c, orc_hash_u32(level * 8u + orc_normal_face(n) + 0x2545f491u)), size);
}
static inline uint32_t orc_hash2_3(i3 c, uint32_t salt) {
    return orc_hash2_u32((uint32_t)c.x * 0x9e3779b1u + orc_hash2_u32((uint32_t)c.y * 0x85ebca77u + orc_hash2_u32((uint32_t)c.z * 0xc2b2ae3du + salt)));
}
static inline uint32_t orc_hash2_grid(i3 c) { return orc_hash2_3(c, 0x27d4eb2fu); }
static inline uint32_t orc_hash2_grid_level(i3 c, uint32_t level) { return orc_hash2_3(c, 0x165667b1u + level); }

/* ---- transmittance / misc -------------------------------------------------------------------- */
static inline float orc_transmittance(float t, float mu_t, float tmax) {
    if (mu_t == 0.0f) return 1.0f;
    return orc_exp(-mu_t * omin(t, tmax));
}
/* raytrace.glsl:62-65 */
static inline v3 orc_ldr_to_hdr(v3 c) {
    float l = oclamp(orc_pow((c.x + c.y + c.z) / 3.0f, 0.1f), 0.0f, 0.99f);
    float k = orc_rh(l / (1.0f - l));
    return orc_rh3(V3(orc_rh(sqrtf(c.x)) * 2.0f * k, orc_rh(sqrtf(c.y)) * 2.0f * k, orc_rh(sqrtf(c.z)) * 2.0f * k));
}

/* ---- participating medium helpers (merian-shaders transmittance.glsl / phase_draine.glsl /
 * sampling.glsl, DEFINED; used by volume.comp:34-238) ------------------------------------------- */
static inline float orc_transmittance_xi_max(float tmax, float mu_t) { return 1.0f - orc_exp(-mu_t * tmax); }
static inline float orc_transmittance_sample2(float mu_t, float xi, float xi_max) { return -orc_log(omax(1.0f - xi * xi_max, 1e-37f)) / mu_t; }
static inline float orc_transmittance_pdf2(float t, float mu_t, float xi_max) { return mu_t * orc_exp(-mu_t * t) / xi_max; }
static inline float orc_sample_normal_box_muller(float mu, float sigma, float xi0, float xi1) {
    float r = sqrtf(-2.0f * orc_log(omax(xi0, 1e-37f)));
    float c, sn;
    orc_sincos2pi(xi1, &c, &sn);
    return mu + sigma * (r * c);
}
static inline float orc_sample_normal_pdf(float mu, float sigma, float x) {
    float d = (x - mu) / sigma;
    return orc_exp(-0.5f * (d * d)) / (sigma * 2.50662827463100024f);
}
/* Draine phase function (Jendersie & d'Eon 2023): cos_t = dot(travel direction in, direction out) */
static inline float orc_draine_eval(float cos_t, float g, float a) {
    float g2 = g * g;
    float s = 1.0f + g2 - 2.0f * g * cos_t;
    float s32 = s * sqrtf(s);
    return ORC_INV_4PI * ((1.0f - g2) / s32) * ((1.0f + a * (cos_t * cos_t)) / (1.0f + a * (1.0f + 2.0f * g2) / 3.0f));
}
/* antiderivative in mu of (1 + a mu^2) s^{-3/2}, s = 1 + g^2 - 2 g mu */
static inline float orc_draine_F(float mu, float g, float a) {
    float A = 1.0f + g * g;
    float s = A - 2.0f * g * mu;
    float rs = sqrtf(s);
    float k = a / (4.0f * (g * g));
    return (1.0f / g) * (1.0f / rs + k * (A * A / rs + 2.0f * A * rs - s * rs / 3.0f));
}
/* exact inverse-CDF sampling of cos(theta): Henyey-Greenstein start + 8 clamped Newton steps on the closed-form CDF */
static inline float orc_draine_sample_cos(float xi, float g, float a) {
    if (!(fabsf(g) > 1e-3f)) return 1.0f - 2.0f * xi;
    float g2 = g * g;
    float q = (1.0f - g2) / (1.0f - g + 2.0f * g * xi);
    float mu = oclamp((1.0f + g2 - q * q) / (2.0f * g), -1.0f, 1.0f);
    float F0 = orc_draine_F(-1.0f, g, a), F1 = orc_draine_F(1.0f, g, a);
    float target = F0 + xi * (F1 - F0);
    for (int i = 0; i < 8; i++) {
        float s = 1.0f + g2 - 2.0f * g * mu;
        float dF = (1.0f + a * (mu * mu)) / (s * sqrtf(s));
        mu = oclamp(mu - (orc_draine_F(mu, g, a) - target) / dF, -1.0f, 1.0f);
    }
    return mu;
}
static inline v3 orc_draine_sample(float xi0, float xi1, v3 wi, float g, float a) {
    float mu = orc_draine_sample_cos(xi0, g, a);
    float sr = sqrtf(omax(1.0f - mu * mu, 0.0f));
    float c, sn;
    orc_sincos2pi(xi1, &c, &sn);
    v3 t, b;
    orc_make_frame(wi, &t, &b);
    return vadd(vadd(vscale(t, sr * c), vscale(b, sr * sn)), vscale(wi, mu));
}
/* cosine-weighted hemisphere direction in the frame of n (make_frame(n) * sample_cos(xi)) */
static inline v3 orc_sample_cos_frame(v3 n, float xi0, float xi1) {
    float r = sqrtf(xi0), z = sqrtf(omax(1.0f - xi0, 0.0f));
    float c, sn;
    orc_sincos2pi(xi1, &c, &sn);
    v3 t, b;
    orc_make_frame(n, &t, &b);
    return vadd(vadd(vscale(t, r * c), vscale(b, r * sn)), vscale(n, z));
}

/* ---- pinhole camera (merian-shaders/camera.glsl, DEFINED): pixel centres, +y down ------------ */
static inline v3 orc_camera_ray_dir(float px, float py, float W, float H, v3 up, v3 fwd, float tan_half) {
    v3 right = vcross(fwd, up);
    float nx = ((px + 0.5f) / W) * 2.0f - 1.0f;
    float ny = ((py + 0.5f) / H) * 2.0f - 1.0f;
    float sx = nx * tan_half;
    float sy = -ny * tan_half * (H / W);
    return vnormalize(vadd(fwd, vadd(vscale(right, sx), vscale(up, sy))));
}
static inline void orc_camera_pixel(v3 dir, float W, float H, v3 up, v3 fwd, float tan_half, float* px, float* py) {
    v3 right = vcross(fwd, up);
    float z = vdot(dir, fwd);
    float x = vdot(dir, right) / z;
    float y = vdot(dir, up) / z;
    float nx = x / tan_half;
    float ny = -y / (tan_half * (H / W));
    *px = (nx * 0.5f + 0.5f) * W - 0.5f;
    *py = (ny * 0.5f + 0.5f) * H - 0.5f;
}


/* ---- debug views (mcpg.comp:212-277): definitions for two more absent symbols -------------------------------- */
/* acos on [-1, 1]: Abramowitz & Stegun 4.4.45 (|error| <= 6.7e-5), mirrored for negative arguments */
static inline float orc_acos(float x) {
    float a = fabsf(x); if (!(a < 1.0f)) a = 1.0f;
    float r = sqrtf(1.0f - a) * (1.5707288f + a * (-0.2121144f + a * (0.0742610f + a * -0.0187293f)));
    return x < 0.0f ? 3.14159274101257324f - r : r;
}
/* OKLCH -> linear sRGB (Ottosson 2020); hue in radians */
static inline v3 orc_oklch_to_rgb(v3 lch) {
    float cs, sn; orc_sincos2pi(lch.z * 0.15915493667125701904296875f, &cs, &sn);
    float a = lch.y * cs, b = lch.y * sn;
    float l_ = lch.x + 0.3963377774f * a + 0.2158037573f * b;
    float m_ = lch.x - 0.1055613458f * a - 0.0638541728f * b;
    float s_ = lch.x - 0.0894841775f * a - 1.2914855480f * b;
    float l = l_ * l_ * l_, m = m_ * m_ * m_, s3 = s_ * s_ * s_;
    return V3(4.0767416621f * l - 3.3077115913f * m + 0.2309699292f * s3,
              -1.2684380046f * l + 2.6097574011f * m - 0.3413193965f * s3,
              -0.0041960863f * l - 0.7034186147f * m + 1.7076147010f * s3);
}

#endif
