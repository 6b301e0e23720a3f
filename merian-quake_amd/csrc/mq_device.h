// mq_device.h -- device-side math and shading primitives of the HIP path tracer.
//
// DEFINITIONS for the helpers the reference pulls from the absent merian-shaders headers
// (listed in DESIGN.md).  Everything here uses IEEE +,-,*,/,sqrt in a fixed
// order and is compiled with -ffp-contract=off, so results are reproducible bit for bit by any
// implementation that performs the same operations (the test oracle does).  Hardware
// transcendentals (v_exp/v_log/v_sin/v_rcp/v_rsq) are deliberately not used in shading code;
// the only fused multiply-adds are the explicit ones in the conservative BVH box tests.
#pragma once
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>

#include "mq_types.h"

#define MQ_DEV __device__ __forceinline__
#define MQ_HD __host__ __device__ __forceinline__

// bit casts usable on host and device
MQ_HD float mq_u2f(uint32_t u) { return __builtin_bit_cast(float, u); }
MQ_HD uint32_t mq_f2u(float f) { return __builtin_bit_cast(uint32_t, f); }

struct f3 { float x, y, z; };
MQ_DEV f3 F3(float x, float y, float z) { f3 r; r.x = x; r.y = y; r.z = z; return r; }
MQ_DEV f3 operator+(f3 a, f3 b) { return F3(a.x + b.x, a.y + b.y, a.z + b.z); }
MQ_DEV f3 operator-(f3 a, f3 b) { return F3(a.x - b.x, a.y - b.y, a.z - b.z); }
MQ_DEV f3 operator*(f3 a, f3 b) { return F3(a.x * b.x, a.y * b.y, a.z * b.z); }
MQ_DEV f3 operator*(f3 a, float s) { return F3(a.x * s, a.y * s, a.z * s); }
MQ_DEV f3 operator-(f3 a) { return F3(-a.x, -a.y, -a.z); }
MQ_DEV float dot(f3 a, f3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
MQ_DEV f3 cross(f3 a, f3 b) { return F3(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x); }
MQ_DEV float length(f3 a) { return sqrtf(dot(a, a)); }
MQ_DEV f3 normalize(f3 a) { float inv = 1.0f / sqrtf(dot(a, a)); return a * inv; }
MQ_DEV float mmax(float a, float b) { return fmaxf(a, b); }
MQ_DEV float mmin(float a, float b) { return fminf(a, b); }
MQ_DEV float mclamp(float x, float lo, float hi) { return fminf(fmaxf(x, lo), hi); }
MQ_DEV float mmix(float a, float b, float t) { return a * (1.0f - t) + b * t; }
MQ_DEV bool mfinite(float x) { return (__float_as_uint(x) & 0x7f800000u) != 0x7f800000u; }

// ---- transcendental replacements (polynomials, no hardware approximations) ------------------
MQ_HD float mq_exp2(float x) {
    if (!(x >= -126.0f)) return (x != x) ? x : 0.0f;
    if (x >= 128.0f) return mq_u2f(0x7f800000u);
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
    if (e > 127) return (p * 2.0f) * mq_u2f((uint32_t)(e - 1 + 127) << 23);
    return p * mq_u2f((uint32_t)(e + 127) << 23);
}
MQ_HD float mq_log2(float x) {
    if (x != x) return x;
    if (!(x > 0.0f)) return mq_u2f(0xff800000u);
    if (x == mq_u2f(0x7f800000u)) return x;
    float bias = 0.0f;
    if (x < 1.17549435e-38f) { x = x * 16777216.0f; bias = -24.0f; }
    uint32_t b = mq_f2u(x);
    int e = (int)(b >> 23) - 127;
    float m = mq_u2f((b & 0x007fffffu) | 0x3f800000u);
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
MQ_HD float mq_exp(float x) { return mq_exp2(x * 1.44269502162933349609375f); }
MQ_HD float mq_log(float x) { return mq_log2(x) * 0.693147182464599609375f; }
MQ_HD float mq_pow(float x, float y) {
    if (x == 0.0f) return (y == 0.0f) ? 1.0f : 0.0f;
    return mq_exp2(y * mq_log2(x));
}
MQ_DEV void mq_sincos2pi(float u, float& c_out, float& s_out) {
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
    if (qi == 0) { c_out = c; s_out = s; }
    else if (qi == 1) { c_out = -s; s_out = c; }
    else if (qi == 2) { c_out = -c; s_out = -s; }
    else { c_out = s; s_out = -c; }
}
MQ_DEV float mq_sin(float x) {
    float c, s;
    mq_sincos2pi(x * 0.15915493667125701904296875f, c, s);
    return s;
}

// ---- half precision: round-to-nearest-even hardware conversions -----------------------------
// float -> half, round to nearest even, of a value that HAS BEEN ROUNDED TO FLOAT FIRST.  Without the
// (empty) asm the instruction selector folds the conversion into the producing multiply / add
// (v_fma_mixlo_f16: one rounding straight to half), which differs from round-to-float-then-to-half --
// what the oracle and any IEEE two-step evaluation give -- whenever the float result sits next to a
// half-precision tie.  One such fold per kernel made a pixel differ in one channel by one half ulp.
MQ_DEV float rounded_f32(float f) { asm("" : "+v"(f)); return f; }
MQ_DEV uint16_t f2h(float f) { return __half_as_ushort(__float2half_rn(rounded_f32(f))); }
MQ_DEV float h2f(uint16_t h) { return __half2float(__ushort_as_half(h)); }
MQ_DEV float rh(float f) { return __half2float(__float2half_rn(rounded_f32(f))); }
MQ_DEV f3 rh3(f3 a) { return F3(rh(a.x), rh(a.y), rh(a.z)); }
MQ_DEV bool h_bad(uint16_t h) { return (h & 0x7c00u) == 0x7c00u; }

// ---- RNG: PCG-4D seed hash + xorshift32 (24-bit uniforms) -----------------------------------
MQ_DEV uint32_t pcg4d16(uint32_t x, uint32_t y, uint32_t z, uint32_t w) {
    x = x * 1664525u + 1013904223u; y = y * 1664525u + 1013904223u;
    z = z * 1664525u + 1013904223u; w = w * 1664525u + 1013904223u;
    x += y * w; y += z * x; z += x * y; w += y * z;
    x ^= x >> 16; y ^= y >> 16; z ^= z >> 16; w ^= w >> 16;
    x += y * w; y += z * x; z += x * y; w += y * z;
    return x ? x : 0x9e3779b9u;
}
MQ_DEV float xorshift(uint32_t& s) {
    s ^= s << 13; s ^= s >> 17; s ^= s << 5;
    return (float)(s >> 8) * 5.9604644775390625e-8f;
}

// ---- octahedral 2x16-bit unit vector codec --------------------------------------------------
MQ_DEV float sgn1(float v) { return v >= 0.0f ? 1.0f : -1.0f; }
MQ_DEV uint32_t encode_normal(f3 n) {
    float inv = 1.0f / (fabsf(n.x) + fabsf(n.y) + fabsf(n.z));
    float px = n.x * inv, py = n.y * inv;
    if (n.z < 0.0f) {
        float tx = (1.0f - fabsf(py)) * sgn1(px);
        float ty = (1.0f - fabsf(px)) * sgn1(py);
        px = tx; py = ty;
    }
    int qx = (int)floorf(mclamp(px, -1.0f, 1.0f) * 32767.0f + 0.5f);
    int qy = (int)floorf(mclamp(py, -1.0f, 1.0f) * 32767.0f + 0.5f);
    return ((uint32_t)qx & 0xffffu) | (((uint32_t)qy & 0xffffu) << 16);
}
MQ_DEV f3 decode_normal(uint32_t e) {
    float px = (float)(int16_t)(e & 0xffffu) * (1.0f / 32767.0f);
    float py = (float)(int16_t)(e >> 16) * (1.0f / 32767.0f);
    float pz = 1.0f - fabsf(px) - fabsf(py);
    if (pz < 0.0f) {
        float tx = (1.0f - fabsf(py)) * sgn1(px);
        float ty = (1.0f - fabsf(px)) * sgn1(py);
        px = tx; py = ty;
    }
    return normalize(F3(px, py, pz));
}
MQ_DEV void make_frame(f3 n, f3& t, f3& b) {
    float sign = n.z >= 0.0f ? 1.0f : -1.0f;
    float a = -1.0f / (sign + n.z);
    float bb = n.x * n.y * a;
    t = F3(1.0f + sign * n.x * n.x * a, sign * bb, -sign * n.x);
    b = F3(bb, sign + n.y * n.y * a, -n.y);
}
MQ_DEV float luminance(f3 c) { return c.x * 0.299f + c.y * 0.587f + c.z * 0.114f; }

// ---- von Mises-Fisher -----------------------------------------------------------------------
#define MQ_INV_4PI 0.079577468335628509521484375f
#define MQ_INV_PI 0.3183098733425140380859375f
#define MQ_INV_2PI 0.15915493667125701904296875f
MQ_DEV float vmf_pdf(f3 w, f3 mu, float kappa) {
    if (!(kappa > 1e-4f)) return MQ_INV_4PI;
    float e2k = mq_exp(-2.0f * kappa);
    return kappa * MQ_INV_2PI / (1.0f - e2k) * mq_exp(kappa * (dot(mu, w) - 1.0f));
}
// the direction-independent factor of vmf_pdf, same operations: kappa * INV_2PI / (1 - e^{-2 kappa})
MQ_DEV float vmf_norm(float kappa) {
    if (!(kappa > 1e-4f)) return MQ_INV_4PI;
    float e2k = mq_exp(-2.0f * kappa);
    return kappa * MQ_INV_2PI / (1.0f - e2k);
}
MQ_DEV float vmf_pdf_normed(f3 w, f3 mu, float kappa, float norm) {
    if (!(kappa > 1e-4f)) return MQ_INV_4PI;
    return norm * mq_exp(kappa * (dot(mu, w) - 1.0f));
}
MQ_DEV f3 vmf_sample(f3 mu, float kappa, float xi0, float xi1) {
    float wz;
    if (!(kappa > 1e-4f)) wz = 1.0f - 2.0f * xi0;
    else {
        float e2k = mq_exp(-2.0f * kappa);
        wz = 1.0f + mq_log(mmax(xi0 + (1.0f - xi0) * e2k, 1e-37f)) / kappa;
    }
    wz = mclamp(wz, -1.0f, 1.0f);
    float sr = sqrtf(mmax(1.0f - wz * wz, 0.0f));
    float c, s;
    mq_sincos2pi(xi1, c, s);
    f3 t, b;
    make_frame(mu, t, b);
    return (t * (sr * c) + b * (sr * s)) + mu * wz;
}

// ---- BSDF: 50/50 Lambert + GGX; wi points INTO the surface ----------------------------------
MQ_DEV float roughness_to_alpha(float r) { return r * r; }
MQ_DEV float ggx_D(float ndoth, float alpha) {
    float a2 = alpha * alpha;
    float d = ndoth * ndoth * (a2 - 1.0f) + 1.0f;
    return a2 * MQ_INV_PI / (d * d);
}
MQ_DEV float ggx_G1(float ndotx, float alpha) {
    float a2 = alpha * alpha;
    return 2.0f * ndotx / (ndotx + sqrtf(a2 + (1.0f - a2) * ndotx * ndotx));
}
MQ_DEV f3 bsdf_sample(f3 wi, f3 n, float alpha, float xi0, float xi1, float xi2) {
    f3 t, b;
    make_frame(n, t, b);
    float c, s;
    mq_sincos2pi(xi1, c, s);
    if (xi2 < 0.5f) {
        float r = sqrtf(xi0);
        float z = sqrtf(mmax(1.0f - xi0, 0.0f));
        return (t * (r * c) + b * (r * s)) + n * z;
    }
    float a2 = alpha * alpha;
    float ct2 = (1.0f - xi0) / (1.0f + (a2 - 1.0f) * xi0);
    float ct = sqrtf(ct2);
    float st = sqrtf(mmax(1.0f - ct2, 0.0f));
    f3 h = (t * (st * c) + b * (st * s)) + n * ct;
    float d = dot(wi, h);
    return wi - h * (2.0f * d);
}
MQ_DEV float bsdf_pdf(f3 wi, f3 wo, f3 n, float alpha) {
    float ndoto = dot(n, wo);
    if (!(ndoto > 0.0f)) return 0.0f;
    f3 v = -wi;
    f3 hs = v + wo;
    float hl = length(hs);
    float pd = 0.5f * ndoto * MQ_INV_PI;
    if (!(hl > 1e-12f)) return pd;
    f3 h = hs * (1.0f / hl);
    float ndoth = dot(n, h), vdoth = dot(v, h);
    if (!(ndoth > 0.0f) || !(vdoth > 0.0f)) return pd;
    return pd + 0.5f * ggx_D(ndoth, alpha) * ndoth / (4.0f * vdoth);
}
MQ_DEV float bsdf_times_wodotn(f3 wi, f3 wo, f3 n, float alpha, float F0) {
    float ndoto = dot(n, wo);
    f3 v = -wi;
    float ndotv = dot(n, v);
    if (!(ndoto > 0.0f) || !(ndotv > 0.0f)) return 0.0f;
    f3 hs = v + wo;
    float hl = length(hs);
    if (!(hl > 1e-12f)) return ndoto * MQ_INV_PI;
    f3 h = hs * (1.0f / hl);
    float ndoth = mmax(dot(n, h), 0.0f), vdoth = mmax(dot(v, h), 0.0f);
    float m = 1.0f - vdoth;
    float m2 = m * m;
    float F = F0 + (1.0f - F0) * (m2 * m2 * m);
    float mv = 1.0f - ndotv;
    float mv2 = mv * mv;
    float Fv = F0 + (1.0f - F0) * (mv2 * mv2 * mv);
    float spec = F * ggx_D(ndoth, alpha) * ggx_G1(ndotv, alpha) * ggx_G1(ndoto, alpha) / (4.0f * ndotv * ndoto);
    return ((1.0f - Fv) * MQ_INV_PI + spec) * ndoto;
}

// ---- hash grid ------------------------------------------------------------------------------
MQ_DEV uint32_t hash_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
    return x;
}
MQ_DEV uint32_t hash2_u32(uint32_t x) {
    x ^= x >> 16; x *= 0x85ebca6bu; x ^= x >> 13; x *= 0xc2b2ae35u; x ^= x >> 16;
    return x;
}
struct i3 { int x, y, z; };
// `inv_width` = 1.0f / width (IEEE division, precomputed on the host or in-kernel)
MQ_DEV i3 grid_idx_interpolate(f3 pos, float inv_width, float xi) {
    i3 r;
    r.x = (int)floorf(pos.x * inv_width + xi);
    r.y = (int)floorf(pos.y * inv_width + xi);
    r.z = (int)floorf(pos.z * inv_width + xi);
    return r;
}
// floor(-log2(1 - xi)) for xi = k * 2^-24 (mc.glsl:70), exact in integers
MQ_DEV uint32_t level_jitter(float xi) {
    uint32_t m = 16777216u - (uint32_t)(xi * 16777216.0f);
    uint32_t fl = 31u - (uint32_t)__clz((int)m);
    return (m & (m - 1u)) ? 23u - fl : 24u - fl;
}
// multiply-high range reduction of a 32-bit hash to [0, size)
MQ_DEV uint32_t reduce_range(uint32_t h, uint32_t size) { return __umulhi(h, size); }
// grid cell width of a level (mc.glsl:73,75; light_cache.glsl:21,23) -- also evaluated on the host to fill the per-level tables
MQ_HD float grid_width(int type, float steps, float minw, float power, uint32_t level) {
    if (type == 0) return minw * mq_pow(power, (float)level / steps);
    return mq_pow((float)level / steps, power) + minw;
}
MQ_DEV uint32_t normal_face(f3 n) {
    float ax = fabsf(n.x), ay = fabsf(n.y), az = fabsf(n.z);
    if (ax >= ay && ax >= az) return n.x < 0.0f ? 1u : 0u;
    if (ay >= az) return n.y < 0.0f ? 3u : 2u;
    return n.z < 0.0f ? 5u : 4u;
}
MQ_DEV uint32_t hash3(i3 c, uint32_t salt) {
    return hash_u32((uint32_t)c.x + hash_u32((uint32_t)c.y + hash_u32((uint32_t)c.z + salt)));
}
MQ_DEV uint32_t hash_grid(i3 c, uint32_t size) { return reduce_range(hash3(c, 0x51ed270bu), size); }
MQ_DEV uint32_t hash_grid_normal_level(i3 c, f3 n, uint32_t level, uint32_t size) {
    return reduce_range(hash3(c, hash_u32(level * 8u + normal_face(n) + 0x2545f491u)), size);
}
MQ_DEV uint32_t hash2_3(i3 c, uint32_t salt) {
    return hash2_u32((uint32_t)c.x * 0x9e3779b1u + hash2_u32((uint32_t)c.y * 0x85ebca77u + hash2_u32((uint32_t)c.z * 0xc2b2ae3du + salt)));
}
MQ_DEV uint32_t hash2_grid(i3 c) { return hash2_3(c, 0x27d4eb2fu); }
MQ_DEV uint32_t hash2_grid_level(i3 c, uint32_t level) { return hash2_3(c, 0x165667b1u + level); }

// ---- misc -----------------------------------------------------------------------------------
MQ_DEV float transmittance(float t, float mu_t, float tmax) {
    if (mu_t == 0.0f) return 1.0f;
    return mq_exp(-mu_t * mmin(t, tmax));
}
MQ_DEV f3 ldr_to_hdr(f3 c) { // raytrace.glsl:62-65
    float l = mclamp(mq_pow((c.x + c.y + c.z) / 3.0f, 0.1f), 0.0f, 0.99f);
    float k = rh(l / (1.0f - l));
    return rh3(F3(rh(sqrtf(c.x)) * 2.0f * k, rh(sqrtf(c.y)) * 2.0f * k, rh(sqrtf(c.z)) * 2.0f * k));
}
// ---- participating medium helpers (volume.comp:34-238; DEFINED) ----------------
MQ_DEV float transmittance_xi_max(float tmax, float mu_t) { return 1.0f - mq_exp(-mu_t * tmax); }
MQ_DEV float transmittance_sample2(float mu_t, float xi, float xi_max) { return -mq_log(mmax(1.0f - xi * xi_max, 1e-37f)) / mu_t; }
MQ_DEV float transmittance_pdf2(float t, float mu_t, float xi_max) { return mu_t * mq_exp(-mu_t * t) / xi_max; }
MQ_DEV float sample_normal_box_muller(float mu, float sigma, float xi0, float xi1) {
    float r = sqrtf(-2.0f * mq_log(mmax(xi0, 1e-37f)));
    float c, sn;
    mq_sincos2pi(xi1, c, sn);
    return mu + sigma * (r * c);
}
MQ_DEV float sample_normal_pdf(float mu, float sigma, float x) {
    float d = (x - mu) / sigma;
    return mq_exp(-0.5f * (d * d)) / (sigma * 2.50662827463100024f);
}
// Draine phase function (Jendersie & d'Eon 2023); cos_t = dot(travel direction in, direction out)
MQ_DEV float draine_eval(float cos_t, float g, float a) {
    float g2 = g * g;
    float s = 1.0f + g2 - 2.0f * g * cos_t;
    float s32 = s * sqrtf(s);
    return MQ_INV_4PI * ((1.0f - g2) / s32) * ((1.0f + a * (cos_t * cos_t)) / (1.0f + a * (1.0f + 2.0f * g2) / 3.0f));
}
MQ_DEV float draine_F(float mu, float g, float a) { // antiderivative in mu of (1 + a mu^2) s^{-3/2}
    float A = 1.0f + g * g;
    float s = A - 2.0f * g * mu;
    float rs = sqrtf(s);
    float k = a / (4.0f * (g * g));
    return (1.0f / g) * (1.0f / rs + k * (A * A / rs + 2.0f * A * rs - s * rs / 3.0f));
}
MQ_DEV float draine_sample_cos(float xi, float g, float a) { // HG start + 8 clamped Newton steps on the closed-form CDF
    if (!(fabsf(g) > 1e-3f)) return 1.0f - 2.0f * xi;
    float g2 = g * g;
    float q = (1.0f - g2) / (1.0f - g + 2.0f * g * xi);
    float mu = mclamp((1.0f + g2 - q * q) / (2.0f * g), -1.0f, 1.0f);
    float F0 = draine_F(-1.0f, g, a), F1 = draine_F(1.0f, g, a);
    float target = F0 + xi * (F1 - F0);
#pragma unroll 1
    for (int i = 0; i < 8; i++) {
        float s = 1.0f + g2 - 2.0f * g * mu;
        float dF = (1.0f + a * (mu * mu)) / (s * sqrtf(s));
        mu = mclamp(mu - (draine_F(mu, g, a) - target) / dF, -1.0f, 1.0f);
    }
    return mu;
}
MQ_DEV f3 draine_sample(float xi0, float xi1, f3 wi, float g, float a) {
    float mu = draine_sample_cos(xi0, g, a);
    float sr = sqrtf(mmax(1.0f - mu * mu, 0.0f));
    float c, sn;
    mq_sincos2pi(xi1, c, sn);
    f3 t, b;
    make_frame(wi, t, b);
    return (t * (sr * c) + b * (sr * sn)) + wi * mu;
}
MQ_DEV f3 sample_cos_frame(f3 n, float xi0, float xi1) {
    float r = sqrtf(xi0), z = sqrtf(mmax(1.0f - xi0, 0.0f));
    float c, sn;
    mq_sincos2pi(xi1, c, sn);
    f3 t, b;
    make_frame(n, t, b);
    return (t * (r * c) + b * (r * sn)) + n * z;
}
MQ_DEV f3 camera_ray_dir(float px, float py, float W, float H, f3 up, f3 fwd, float tan_half) {
    f3 right = cross(fwd, up);
    float nx = ((px + 0.5f) / W) * 2.0f - 1.0f;
    float ny = ((py + 0.5f) / H) * 2.0f - 1.0f;
    float sx = nx * tan_half;
    float sy = -ny * tan_half * (H / W);
    return normalize(fwd + (right * sx + up * sy));
}
MQ_DEV void camera_pixel(f3 dir, float W, float H, f3 up, f3 fwd, float tan_half, float& px, float& py) {
    f3 right = cross(fwd, up);
    float z = dot(dir, fwd);
    float x = dot(dir, right) / z;
    float y = dot(dir, up) / z;
    float nx = x / tan_half;
    float ny = -y / (tan_half * (H / W));
    px = (nx * 0.5f + 0.5f) * W - 0.5f;
    py = (ny * 0.5f + 0.5f) * H - 0.5f;
}

// ---- debug views (mcpg.comp:212-277): acos (Abramowitz & Stegun 4.4.45, mirrored) and OKLCH -> linear sRGB, as in orc_math.h
MQ_DEV float mq_acos(float x) {
    float a = fabsf(x); if (!(a < 1.0f)) a = 1.0f;
    float r = sqrtf(1.0f - a) * (1.5707288f + a * (-0.2121144f + a * (0.0742610f + a * -0.0187293f)));
    return x < 0.0f ? 3.14159274101257324f - r : r;
}
MQ_DEV f3 oklch_to_rgb(f3 lch) {
    float cs, sn; mq_sincos2pi(lch.z * 0.15915493667125701904296875f, cs, sn);
    float a = lch.y * cs, b = lch.y * sn;
    float l_ = lch.x + 0.3963377774f * a + 0.2158037573f * b;
    float m_ = lch.x - 0.1055613458f * a - 0.0638541728f * b;
    float s_ = lch.x - 0.0894841775f * a - 1.2914855480f * b;
    float l = l_ * l_ * l_, m = m_ * m_ * m_, s3 = s_ * s_ * s_;
    return F3(4.0767416621f * l - 3.3077115913f * m + 0.2309699292f * s3,
              -1.2684380046f * l + 2.6097574011f * m - 0.3413193965f * s3,
              -0.0041960863f * l - 0.7034186147f * m + 1.7076147010f * s3);
}
