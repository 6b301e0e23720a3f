// mq_kernels.hip -- hand-written gfx950 kernels of the MCPG path tracer.
//
//   mq_primary_trace_kernel closest hit of the camera rays, one 8x8 tile per wave
//   mq_primary_kernel    g-buffer node on those hits (res/shader/gbuffer/gbuffer.comp:75-131) and the
//                        first direction choice of the surface estimator
//   mq_trace_queue_kernel software CWBVH closest hit for every queued ray
//   mq_bounce_kernel     shading + guiding + learning of res/shader/render_mcpg/mcpg.comp:39-210
//   mq_apply_kernel      Markov-chain update application (compute_updates.comp:56-124) over a
//                        compact update queue instead of the reference's 17 GB slot array.
//   mq_clear_kernel      clear.comp:15-23 + the CLEAR variant of gbuffer.comp:83-90.
//   mq_untile_kernel     multi-GPU: gathered tile-major radiance -> full image.
//   mq_trace_kernel      closest-hit queries (raytrace.glsl:82-119 semantics).
//   mq_math_kernel       device-side known-answer evaluation of the shading primitives.
//
// Scheduling: a wavefront pipeline with stream compaction between rounds (see the block comment
// above mq_primary_kernel).  Traversal stacks live in LDS ([entry][lane] layout, conflict free)
// with a global spill area behind them.  No MFMA: the path is divergent traversal, not a contraction.
#include "mq_device.h"

#define MQ_BLOCK 256
#define MQ_WAVES (MQ_BLOCK / 64)
#ifndef MQ_STACK_LDS
#define MQ_STACK_LDS 12
#endif
#define MQ_SPILL_ENTRIES (64 - MQ_STACK_LDS)
// The shading kernels take their LDS dynamically: F.lds_rows2 rows of 64 x 8 bytes per wave, sized by the
// host from the run-time sample count K (3 rows per Markov-chain sample; the primary kernel's
// traversal stack shares the region), so that K = 5 costs 30 KB per block instead of the 48 KB of K = 8.
#ifndef MQ_OCC_SHADE
#define MQ_OCC_SHADE 2
#endif
#ifndef MQ_OCC_TRACE
#define MQ_OCC_TRACE 5
#endif

struct RayHit { uint32_t tri; float t, u, v; };

struct Hit { // res/shader/hit.glsl.h:6-17
    f3 pos, prev_pos, wi, normal;
    uint32_t enc_geonormal;
    f3 albedo;
    float roughness;
};

// image tile of local tile `ltile` of this launch (interleaved partition, or a band of tile rows: MqFrame::tile_mul)
#define MQ_GTILE(F, ltile) ((ltile) * (F).tile_mul + (F).tile_add)
// position of entry k of shard `shard` in a sharded queue (see "sharded queues" below)
MQ_DEV uint32_t shard_pos(uint32_t shard, uint32_t k) { return (((k >> 6) * MQ_SHARDS + shard) << 6) | (k & 63u); }

struct Ctr {
    uint32_t rays, nodes, tris, segments, guided, lc, upd_ok, upd_drop, mc_reads, pixels, lc_ok, lc_cancel;
#ifdef MQ_PROF
    uint32_t pt, prof[MQ_PROF_SECTIONS];
#endif
};
// Lap profiling (-DMQ_PROF builds only): PLAP(ctr, i) charges the shader clocks since the previous
// lap of this wave to section i.  Sections: see tools/prof_sections.py.
#ifdef MQ_PROF
MQ_DEV void prof_lap(Ctr& c, int i) {
    const uint32_t t = (uint32_t)__builtin_readcyclecounter();
    const int32_t d = (int32_t)(t - c.pt); // two back-to-back reads of the scalar clock can complete out of order: never charge a "negative" lap as 2^32
    if (d > 0) c.prof[i] += (uint32_t)d;
    c.pt = t;
}
MQ_DEV void prof_start(Ctr& c) { c.pt = (uint32_t)__builtin_readcyclecounter(); }
MQ_DEV void prof_flush(MqCountersDev* g, const Ctr& c) {
    if ((threadIdx.x & 63) == 0) for (int i = 0; i < MQ_PROF_SECTIONS; i++) if (c.prof[i]) atomicAdd(&g->prof[i], (unsigned long long)c.prof[i]);
}
#define PLAP(c, i) prof_lap(c, i)
#define PSTART(c) prof_start(c)
#define PFLUSH(g, c) prof_flush(g, c)
#else
#define PLAP(c, i)
#define PSTART(c)
#define PFLUSH(g, c)
#endif

// ------------------------------------------------------------------------------------------------
// textures: texel pool decoded to linear RGBA32F at commit (sRGB LUT or x/255, the values a per-fetch decode
// would give), REPEAT wrap, nearest or bilinear: a texel is ONE 16-byte gather instead of a packed texel plus
// three LUT gathers
// ------------------------------------------------------------------------------------------------
struct f4 { float r, g, b, a; };

// REPEAT addressing without integer division: u = s - floor(s) in [0,1], then scale by the size
MQ_DEV f4 texel(const MqSceneDev& sc, const MqTexDesc& t, int x, int y) { // x, y already in range
    const float4 p = sc.texels[t.offset + (uint32_t)y * t.w + (uint32_t)x];
    f4 r; r.r = p.x; r.g = p.y; r.b = p.z; r.a = p.w;
    return r;
}
MQ_DEV int tex_nearest_coord(float s, float fw, int w) {
    float u = s - floorf(s);
    int i = (int)floorf(u * fw);
    return i > w - 1 ? w - 1 : i;
}
MQ_DEV void tex_linear_coord(float s, float fw, int w, int& i0, int& i1, float& f) {
    float u = s - floorf(s);
    float x = u * fw - 0.5f;
    float x0 = floorf(x);
    f = x - x0;
    int a = (int)x0, b = a + 1;
    if (a < 0) a += w;
    if (b >= w) b -= w;
    i0 = a; i1 = b;
}
MQ_DEV f4 tex_sample_desc(const MqSceneDev& sc, const MqTexDesc& tx, float s, float t) {
    if (tx.offset == MQ_NIL) { f4 g; g.r = g.g = g.b = 0.5f; g.a = 1.0f; return g; }
    float fw = (float)tx.w, fh = (float)tx.h;
    if (!(tx.flags & MQ_TEX_LINEAR)) return texel(sc, tx, tex_nearest_coord(s, fw, (int)tx.w), tex_nearest_coord(t, fh, (int)tx.h));
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, fw, (int)tx.w, x0, x1, fx);
    tex_linear_coord(t, fh, (int)tx.h, y0, y1, fy);
    f4 a = texel(sc, tx, x0, y0), b = texel(sc, tx, x1, y0), d = texel(sc, tx, x0, y1), e = texel(sc, tx, x1, y1);
    f4 r;
    r.r = mmix(mmix(a.r, b.r, fx), mmix(d.r, e.r, fx), fy);
    r.g = mmix(mmix(a.g, b.g, fx), mmix(d.g, e.g, fx), fy);
    r.b = mmix(mmix(a.b, b.b, fx), mmix(d.b, e.b, fx), fy);
    r.a = mmix(mmix(a.a, b.a, fx), mmix(d.a, e.a, fx), fy);
    return r;
}
MQ_DEV f4 tex_sample(const MqSceneDev& sc, uint32_t texnum, float s, float t) {
    if (texnum > MQ_MAX_GLTEXTURES - 1) texnum = MQ_MAX_GLTEXTURES - 1;
    return tex_sample_desc(sc, sc.tex[texnum], s, t);
}
// ---- mip chain + textureGrad (first hit only, raytrace.glsl:232-245,299-303); definitions as in the oracle:
// LOD from the Vulkan footprint formula, lambda <= 0 (or NaN) -> magnification path, else bilinear in the two
// nearest levels mixed by frac(lambda).  Level k follows level k-1 in the pool.
MQ_DEV uint32_t mip_dim(uint32_t d, uint32_t k) { uint32_t v = d >> k; return v ? v : 1u; }
MQ_DEV f4 tex_bilinear_level(const MqSceneDev& sc, const MqTexDesc& tx, uint32_t level, float s, float t) {
    uint32_t off = tx.offset;
    for (uint32_t k = 0; k < level; k++) off += mip_dim(tx.w, k) * mip_dim(tx.h, k);
    const int w = (int)mip_dim(tx.w, level), h = (int)mip_dim(tx.h, level);
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, (float)w, w, x0, x1, fx);
    tex_linear_coord(t, (float)h, h, y0, y1, fy);
    const float4 a = sc.texels[off + (uint32_t)y0 * w + (uint32_t)x0], b = sc.texels[off + (uint32_t)y0 * w + (uint32_t)x1];
    const float4 d = sc.texels[off + (uint32_t)y1 * w + (uint32_t)x0], e = sc.texels[off + (uint32_t)y1 * w + (uint32_t)x1];
    f4 r;
    r.r = mmix(mmix(a.x, b.x, fx), mmix(d.x, e.x, fx), fy);
    r.g = mmix(mmix(a.y, b.y, fx), mmix(d.y, e.y, fx), fy);
    r.b = mmix(mmix(a.z, b.z, fx), mmix(d.z, e.z, fx), fy);
    r.a = mmix(mmix(a.w, b.w, fx), mmix(d.w, e.w, fx), fy);
    return r;
}
MQ_DEV f4 tex_sample_grad_desc(const MqSceneDev& sc, const MqTexDesc& tx, float s, float t, float dsdx, float dtdx, float dsdy, float dtdy) {
    const uint32_t levels = (tx.flags >> 8) & 0xffu;
    if (tx.offset == MQ_NIL || levels <= 1u) return tex_sample_desc(sc, tx, s, t);
    const float fw = (float)tx.w, fh = (float)tx.h;
    const float ax = dsdx * fw, ay = dtdx * fh, bx = dsdy * fw, by = dtdy * fh;
    const float rx = sqrtf(ax * ax + ay * ay), ry = sqrtf(bx * bx + by * by);
    const float rho = rx > ry ? rx : ry;
    if (!(rho > 1.0f)) return tex_sample_desc(sc, tx, s, t);
    float lambda = mq_log2(rho);
    const float top = (float)(levels - 1u);
    if (!(lambda < top)) lambda = top;
    const float fl = floorf(lambda);
    const uint32_t lo = (uint32_t)fl, hi = lo + 1u < levels ? lo + 1u : lo;
    const float f = lambda - fl;
    const f4 c0 = tex_bilinear_level(sc, tx, lo, s, t);
    if (hi == lo || !(f > 0.0f)) return c0;
    const f4 c1 = tex_bilinear_level(sc, tx, hi, s, t);
    f4 r; r.r = mmix(c0.r, c1.r, f); r.g = mmix(c0.g, c1.g, f); r.b = mmix(c0.b, c1.b, f); r.a = mmix(c0.a, c1.a, f);
    return r;
}
MQ_DEV float tex_gather_alpha_r(const MqSceneDev& sc, const MqTexDesc& tx, float s, float t) {
    if (tx.offset == MQ_NIL) return 1.0f;
    int x0, x1, y0, y1; float fx, fy;
    tex_linear_coord(s, (float)tx.w, (int)tx.w, x0, x1, fx);
    tex_linear_coord(t, (float)tx.h, (int)tx.h, y0, y1, fy);
    return ((const float*)(sc.texels + (tx.offset + (uint32_t)y1 * tx.w + (uint32_t)x0)))[3]; // alpha only: 4 bytes
}


// shading record of BVH triangle `tri`: extra data + resolved texture descriptors, four 16-byte loads
MQ_DEV MqTexDesc desc_from(uint32_t a, uint32_t b, uint32_t c) { MqTexDesc d; d.offset = a; d.w = (uint16_t)(b & 0xffffu); d.h = (uint16_t)(b >> 16); d.flags = c; return d; }
MQ_DEV void load_shade(const MqSceneDev& sc, uint32_t tri, mq_ext& e, MqTexDesc& albedo, MqTexDesc& fb) {
    const uint4* p = (const uint4*)(sc.shade + tri);
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    union { mq_ext e; uint32_t w[7]; } u;
    u.w[0] = a.x; u.w[1] = a.y; u.w[2] = a.z; u.w[3] = a.w; u.w[4] = b.x; u.w[5] = b.y; u.w[6] = b.z;
    e = u.e;
    albedo = desc_from(c.x, c.y, c.z);
    fb = desc_from(d.x, d.y, d.z);
}

// any-hit confirmation, raytrace.glsl:100-118
MQ_DEV bool anyhit_confirm(const MqSceneDev& sc, uint32_t tri, float u, float v) {
    const uint4* p = (const uint4*)(sc.shade + tri);
    const uint4 a = p[0], b = p[1], c = p[2];
    union { mq_ext e; uint32_t w[7]; } x;
    x.w[0] = a.x; x.w[1] = a.y; x.w[2] = a.z; x.w[3] = a.w; x.w[4] = b.x; x.w[5] = b.y; x.w[6] = b.z;
    const mq_ext& e = x.e;
    uint32_t flags = e.texnum_fb_flags >> 12, alpha = e.texnum_alpha >> 12;
    if (flags > 0 && flags < 7) return true;
    if (alpha != 0) return rh((float)(alpha - 1) / 14.0f) >= MQ_ALPHA_THRESHOLD;
    float b0 = 1.0f - u - v;
    float s = h2f(e.st[0]) * b0 + h2f(e.st[2]) * u + h2f(e.st[4]) * v;
    float t = h2f(e.st[1]) * b0 + h2f(e.st[3]) * u + h2f(e.st[5]) * v;
    return tex_gather_alpha_r(sc, desc_from(c.x, c.y, c.z), s, t) >= MQ_ALPHA_THRESHOLD;
}

// ------------------------------------------------------------------------------------------------
// traversal of the 8-wide compressed BVH
// ------------------------------------------------------------------------------------------------

// Moeller-Trumbore, front faces only (geometric normal = cross(v2-v0, v1-v0), raytrace.glsl:221-223;
// back faces culled, raytrace.glsl:73,85).  Exact op order matters: results are compared bit for bit.
MQ_DEV bool tri_isect(f3 o, f3 d, f3 v0, f3 v1, f3 v2, float& t, float& u, float& v) {
    f3 e1 = v1 - v0, e2 = v2 - v0;
    f3 pv = cross(d, e2);
    float det = dot(e1, pv);
    if (!(det < 0.0f)) return false;
    float inv = 1.0f / det;
    f3 tv = o - v0;
    float uu = dot(tv, pv) * inv;
    if (!(uu >= -MQ_BARY_EPS) || !(uu <= 1.0f + MQ_BARY_EPS)) return false;
    f3 qv = cross(tv, e1);
    float vv = dot(d, qv) * inv;
    if (!(vv >= -MQ_BARY_EPS) || !(uu + vv <= 1.0f + MQ_BARY_EPS)) return false;
    float tt = dot(e2, qv) * inv;
    if (!(tt > 0.0f)) return false;
    t = tt; u = uu; v = vv;
    return true;
}

// One group of four children: bytes of nq/fq hold near/far quantised planes per axis.  The slab
// distances of two children are evaluated per instruction (v_pk_fma_f32, one rounding per product-sum
// exactly like fmaf), and the per-child meta bytes (child bits << 5 | bit index, CWBVH paper sect. 4)
// are decoded for all four children at once with byte-parallel integer arithmetic.  An empty child
// has meta 0 and contributes no bits whatever its box test says.
typedef float v2f __attribute__((ext_vector_type(2)));
MQ_DEV v2f ub2(uint32_t w, int k) { v2f r; r.x = (float)((w >> (8 * k)) & 0xffu); r.y = (float)((w >> (8 * k + 8)) & 0xffu); return r; }
MQ_DEV uint32_t box4(uint32_t nx, uint32_t ny, uint32_t nz, uint32_t fx, uint32_t fy, uint32_t fz, uint32_t meta,
                     float adx, float ady, float adz, float ox, float oy, float oz, float tlim, uint32_t oct4) {
    const uint32_t inner = ((meta & (meta << 1)) & 0x10101010u) >> 4;      // 1 per byte whose meta has (m & 0x18) == 0x18
    const uint32_t bit4 = (meta ^ (oct4 & ((inner << 8) - inner))) & 0x1f1f1f1fu; // bit index per child, octant-relative for inner nodes (x * 255 as shift - x: full rate)
    const uint32_t cb4 = (meta >> 5) & 0x07070707u;                        // child bits per child
    const v2f ax = {adx, adx}, ay = {ady, ady}, az = {adz, adz}, bx = {ox, ox}, by = {oy, oy}, bz = {oz, oz};
    uint32_t mask = 0;
#pragma unroll
    for (int j = 0; j < 4; j += 2) {
        v2f tnx = __builtin_elementwise_fma(ub2(nx, j), ax, bx), tfx = __builtin_elementwise_fma(ub2(fx, j), ax, bx);
        v2f tny = __builtin_elementwise_fma(ub2(ny, j), ay, by), tfy = __builtin_elementwise_fma(ub2(fy, j), ay, by);
        v2f tnz = __builtin_elementwise_fma(ub2(nz, j), az, bz), tfz = __builtin_elementwise_fma(ub2(fz, j), az, bz);
        float tn0 = fmaxf(fmaxf(tnx.x, tny.x), fmaxf(tnz.x, 0.0f)), tf0 = fminf(fminf(tfx.x, tfy.x), fminf(tfz.x, tlim));
        float tn1 = fmaxf(fmaxf(tnx.y, tny.y), fmaxf(tnz.y, 0.0f)), tf1 = fminf(fminf(tfx.y, tfy.y), fminf(tfz.y, tlim));
        if (tn0 <= tf0) mask |= ((cb4 >> (8 * j)) & 0xffu) << ((bit4 >> (8 * j)) & 0xffu);
        if (tn1 <= tf1) mask |= ((cb4 >> (8 * j + 8)) & 0xffu) << ((bit4 >> (8 * j + 8)) & 0xffu);
    }
    return mask;
}

// Per-lane traversal state.  trav_step() performs ONE node visit (pop the nearest pending child,
// intersect its 8 children, test the triangles of the leaves that were hit); traverse() loops it,
// and the persistent queue kernel interleaves steps of different rays in one wave.
struct Trav {
    f3 o, d;
    float idx, idy, idz;
    float tlim;      // box-test limit: min(closest hit, MQ_T_MAX) * (1 + 2^-20) + 1e-6, refreshed when a hit is accepted
    uint32_t oct4;   // octinv in every byte; octinv bit k set: direction component k is >= 0
    uint2 G;
    int sp;
    int sb;          // stack base: entries [sb, sp) are this lane's (rows below sb were handed to helper lanes, see work sharing)
    RayHit hit;
    uint32_t best_key;
    uint32_t tmask, tbase; // triangles of the last visited node that still have to be tested
};

MQ_DEV float trav_limit(float t_closest) { return t_closest * 1.000001f + 1e-6f; }
MQ_DEV void trav_init(Trav& t, f3 o, f3 d) { // rays end at MQ_T_MAX (raytrace.glsl:82-119)
    t.o = o; t.d = d; t.tlim = trav_limit(MQ_T_MAX);
    t.idx = 1.0f / (fabsf(d.x) > 1e-20f ? d.x : (d.x < 0.0f ? -1e-20f : 1e-20f));
    t.idy = 1.0f / (fabsf(d.y) > 1e-20f ? d.y : (d.y < 0.0f ? -1e-20f : 1e-20f));
    t.idz = 1.0f / (fabsf(d.z) > 1e-20f ? d.z : (d.z < 0.0f ? -1e-20f : 1e-20f));
    t.oct4 = ((d.x < 0.0f ? 0u : 1u) | (d.y < 0.0f ? 0u : 2u) | (d.z < 0.0f ? 0u : 4u)) * 0x01010101u;
    t.G = make_uint2(0u, 0x80000000u);
    t.sp = 0; t.sb = 0;
    t.hit.tri = MQ_NIL; t.hit.t = __uint_as_float(0x7f800000u); t.hit.u = 0.0f; t.hit.v = 0.0f;
    t.best_key = MQ_NIL;
    t.tmask = 0; t.tbase = 0;
}

// The per-frame tree (quake_node.cpp:896-983) hangs under no node of the static tree: its root waits at the bottom
// of the ray's stack, so it is visited last, when the closest static hit already bounds it.  After trav_init.
MQ_DEV void trav_defer(const MqSceneDev& sc, Trav& t, uint2* stk) {
    if (sc.dyn_root != MQ_NIL) { stk[0] = make_uint2(sc.dyn_root, 0x80000000u); t.sp = 1; }
}

// The two halves of a node visit.  trav_node_pop: the nearest pending child of the current group leaves it (the rest of
// the group goes to the stack) -- returns the child's record; trav_node_test: its 8 children against the ray.
MQ_DEV const uint4* trav_node_pop(const MqSceneDev& sc, Trav& t, uint2* stk /* &lds[0][lane] */, unsigned long long* spill) {
    uint2 G = t.G;
    uint32_t bit = 31u - (uint32_t)__clz((int)G.y);
    G.y &= ~(1u << bit);
    if (G.y > 0x00ffffffu) {
        if (t.sp < MQ_STACK_LDS) stk[t.sp * 64] = G;
        else spill[t.sp - MQ_STACK_LDS] = ((unsigned long long)G.y << 32) | G.x;
        t.sp++;
    }
    uint32_t slot = (bit - 24u) ^ (t.oct4 & 7u);
    uint32_t rel = (uint32_t)__popc(G.y & 0xffu & ((1u << slot) - 1u));
    return (const uint4*)(sc.nodes + (G.x + rel));
}
MQ_DEV void trav_node_test(Trav& t, const uint4 n0, const uint4 n1, const uint4 n2, const uint4 n3, const uint4 n4) {
    const bool sx = !(t.oct4 & 1u), sy = !(t.oct4 & 2u), sz = !(t.oct4 & 4u);
    const float tlim = t.tlim;
    // 1/d scaled by the node's power-of-two cell size: an exponent-field add (exact; |1/d| is in [1, 1e20] and the
    // biased exponents are far from both ends of the range)
    float adx = __uint_as_float(__float_as_uint(t.idx) + ((n0.w & 0xffu) << 23) - (127u << 23));
    float ady = __uint_as_float(__float_as_uint(t.idy) + (((n0.w >> 8) & 0xffu) << 23) - (127u << 23));
    float adz = __uint_as_float(__float_as_uint(t.idz) + (((n0.w >> 16) & 0xffu) << 23) - (127u << 23));
    float ox = (__uint_as_float(n0.x) - t.o.x) * t.idx;
    float oy = (__uint_as_float(n0.y) - t.o.y) * t.idy;
    float oz = (__uint_as_float(n0.z) - t.o.z) * t.idz;
    // near / far plane bytes per axis, by ray direction sign
    uint32_t nx0 = sx ? n3.z : n2.x, nx1 = sx ? n3.w : n2.y, fx0 = sx ? n2.x : n3.z, fx1 = sx ? n2.y : n3.w;
    uint32_t ny0 = sy ? n4.x : n2.z, ny1 = sy ? n4.y : n2.w, fy0 = sy ? n2.z : n4.x, fy1 = sy ? n2.w : n4.y;
    uint32_t nz0 = sz ? n4.z : n3.x, nz1 = sz ? n4.w : n3.y, fz0 = sz ? n3.x : n4.z, fz1 = sz ? n3.y : n4.w;
    const uint32_t oct4 = t.oct4;
    uint32_t hm = box4(nx0, ny0, nz0, fx0, fy0, fz0, n1.z, adx, ady, adz, ox, oy, oz, tlim, oct4) |
                  box4(nx1, ny1, nz1, fx1, fy1, fz1, n1.w, adx, ady, adz, ox, oy, oz, tlim, oct4);
    t.G = make_uint2(n1.x, (hm & 0xff000000u) | (n0.w >> 24));
    t.tmask = hm & 0x00ffffffu;
    t.tbase = n1.y;
}
// One node visit: pops the nearest pending child, intersects its 8 children; the triangles of the
// leaves that were hit are left in t.tmask / t.tbase.  Precondition: t.G has a pending child.
template <bool COUNT>
MQ_DEV void trav_node(const MqSceneDev& sc, Trav& t, uint2* stk /* &lds[0][lane] */, unsigned long long* spill, Ctr& ctr) {
    const uint4* np = trav_node_pop(sc, t, stk, spill);
    uint4 n0 = np[0], n1 = np[1], n2 = np[2], n3 = np[3], n4 = np[4];
    if (COUNT) ctr.nodes++;
    trav_node_test(t, n0, n1, n2, n3, n4);
}

// Tests ONE pending leaf record (lowest bit of t.tmask): one or two triangles that share an edge, as four vertices in four
// 16-byte loads (MqLeafRec).  Both triangles in ONE loop step: measured with the old 48-byte records (two triangles per step,
// six loads) the queue kernel took 8 % less time than with a step per triangle (profiles/r03_g_*).  TMIN / TMAX: the ray's
// interval (visibility rays); the closest-hit queries pass (0, MQ_T_MAX): `tt > 0` is part of tri_isect already.
struct LeafTris { f3 a0, a1, a2, b0, b1, b2; uint32_t key0, key1, tri0, sel; };
MQ_DEV LeafTris load_leaf(const MqSceneDev& sc, uint32_t rec) {
    typedef uint32_t u4v __attribute__((ext_vector_type(4)));
    const u4v* lp = (const u4v*)(sc.leaves + rec);
    u4v r0 = lp[0], r1 = lp[1], r2 = lp[2], r3 = lp[3];
    asm volatile("" : "+v"(r0), "+v"(r1), "+v"(r2), "+v"(r3)); // four 16-byte loads, as written
    LeafTris L;
    const f3 v0 = F3(__uint_as_float(r0.x), __uint_as_float(r0.y), __uint_as_float(r0.z)), v1 = F3(__uint_as_float(r0.w), __uint_as_float(r1.x), __uint_as_float(r1.y));
    const f3 v2 = F3(__uint_as_float(r1.z), __uint_as_float(r1.w), __uint_as_float(r2.x)), v3 = F3(__uint_as_float(r2.y), __uint_as_float(r2.z), __uint_as_float(r2.w));
    L.a0 = v0; L.a1 = v1; L.a2 = v2; L.key0 = r3.x; L.key1 = r3.y; L.tri0 = r3.z; L.sel = r3.w;
    L.b0 = v0; L.b1 = v2; L.b2 = v3; // the fan pattern (v0, v2, v3): nearly every record; anything else is selected below, wave-uniformly skipped when no lane needs it
    if (__ballot((r3.w & MQ_LEAF_HAS_B) && (r3.w & 0x3fu) != MQ_LEAF_SEL_FAN) != 0ull) {
        const uint32_t s0 = r3.w & 3u, s1 = (r3.w >> 2) & 3u, s2 = (r3.w >> 4) & 3u;
        L.b0 = s0 == 0u ? v0 : (s0 == 1u ? v1 : (s0 == 2u ? v2 : v3));
        L.b1 = s1 == 0u ? v0 : (s1 == 1u ? v1 : (s1 == 2u ? v2 : v3));
        L.b2 = s2 == 0u ? v0 : (s2 == 1u ? v1 : (s2 == 2u ? v2 : v3));
    }
    return L;
}
template <bool COUNT>
MQ_DEV void trav_tri(const MqSceneDev& sc, Trav& t, Ctr& ctr, float tmin = 0.0f, float tmax = MQ_T_MAX) {
    const uint32_t k = (uint32_t)__ffs((int)t.tmask) - 1u;
    t.tmask &= t.tmask - 1u;
    const LeafTris L = load_leaf(sc, t.tbase + k);
    if (COUNT) ctr.tris += (L.sel & MQ_LEAF_HAS_B) ? 2u : 1u;
    float tt = 0.0f, u = 0.0f, v = 0.0f;
    bool accept = tri_isect(t.o, t.d, L.a0, L.a1, L.a2, tt, u, v);
    accept = accept && tt > tmin && tt < tmax && (tt < t.hit.t || (tt == t.hit.t && L.key0 < t.best_key));
    if (accept && (L.sel & 0x10000u)) accept = anyhit_confirm(sc, L.tri0, u, v);
    if (accept) { t.hit.t = tt; t.hit.u = u; t.hit.v = v; t.hit.tri = L.tri0; t.best_key = L.key0; t.tlim = trav_limit(tt); }
    if (L.sel & MQ_LEAF_HAS_B) {
        accept = tri_isect(t.o, t.d, L.b0, L.b1, L.b2, tt, u, v);
        accept = accept && tt > tmin && tt < tmax && (tt < t.hit.t || (tt == t.hit.t && L.key1 < t.best_key));
        if (accept && (L.sel & 0x20000u)) accept = anyhit_confirm(sc, L.tri0 + 1u, u, v);
        if (accept) { t.hit.t = tt; t.hit.u = u; t.hit.v = v; t.hit.tri = L.tri0 + 1u; t.best_key = L.key1; t.tlim = trav_limit(tt); }
    }
}

// After the pending triangles are done: continue with the current group or pop the stack.
// Returns true when the traversal is complete.
MQ_DEV bool trav_next(Trav& t, uint2* stk, unsigned long long* spill) {
    if (t.G.y > 0x00ffffffu) return false;
    if (t.sp == t.sb) return true;
    t.sp--;
    t.G = stk[(t.sp < MQ_STACK_LDS ? t.sp : 0) * 64]; // the common case: one LDS read, no address select
    if (t.sp >= MQ_STACK_LDS) { unsigned long long e = spill[t.sp - MQ_STACK_LDS]; t.G = make_uint2((uint32_t)e, (uint32_t)(e >> 32)); } // deep stacks only
    return false;
}

// returns true when the traversal is complete
template <bool COUNT>
MQ_DEV bool trav_step(const MqSceneDev& sc, Trav& t, uint2* stk, unsigned long long* spill, Ctr& ctr) {
    trav_node<COUNT>(sc, t, stk, spill, ctr);
    while (t.tmask) trav_tri<COUNT>(sc, t, ctr);
    return trav_next(t, stk, spill);
}

template <bool COUNT>
MQ_DEV void traverse(const MqSceneDev& sc, f3 o, f3 d, RayHit& hit, uint2* stk, unsigned long long* spill, Ctr& ctr) {
    Trav t;
    trav_init(t, o, d);
    trav_defer(sc, t, stk);
    if (COUNT) ctr.rays++;
    if (sc.n_nodes != 0) while (!trav_step<COUNT>(sc, t, stk, spill, ctr)) {}
    hit = t.hit;
}

// closest hit within (tmin, tmax): visibility rays (trace_visibility_ray_init, raytrace.glsl:66-80)
template <bool COUNT>
MQ_DEV void traverse_range(const MqSceneDev& sc, f3 o, f3 d, float tmin, float tmax, RayHit& hit, uint2* stk, unsigned long long* spill, Ctr& ctr) {
    Trav t;
    trav_init(t, o, d);
    t.tlim = trav_limit(tmax);
    trav_defer(sc, t, stk);
    if (sc.n_nodes != 0) for (;;) {
        trav_node<COUNT>(sc, t, stk, spill, ctr);
        while (t.tmask) trav_tri<COUNT>(sc, t, ctr, tmin, tmax); // the ray's own interval
        if (trav_next(t, stk, spill)) break;
    }
    hit = t.hit;
}

// ------------------------------------------------------------------------------------------------
// sky + trace_ray shading (raytrace.glsl:25-65,156-311)
// ------------------------------------------------------------------------------------------------
MQ_DEV f3 get_sky(const MqSceneDev& sc, const MqParams& P, const mq_uniform& U, f3 w, f3 sun_color) {
    f3 sun = F3(P.sun_w[0], P.sun_w[1], P.sun_w[2]);
    float a = 0.5f * (1.0f + dot(sun, w));
    float a2 = a * a;
    float glow = 0.5f * (a2 * a2) + 5.0f * vmf_pdf(w, sun, 3000.0f);
    f3 emm = rh3(sun_color * rh(glow));
    if ((U.sky_lf_ft & 0xffffu) == 0xffffu) {
        float az = fabsf(w.z);
        float s = 0.5f + w.x / az, t = 0.5f + w.y / az;
        float tm = U.cl_time * 0.12f;
        f4 bck = tex_sample(sc, U.sky_rt_bk & 0xffffu, s + 0.5f * tm, t + 0.5f * tm);
        f4 fnt = tex_sample(sc, U.sky_rt_bk >> 16, s + tm, t + tm);
        f3 tex = F3(mmix(bck.r, fnt.r, fnt.a), mmix(bck.g, fnt.g, fnt.a), mmix(bck.b, fnt.b, fnt.a));
        emm = rh3(F3(10.0f * (mq_exp2(3.5f * rh(tex.x)) - 1.0f), 10.0f * (mq_exp2(3.5f * rh(tex.y)) - 1.0f), 10.0f * (mq_exp2(3.5f * rh(tex.z)) - 1.0f)));
    } else {
        float ax = fabsf(w.x), ay = fabsf(w.y), az = fabsf(w.z);
        int side_i;
        if (ax >= ay && ax >= az) side_i = w.x >= 0.0f ? 0 : 1; else if (ay >= az) side_i = w.y >= 0.0f ? 2 : 3; else side_i = w.z >= 0.0f ? 4 : 5;
        uint32_t side = 0; float s = 0.0f, t = 0.0f;
        switch (side_i) {
        case 0: side = U.sky_rt_bk & 0xffffu; s = 0.5f + 0.5f * -w.y / ax; t = 0.5f + 0.5f * -w.z / ax; break;
        case 1: side = U.sky_lf_ft & 0xffffu; s = 0.5f + 0.5f * w.y / ax; t = 0.5f + 0.5f * -w.z / ax; break;
        case 2: side = U.sky_rt_bk >> 16; s = 0.5f + 0.5f * w.x / ay; t = 0.5f + 0.5f * -w.z / ay; break;
        case 3: side = U.sky_lf_ft >> 16; s = 0.5f + 0.5f * -w.x / ay; t = 0.5f + 0.5f * -w.z / ay; break;
        case 4: side = U.sky_up_dn & 0xffffu; s = 0.5f + 0.5f * -w.y / az; t = 0.5f + 0.5f * w.x / az; break;
        default: side = U.sky_up_dn >> 16; s = 0.5f + 0.5f * -w.y / az; t = 0.5f + 0.5f * -w.x / az; break;
        }
        if (side < MQ_MAX_GLTEXTURES) { f4 tx = tex_sample(sc, side, s, t); emm = rh3(F3(emm.x + rh(tx.r), emm.y + rh(tx.g), emm.z + rh(tx.b))); }
    }
    return emm;
}

MQ_DEV f3 ld3(const float* p, uint32_t i) { return F3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }

// Shades the closest hit `rhit` of the ray (hit.pos, hit.wi).  Mirrors raytrace.glsl:166-311.
// FIRST: the g-buffer's first hit (MERIAN_QUAKE_FIRST_HIT, raytrace.glsl:153-156): r_x / r_y are the directions of
// the camera rays one pixel to the right / below and select the mip level of the albedo / emission fetch.
template <bool FIRST = false>
MQ_DEV void shade_hit(const MqSceneDev& sc, const MqParams& P, const mq_uniform& U, const RayHit& rhit,
                      f3& throughput, f3& contribution, Hit& hit, f3 sun_color, f3 r_x = F3(0, 0, 0), f3 r_y = F3(0, 0, 0)) {
    float tq = rhit.tri == MQ_NIL ? MQ_T_MAX : rhit.t;
    float tr = rh(transmittance(tq, U.cam_x[3], P.volume_max_t));
    throughput = rh3(throughput * tr);
    hit.roughness = rh(0.6f);
    uint32_t key = 0, tflags = 0;
    f3 p0, p1, p2;
    mq_ext e;
    MqTexDesc tx_albedo, tx_fb;
    uint32_t flags = 0;
    bool have = rhit.tri != MQ_NIL;
    if (have) {
        const uint4* tp = (const uint4*)(sc.tris + rhit.tri);
        uint4 a = tp[0], b = tp[1], c = tp[2];
        load_shade(sc, rhit.tri, e, tx_albedo, tx_fb); // independent of the triangle fetch: both go out together
        p0 = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
        p1 = F3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
        p2 = F3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
        key = c.y; tflags = c.z;
        flags = e.texnum_fb_flags >> 12;
    }
    if (!have || flags == MQ_MAT_FLAGS_SKY) { // :170-194
        f3 sky = get_sky(sc, P, U, hit.wi, sun_color);
        f3 add = rh3(throughput * sky);
        contribution = have ? rh3(contribution + add) : add;
        hit.albedo = sky;
        hit.pos = hit.pos + hit.wi * MQ_T_MAX; hit.prev_pos = hit.pos;
        hit.normal = -hit.wi; hit.enc_geonormal = encode_normal(hit.normal);
        return;
    }
    float b0 = 1.0f - rhit.u - rhit.v, b1 = rhit.u, b2 = rhit.v;
    float st0s = h2f(e.st[0]), st0t = h2f(e.st[1]), st1s = h2f(e.st[2]), st1t = h2f(e.st[3]), st2s = h2f(e.st[4]), st2t = h2f(e.st[5]);
    float s = st0s * b0 + st1s * b1 + st2s * b2, t = st0t * b0 + st1t * b1 + st2t * b2;
    if (flags > 0 && flags < 5) { // :198-204
        float ws = s + 0.125f * mq_sin(8.0f * t + U.cl_time), wt = t + 0.125f * mq_sin(8.0f * s + U.cl_time);
        s = ws; t = wt;
        if (flags == MQ_MAT_FLAGS_WATER) {
            float as = 0.02f * mq_sin(20.0f * t + 1.7f * U.cl_time), at = 0.02f * mq_sin(20.0f * s + 1.3f * U.cl_time);
            s += as; t += at;
            hit.roughness = rh(0.4f);
        }
    }
    hit.pos = (p0 * b0 + p1 * b1) + p2 * b2;
    f3 du = p2 - p0, dv = p1 - p0;
    hit.normal = normalize(cross(du, dv));
    hit.enc_geonormal = encode_normal(hit.normal);
    if (tflags & MQ_TRI_DYNAMIC) { // :226-228
        const MqGeoDev& g = sc.geo[key >> 28];
        uint32_t prim = key & 0x0fffffffu;
        uint32_t i0 = g.idx[3 * prim], i1 = g.idx[3 * prim + 1], i2 = g.idx[3 * prim + 2];
        hit.prev_pos = (ld3(g.prev_vtx, i0) * b0 + ld3(g.prev_vtx, i1) * b1) + ld3(g.prev_vtx, i2) * b2;
    } else hit.prev_pos = hit.pos;
    // :232-239 texture-space footprint of the pixel: Igehy transfer of {dO = 0, dD = r} over the hit distance onto
    // the triangle's plane, pseudoinverse of the edge matrix [du dv] by the normal equations, half-precision
    // texture-coordinate edge differences (f16mat2 st_dudv, :208-209) -- same operation order as the oracle
    float gxs = 0.0f, gxt = 0.0f, gys = 0.0f, gyt = 0.0f;
    const bool use_grad = FIRST && (P.enable_albedo_mipmap || P.enable_emission_mipmap);
    if (use_grad) {
        const f3 n = hit.normal, D = hit.wi;
        const float dn = dot(D, n);
        const float a = dot(du, du), b = dot(du, dv), cc = dot(dv, dv);
        const float det = a * cc - b * b;
        const float d0x = rh(st2s - st0s), d0y = rh(st2t - st0t), d1x = rh(st1s - st0s), d1y = rh(st1t - st0t);
#pragma unroll
        for (int k = 0; k < 2; k++) {
            f3 dO = (k == 0 ? r_x : r_y) * rhit.t;
            dO = dO - D * (dot(dO, n) / dn);
            const float pu = dot(du, dO), pv = dot(dv, dO);
            const float ca = (cc * pu - b * pv) / det, cb = (a * pv - b * pu) / det;
            const float gs = d0x * ca + d1x * cb, gt = d0y * ca + d1y * cb;
            if (k == 0) { gxs = gs; gxt = gt; } else { gys = gs; gyt = gt; }
        }
    }
    f4 at = (use_grad && P.enable_albedo_mipmap) ? tex_sample_grad_desc(sc, tx_albedo, s, t, gxs, gxt, gys, gyt) : tex_sample_desc(sc, tx_albedo, s, t);
    f3 albedo_tex = rh3(F3(mq_pow(rh(at.r), 1.0f / 1.2f), mq_pow(rh(at.g), 1.0f / 1.2f), mq_pow(rh(at.b), 1.0f / 1.2f)));
    if (e.n1_brush == 0xffffffffu) { // :249-274
        uint32_t tn_norm = e.n0_gloss_norm >> 16, tn_gloss = e.n0_gloss_norm & 0xffffu;
        if (tn_norm > 0 && tn_norm < MQ_MAX_GLTEXTURES) {
            f4 nt = tex_sample(sc, tn_norm, s, t);
            f3 tn = F3((nt.r - 0.5f) * 2.0f, (nt.g - 0.5f) * 2.0f, (nt.b - 0.5f) * 2.0f);
            float d0x = rh(st2s - st0s), d0y = rh(st2t - st0t), d1x = rh(st1s - st0s), d1y = rh(st1t - st0t);
            float det = rh(rh(d0x * d1y) - rh(d1x * d0y));
            if (fabsf(det) > 1e-8f) {
                f3 du2 = normalize((du * d1y - dv * d0y) * (1.0f / det));
                dv = -normalize((du * (-d1x) + dv * d0x) * (1.0f / det));
                du = du2;
            }
            f3 gn = hit.normal;
            hit.normal = normalize((du * tn.x + dv * tn.y) + gn * tn.z);
            f3 r = hit.wi - hit.normal * (2.0f * dot(hit.wi, hit.normal));
            if (dot(r, gn) < 0.0f) hit.normal = normalize(-hit.wi + normalize(r - gn * dot(gn, r)));
        }
        if (tn_gloss > 0 && tn_gloss < MQ_MAX_GLTEXTURES) hit.roughness = rh(tex_sample(sc, tn_gloss, s, t).r);
    } else if (flags == MQ_MAT_FLAGS_SOLID) { // :275-278
        uint32_t a = e.n0_gloss_norm, b = e.n1_brush;
        hit.albedo = rh3(F3(rh((float)(a & 0xff)) / 255.0f, rh((float)((a >> 8) & 0xff)) / 255.0f, rh((float)((a >> 16) & 0xff)) / 255.0f));
        f3 em = ldr_to_hdr(rh3(F3(rh((float)(b & 0xff)) / 255.0f, rh((float)((b >> 8) & 0xff)) / 255.0f, rh((float)((b >> 16) & 0xff)) / 255.0f)));
        contribution = rh3(contribution + rh3(throughput * em));
        return;
    }
    if (flags == MQ_MAT_FLAGS_WATERFALL) { // :288-310
        hit.albedo = albedo_tex;
        contribution = rh3(contribution + rh3(throughput * hit.albedo));
    } else if (flags == MQ_MAT_FLAGS_SPRITE || flags == MQ_MAT_FLAGS_TELE) {
        hit.albedo = ldr_to_hdr(albedo_tex);
        contribution = rh3(contribution + rh3(throughput * hit.albedo));
    } else {
        uint32_t fb = e.texnum_fb_flags & 0xfffu;
        hit.albedo = albedo_tex;
        if (fb > 0 && fb < MQ_MAX_GLTEXTURES) {
            f4 ft = (use_grad && P.enable_emission_mipmap) ? tex_sample_grad_desc(sc, tx_fb, s, t, gxs, gxt, gys, gyt) : tex_sample_desc(sc, tx_fb, s, t);
            f3 em = ldr_to_hdr(rh3(F3(ft.r, ft.g, ft.b)));
            if (em.x > 0.0f || em.y > 0.0f || em.z > 0.0f) {
                contribution = rh3(contribution + rh3(throughput * em));
                hit.albedo = em;
            }
        }
    }
}

// ------------------------------------------------------------------------------------------------
// hash-grid addressing (mc.glsl:62-88,117-121; light_cache.glsl:13-29)
// ------------------------------------------------------------------------------------------------
// log_power = mq_log(power), inv_power = 1 / power: from the host (MqParams), the same floats the expressions give here
MQ_DEV uint32_t grid_level(int type, float steps, float tan_half, float minw, float log_power, float inv_power, f3 cam, f3 pos) {
    float w = 2.0f * tan_half * length(cam - pos);
    float lv;
    if (type == 0) lv = steps * mq_log(mmax(w, minw) / minw) / log_power;
    else lv = steps * mq_pow(mmax(w - minw, 0.0f), inv_power);
    return (uint32_t)floorf(lv + 0.5f);
}
MQ_DEV float mc_inv_width(const MqParams& P, uint32_t level) {
    if (level < MQ_WIDTH_LUT) return P.mc_inv_width_lut[level];
    return 1.0f / grid_width(P.adaptive_grid_type, P.mc_adaptive_grid_steps_per_unit_size, P.mc_adaptive_grid_min_width, P.mc_adaptive_grid_power, level);
}
MQ_DEV float lc_inv_width(const MqParams& P, uint32_t level) {
    if (level < MQ_WIDTH_LUT) return P.lc_inv_width_lut[level];
    return 1.0f / grid_width(P.lc_grid_type, P.lc_grid_steps_per_unit_size, P.lc_grid_min_width, P.lc_grid_power, level);
}
MQ_DEV f3 cam_pos(const mq_uniform& U) { return F3(U.cam_x[0], U.cam_x[1], U.cam_x[2]); }

// the level of the adaptive grid at `pos` before the jitter (mc.glsl:62-68): a square root, two logarithms and two
// divisions that depend on the position only -- the K lookups of one path vertex compute it once
MQ_DEV uint32_t mc_adaptive_base_level(const MqParams& P, const mq_uniform& U, f3 pos) {
    return grid_level(P.adaptive_grid_type, P.mc_adaptive_grid_steps_per_unit_size, P.mc_adaptive_grid_tan_alpha_half, P.mc_adaptive_grid_min_width, P.mc_log_power, P.mc_inv_power, cam_pos(U), pos);
}
MQ_DEV void mc_adaptive_buffer_index_at(const MqParams& P, uint32_t base_level, uint32_t& rng, f3 pos, f3 normal, uint32_t& index, uint32_t& hash16) {
    uint32_t level = base_level + level_jitter(xorshift(rng)); // mc.glsl:70
    i3 g = grid_idx_interpolate(pos, mc_inv_width(P, level), xorshift(rng));
    index = hash_grid_normal_level(g, normal, level, P.mc_adaptive_buffer_size);
    hash16 = hash2_grid_level(g, level) & 0xffffu;
}
MQ_DEV void mc_adaptive_buffer_index(const MqParams& P, const mq_uniform& U, uint32_t& rng, f3 pos, f3 normal, uint32_t& index, uint32_t& hash16) {
    uint32_t level = mc_adaptive_base_level(P, U, pos);
    level += level_jitter(xorshift(rng)); // mc.glsl:70
    i3 g = grid_idx_interpolate(pos, mc_inv_width(P, level), xorshift(rng));
    index = hash_grid_normal_level(g, normal, level, P.mc_adaptive_buffer_size);
    hash16 = hash2_grid_level(g, level) & 0xffffu;
}
MQ_DEV void mc_static_buffer_index(const MqParams& P, uint32_t& rng, f3 pos, uint32_t& index, uint32_t& hash16) {
    i3 g = grid_idx_interpolate(pos, P.mc_static_inv_width, xorshift(rng));
    index = hash_grid(g, P.mc_static_buffer_size) + P.mc_adaptive_buffer_size;
    hash16 = hash2_grid(g) & 0xffffu;
}

// Markov-chain state in registers
struct MCS { f3 w_tgt; float sum_w, w_cos, T; uint32_t id, N, hash; uint16_t mv[3]; };

MQ_DEV MCS mc_load(const MqMCState* mc, uint32_t i) {
    const uint4* p = (const uint4*)(mc + i);
    uint4 a = p[0], b = p[1];
    uint2 c = *(const uint2*)(p + 2);
    MCS s;
    s.w_tgt = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)); s.sum_w = __uint_as_float(a.w);
    s.w_cos = __uint_as_float(b.x); s.T = __uint_as_float(b.y); s.id = b.z; s.N = b.w & 0xffffu; s.hash = b.w >> 16;
    s.mv[0] = (uint16_t)(c.x & 0xffffu); s.mv[1] = (uint16_t)(c.x >> 16); s.mv[2] = (uint16_t)(c.y & 0xffffu);
    return s;
}
MQ_DEV void mc_store(MqMCState* mc, uint32_t i, const MCS& s) {
    uint4* p = (uint4*)(mc + i);
    p[0] = make_uint4(__float_as_uint(s.w_tgt.x), __float_as_uint(s.w_tgt.y), __float_as_uint(s.w_tgt.z), __float_as_uint(s.sum_w));
    p[1] = make_uint4(__float_as_uint(s.w_cos), __float_as_uint(s.T), s.id, (s.N & 0xffffu) | (s.hash << 16));
    *(uint2*)(p + 2) = make_uint2((uint32_t)s.mv[0] | ((uint32_t)s.mv[1] << 16), (uint32_t)s.mv[2]);
}
MQ_DEV f3 mc_state_pos(const MCS& s) { return s.sum_w > 0.0f ? s.w_tgt * (1.0f / s.sum_w) : s.w_tgt; }
MQ_DEV f3 mc_state_dir(const MCS& s, f3 pos) { return normalize(mc_state_pos(s) - pos); }
MQ_DEV float mc_state_mean_cos(const MqParams& P, const MCS& s, f3 pos) { // mc.glsl:24-26
    f3 d = pos - mc_state_pos(s);
    float prior = mmax(0.0001f, P.dir_guide_prior / dot(d, d));
    uint32_t nn = s.N * s.N;
    if (P.quirk_n16_wrap) nn &= 0xffffu;
    float n2 = (float)nn;
    return (n2 * mclamp(s.w_cos / s.sum_w, 0.0f, 0.9999999f)) / (n2 + prior);
}
MQ_DEV float mc_state_kappa(const MqParams& P, const MCS& s, f3 pos) { // mc.glsl:43-46
    float r = mc_state_mean_cos(P, s, pos);
    return (3.0f * r - r * r * r) / (1.0f - r * r);
}
MQ_DEV MCS mc_state_new(uint32_t& rng) { // mc.glsl:17
    MCS s;
    s.id = (uint32_t)(xorshift(rng) * 4294967296.0f);
    s.w_tgt = F3(0.0f, 0.0f, 0.0f); s.sum_w = 0.0f; s.w_cos = 0.0f; s.T = 0.0f; s.N = 0; s.hash = 0; s.mv[0] = s.mv[1] = s.mv[2] = 0;
    return s;
}
MQ_DEV void mc_finalize_load(const mq_uniform& U, MCS& s, uint32_t hash16, bool is_static, f3 pos, f3 normal) { // mc.glsl:90-96,130-135
    bool bad = s.sum_w < 0.0f || hash16 != s.hash;
    if (!bad && is_static) bad = !(dot(normal, mc_state_dir(s, pos)) > 0.0f);
    if (bad) s.sum_w = 0.0f;
    float k = s.sum_w * (U.cl_time - s.T);
    s.w_tgt = s.w_tgt + F3(h2f(s.mv[0]), h2f(s.mv[1]), h2f(s.mv[2])) * k;
}

// ---- learning-write log (test hook, property "debug: log learning writes"; record layouts: include/mq.h) ----
MQ_DEV void learn_log(const MqFrame& F, uint4 a, uint4 b, uint4 c, uint4 d) {
    const uint32_t at = atomicAdd(F.learn_log_count, 1u);
    if (at < F.learn_log_cap) { uint4* e = F.learn_log + 4 * (size_t)at; e[0] = a; e[1] = b; e[2] = c; e[3] = d; }
}
MQ_DEV void learn_log_simple(const MqFrame& F, uint32_t kind, uint32_t index, uint32_t a, uint32_t b, uint32_t c, uint32_t d) {
    learn_log(F, make_uint4(a, b, c, d), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, 0u, 0u), make_uint4(0u, 0u, index, kind));
}

// ---- light cache (light_cache.glsl:31-84) ------------------------------------------------------
MQ_DEV void lc_address(const MqParams& P, uint32_t& rng, uint32_t level, f3 pos, f3 normal, uint32_t& idx, uint32_t& chk) {
    i3 g = grid_idx_interpolate(pos, lc_inv_width(P, level), xorshift(rng));
    idx = hash_grid_normal_level(g, normal, level, P.lc_buffer_size);
    chk = hash2_grid_level(g, level);
}
MQ_DEV void light_cache_get_level(const MqParams& P, const MqLCCell* lc, uint32_t& rng, f3& irr, uint32_t& N, uint32_t level, f3 pos, f3 normal) {
    uint32_t idx, chk;
    lc_address(P, rng, level, pos, normal, idx, chk);
    uint4 c = *(const uint4*)(lc + idx);
    uint16_t i0 = (uint16_t)(c.z & 0xffffu), i1 = (uint16_t)(c.z >> 16), i2 = (uint16_t)(c.w & 0xffffu);
    if (c.x == chk && !h_bad(i0) && !h_bad(i1) && !h_bad(i2)) { irr = F3(h2f(i0), h2f(i1), h2f(i2)); N = c.w >> 16; }
    else { irr = F3(0.0f, 0.0f, 0.0f); N = 0; }
}
MQ_DEV uint32_t lc_level(const MqParams& P, const mq_uniform& U, f3 pos) {
    return grid_level(P.lc_grid_type, P.lc_grid_steps_per_unit_size, P.lc_grid_tan_alpha_half, P.lc_grid_min_width, P.lc_log_power, P.lc_inv_power, cam_pos(U), pos);
}
MQ_DEV f3 light_cache_get(const MqParams& P, const mq_uniform& U, const MqLCCell* lc, uint32_t& rng, f3 pos, f3 normal) {
    f3 irr; uint32_t N;
    light_cache_get_level(P, lc, rng, irr, N, lc_level(P, U, pos), pos, normal);
    return irr;
}
// light_cache.glsl:54-84.  The reference takes a per-cell try-lock (atomicExchange of the frame number), DROPS the update
// when the lock is contended and writes the cell's members one by one while it holds the lock.  Here (default) there is no
// lock word traffic: every writer publishes its result with 8-byte single-copy-atomic stores -- the (irradiance, N) payload
// in ONE store, a re-keyed cell as two (key first) -- so a cell never holds the halves of two writers' payloads, and of N
// writers that race on a cell one update survives (the last store) where the reference keeps the first and cancels N - 1:
// the same number of updates applied, the same N.  "LC try-lock" (and "debug: LC lock statistics", which adds the
// reference's per-cell counters) runs the reference's protocol itself; measured cost: DESIGN.md section 3.
// Frame 0 cancels every update, as the reference does (its zero-initialised lock word equals params.frame,
// light_cache.glsl:59-64).
typedef unsigned long long mq_u64;
// Scope of the publishing stores.  An AGENT-scope atomic store is a write-through on this multi-die part (global_store ... sc1:
// the per-die L2s are not coherent with each other inside a kernel) and cost +0.22 ms per 1080p frame in the terminal shading
// launches; a WORKGROUP-scope atomic store is the same single 8-byte store instruction without the write-through -- the
// non-tearing property is the instruction's, the visibility to other dies is that of every other store of these kernels
// (the tables are read "eventually": the reference races on them too).  profiles/r03_b_lc_publish.txt
#ifndef MQ_LC_PUBLISH
#define MQ_LC_PUBLISH 1
#endif
#if MQ_LC_PUBLISH == 0
#define MQ_LC_SCOPE __HIP_MEMORY_SCOPE_AGENT
#else
#define MQ_LC_SCOPE __HIP_MEMORY_SCOPE_WORKGROUP
#endif
MQ_DEV void light_cache_update(const MqParams& P, const MqFrame& F, uint32_t& rng, f3 pos, f3 normal, f3 irr, Ctr& ctr) {
    const mq_uniform& U = F.u; MqLCCell* const lc = F.lc;
    uint32_t level = lc_level(P, U, pos), idx, chk;
    lc_address(P, rng, level, pos, normal, idx, chk);
    const bool stats = P.lc_lock_protocol && F.lc_stats;
    if (U.frame == 0u) { ctr.lc_cancel++; if (stats) atomicAdd(&F.lc_stats[idx].y, 1u); return; }
    MqLCCell* cell = lc + idx;
    const bool locked = (P.lc_try_lock || stats) && !P.freeze_learning; // the reference's try-lock, light_cache.glsl:59-64
    if (locked && atomicExch(&cell->lock, U.frame) == U.frame) { if (stats) atomicAdd(&F.lc_stats[idx].y, 1u); ctr.lc_cancel++; return; }
    uint4 c = *(const uint4*)cell;
    uint16_t i0 = (uint16_t)(c.z & 0xffffu), i1 = (uint16_t)(c.z >> 16), i2 = (uint16_t)(c.w & 0xffffu);
    f3 cur; uint32_t N;
    const bool rekey = c.x != chk || h_bad(i0) || h_bad(i1) || h_bad(i2);
    if (rekey) { // :68-75 seed from the next coarser level
        f3 ci; uint32_t cn;
        light_cache_get_level(P, lc, rng, ci, cn, level + 1, pos, normal);
        cur = rh3(ci); N = cn;
    } else { cur = F3(h2f(i0), h2f(i1), h2f(i2)); N = c.w >> 16; }
    N = N + 1 < MQ_LC_MAX_N ? N + 1 : MQ_LC_MAX_N;
    float a = mmax(1.0f / (float)N, MQ_LC_MIN_ALPHA);
    uint32_t o0 = f2h(mmix(cur.x, irr.x, a)), o1 = f2h(mmix(cur.y, irr.y, a)), o2 = f2h(mmix(cur.z, irr.z, a));
    const uint32_t nz = o0 | (o1 << 16), nw = o2 | (N << 16);
    if (P.log_learning) learn_log_simple(F, 2u, idx, chk, rekey ? 1u : 0u, nz, nw);
    if (P.freeze_learning) return;
    // (scattered read-modify-write atomics run at about 20 G/s on this chip; plain 8-byte stores are free of that limit)
    mq_u64* const c64 = (mq_u64*)__builtin_assume_aligned(cell, 16); // cells are 16-byte records in a hipMalloc'ed table
#if MQ_LC_PUBLISH == 2 // round 2's publish, kept for the A/B: one 16-byte store for a re-keyed cell -- tears at dword granularity
    if (rekey) *(uint4*)cell = make_uint4(chk, locked ? U.frame : 0u, nz, nw);
    else *(uint2*)&cell->irr[0] = make_uint2(nz, nw);
#else
    if (rekey) __hip_atomic_store(c64, (mq_u64)chk | ((mq_u64)(locked ? U.frame : 0u) << 32), __ATOMIC_RELAXED, MQ_LC_SCOPE);
    __hip_atomic_store(c64 + 1, (mq_u64)nz | ((mq_u64)nw << 32), __ATOMIC_RELAXED, MQ_LC_SCOPE);
#endif
    if (stats) atomicAdd(&F.lc_stats[idx].x, 1u);
    if (locked) { __threadfence(); cell->lock = 0u; } // :82-83
    ctr.lc_ok++;
}

MQ_DEV void store_chit(uint32_t* dst, const Hit& h) { // hit.glsl.h:34-43, 40-byte record
    uint32_t m0 = f2h(h.pos.x - h.prev_pos.x), m1 = f2h(h.pos.y - h.prev_pos.y), m2 = f2h(h.pos.z - h.prev_pos.z);
    uint2* d2 = (uint2*)dst; // 40-byte records are 8-byte aligned
    d2[0] = make_uint2(__float_as_uint(h.pos.x), __float_as_uint(h.pos.y));
    d2[1] = make_uint2(__float_as_uint(h.pos.z), m0 | (m1 << 16));
    d2[2] = make_uint2(m2, encode_normal(h.wi));
    d2[3] = make_uint2(encode_normal(h.normal), h.enc_geonormal);
    d2[4] = make_uint2((uint32_t)f2h(h.albedo.x) | ((uint32_t)f2h(h.albedo.y) << 16), (uint32_t)f2h(h.albedo.z) | ((uint32_t)f2h(h.roughness) << 16));
}
MQ_DEV void load_chit(const uint32_t* src, Hit& h) { // hit.glsl.h:45-53
    const uint2* s2 = (const uint2*)src;
    uint2 a = s2[0], b = s2[1], c = s2[2], d = s2[3], e = s2[4];
    h.pos = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(b.x));
    h.prev_pos = F3(h.pos.x - h2f((uint16_t)(b.y & 0xffffu)), h.pos.y - h2f((uint16_t)(b.y >> 16)), h.pos.z - h2f((uint16_t)(c.x & 0xffffu)));
    h.wi = decode_normal(c.y); h.normal = decode_normal(d.x); h.enc_geonormal = d.y;
    h.albedo = F3(h2f((uint16_t)(e.x & 0xffffu)), h2f((uint16_t)(e.x >> 16)), h2f((uint16_t)(e.y & 0xffffu)));
    h.roughness = h2f((uint16_t)(e.y >> 16));
}

// ------------------------------------------------------------------------------------------------
// Wavefront pipeline.  One frame =
//     mq_primary_kernel                      first hit (g-buffer node) + first direction choice
//     R x { mq_trace_queue_kernel ;          closest hit for every queued ray (lean, high occupancy)
//           mq_bounce_kernel }               shade the hit, learn, finish samples, choose next direction
// with R = spp * (max_path_length - 1) rounds.  Live paths are COMPACTED between rounds: a kernel
// appends the paths that continue to the next round's queue with one aggregated atomic per wave
// (ballot + prefix popcount), so every kernel runs on dense, uniform work and the traversal kernel
// keeps its own small register footprint.  Path state lives in a 160-byte record per pixel slot.
//
// Measured motivation (profiles/r01_*): fused into one persistent megakernel the same work took
// 6.3 ms per 1080p frame (mixed-state divergence, 2-3 waves/SIMD), while the frame's rays alone
// traverse in about 1.4 ms when the traversal runs as its own kernel.
// ------------------------------------------------------------------------------------------------
// Appends a Markov-chain update (mc_state_add_sample + send_update_to_buffer, mc.glsl:159-222).
MQ_DEV void enqueue_update(const MqParams& P, const MqFrame& F, uint32_t& rng, uint32_t index, uint32_t id, f3 pos, float w, f3 target, f3 target_mv, f3 normal, Ctr& ctr) {
    const mq_uniform& U = F.u;
    if (index == MQ_NIL) { uint32_t h16; mc_adaptive_buffer_index(P, U, rng, pos, normal, index, h16); }
    const uint32_t mv01 = (uint32_t)f2h(target_mv.x) | ((uint32_t)f2h(target_mv.y) << 16), mv2 = (uint32_t)f2h(target_mv.z);
    const uint4 r0 = make_uint4(__float_as_uint(pos.x), __float_as_uint(pos.y), __float_as_uint(pos.z), __float_as_uint(w));
    const uint4 r1 = make_uint4(__float_as_uint(target.x), __float_as_uint(target.y), __float_as_uint(target.z), id);
    const uint4 r2 = make_uint4(__float_as_uint(normal.x), __float_as_uint(normal.y), __float_as_uint(normal.z), __float_as_uint(U.cl_time));
    if (P.log_learning && P.freeze_learning) learn_log(F, r0, r1, r2, make_uint4(mv01, mv2, index, 1u)); // proposed, never queued: no arrival rank
    if (P.freeze_learning) return;
    // Cap (mc.glsl:169-184): the returning increment of the slot's counter is this update's arrival
    // rank; ranks >= MQ_MAX_UPDATES are dropped.  The queue position is allocated at the same time
    // (wave-aggregated append to this wave's shard), so the two atomics overlap and only ONE memory
    // round trip sits on the path; a dropped update leaves a 16-byte "no slot" marker in its position.
    uint32_t uq = 0, rank_in_slot = 0;
    {
        unsigned long long m = __ballot(1); // lanes arrive here divergently
        const int lane = threadIdx.x & 63;
        const uint32_t shard = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (MQ_SHARDS - 1);
        int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&F.ctrl[MQ_CTRL_UPDATES + shard * MQ_SHARD_STRIDE], (uint32_t)__popcll(m));
        rank_in_slot = atomicAdd(&F.upd_count[index], 1u);
        base = __shfl(base, leader, 64);
        uq = shard_pos(shard, base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
    }
    if (P.log_learning) learn_log(F, r0, r1, r2, make_uint4(mv01, mv2 | ((rank_in_slot < 0xffffu ? rank_in_slot : 0xffffu) << 16), index, 1u)); // with the arrival rank (>= 10: dropped by the cap)
    if (uq < F.queue_cap) {
        uint4* e = (uint4*)(F.queue + uq);
        if (rank_in_slot < MQ_MAX_UPDATES) {
            e[0] = r0; e[1] = r1; e[2] = r2;
            e[3] = make_uint4(mv01, mv2 | (rank_in_slot << 16), index, 0u); // the arrival rank travels with the entry: the update pass replays by rank
        } else {
            e[3] = make_uint4(0u, 0u, MQ_NIL, 0u);
            ctr.upd_drop++;
        }
    } else { // no room in the queue (never expected: mq_api.cpp sizes it for every segment of the frame plus shard slack): flagged, and the slot's counter is given back
        atomicOr(&F.ctrl[0], 2u);
        atomicSub(&F.upd_count[index], 1u);
    }
}

struct Path {
    Hit cur;
    f3 thr, fval, irr, wo;
    float pp, m2, wo_p, bsdf, wodotn, score_sum, mc_sum_w;
    uint32_t rng, px, py, mc_index, mc_id;
    int seg, smp;
    bool lm_dir_ok;
};

// Path records are FIELD-MAJOR: 16-byte field k of slot s sits at paths[k * n_slots + s], so the loads and
// stores of a wave (consecutive slots) are contiguous.  A record holds only what a path needs AFTER its ray
// returns (7 fields = 112 bytes): the previous vertex's position / incoming direction / normal / albedo, the
// estimator's accumulators and the pre-trace sampling data.  The outgoing direction is read back from the ray
// buffer and its cosine recomputed (same expression, same bits); the vertex's previous position, geometric
// normal and roughness, and the sample's f value are dead once the ray is emitted.
#define MQ_PATH_FIELDS 7
MQ_DEV void store_path(uint4* dst /* paths + slot */, size_t n, const Path& p) {
    dst[0 * n] = make_uint4(__float_as_uint(p.cur.pos.x), __float_as_uint(p.cur.pos.y), __float_as_uint(p.cur.pos.z), __float_as_uint(p.cur.wi.x));
    dst[1 * n] = make_uint4(__float_as_uint(p.cur.wi.y), __float_as_uint(p.cur.wi.z), __float_as_uint(p.cur.normal.x), __float_as_uint(p.cur.normal.y));
    dst[2 * n] = make_uint4(__float_as_uint(p.cur.normal.z), (uint32_t)f2h(p.cur.albedo.x) | ((uint32_t)f2h(p.cur.albedo.y) << 16),
                            (uint32_t)f2h(p.cur.albedo.z) | ((uint32_t)p.seg << 16) | ((uint32_t)p.smp << 21) | (p.lm_dir_ok ? 0x20000000u : 0u), p.rng);
    dst[3 * n] = make_uint4(__float_as_uint(p.thr.x), __float_as_uint(p.thr.y), __float_as_uint(p.thr.z), __float_as_uint(p.pp));
    dst[4 * n] = make_uint4(__float_as_uint(p.irr.x), __float_as_uint(p.irr.y), __float_as_uint(p.irr.z), __float_as_uint(p.m2));
    dst[5 * n] = make_uint4(p.px | (p.py << 16), __float_as_uint(p.wo_p), __float_as_uint(p.bsdf), __float_as_uint(p.score_sum));
    dst[6 * n] = make_uint4(__float_as_uint(p.mc_sum_w), p.mc_index, p.mc_id, 0u);
}
MQ_DEV void load_path(const uint4* src /* paths + slot */, size_t n, float4 ray_dir, Path& p) {
    uint4 a = src[0], b = src[n], c = src[2 * n], d = src[3 * n], e = src[4 * n], f = src[5 * n], g = src[6 * n];
    p.cur.pos = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z));
    p.cur.wi = F3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
    p.cur.normal = F3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
    p.cur.albedo = F3(h2f((uint16_t)(c.y & 0xffffu)), h2f((uint16_t)(c.y >> 16)), h2f((uint16_t)(c.z & 0xffffu)));
    p.cur.prev_pos = p.cur.pos; p.cur.enc_geonormal = 0; p.cur.roughness = 0.0f; // not kept: dead after the ray was emitted
    p.seg = (int)((c.z >> 16) & 0x1fu); p.smp = (int)((c.z >> 21) & 0xffu); p.lm_dir_ok = (c.z & 0x20000000u) != 0;
    p.rng = c.w;
    p.thr = F3(__uint_as_float(d.x), __uint_as_float(d.y), __uint_as_float(d.z)); p.pp = __uint_as_float(d.w);
    p.irr = F3(__uint_as_float(e.x), __uint_as_float(e.y), __uint_as_float(e.z)); p.m2 = __uint_as_float(e.w);
    p.px = f.x & 0xffffu; p.py = f.x >> 16; p.wo_p = __uint_as_float(f.y); p.bsdf = __uint_as_float(f.z); p.score_sum = __uint_as_float(f.w);
    p.mc_sum_w = __uint_as_float(g.x); p.mc_index = g.y; p.mc_id = g.z;
    p.fval = F3(0, 0, 0);
    p.wo = F3(ray_dir.x, ray_dir.y, ray_dir.z);
    p.wodotn = dot(p.wo, p.cur.normal); // as computed when the direction was chosen
}

MQ_DEV void flush_counters(MqCountersDev* g, const Ctr& c) {
    const uint32_t v[12] = {c.rays, c.nodes, c.tris, c.segments, c.guided, c.lc, c.upd_ok, c.upd_drop, c.mc_reads, c.pixels, c.lc_ok, c.lc_cancel};
    unsigned long long* dst = (unsigned long long*)g;
#pragma unroll
    for (int i = 0; i < 12; i++) {
        uint32_t x = v[i];
        for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
        if ((threadIdx.x & 63) == 0 && x) atomicAdd(dst + i, (unsigned long long)x);
    }
}

// Runs the "choose next direction / finish sample / restart / finish pixel" logic of
// mcpg.comp:54-137,193-210 until the path either has a ray to trace (returns true; ro/rd in
// path.wo and the caller derives the origin) or the pixel is complete (returns false, outputs written).
template <bool GUIDED, bool COUNT>
MQ_DEV bool advance_path(const MqParams& P, const MqFrame& F, Path& p, uint32_t slot, bool need_dir, bool sample_done, float* lobes /* LDS, &buf[0][lane] */, Ctr& ctr) {
    const mq_uniform& U = F.u;
    const int K = P.mc_samples < MQ_MAX_MC_SAMPLES ? P.mc_samples : MQ_MAX_MC_SAMPLES;
    const size_t pidx = (size_t)p.py * F.W + p.px;
    while (need_dir || sample_done) {
        if (need_dir) {
            need_dir = false;
            if (COUNT) ctr.segments++;
            PLAP(ctr, 4);
            const float alpha = roughness_to_alpha(p.cur.roughness);
            f3 wo;
            bool rejected = false;
            if (!GUIDED) { // mcpg.comp:59-64
                float x0 = xorshift(p.rng), x1 = xorshift(p.rng), x2 = xorshift(p.rng);
                wo = bsdf_sample(p.cur.wi, p.cur.normal, alpha, x0, x1, x2);
                p.wodotn = dot(wo, p.cur.normal);
                if (p.wodotn <= 1e-3f || dot(wo, decode_normal(p.cur.enc_geonormal)) <= 1e-3f) rejected = true;
                else p.wo_p = bsdf_pdf(p.cur.wi, wo, p.cur.normal, alpha);
            } else { // mcpg.comp:67-137
                if (COUNT) ctr.guided++;
                // The K lookups run as a rolled loop (code size: the instruction cache is the scarce
                // resource of these kernels); the lobes (score, direction, kappa) live in LDS at
                // lobes[(6 * i + c) * 64] so that K may be a run-time value without indexed VGPRs.
                // The 48-byte state of lookup i + 1 is requested before state i is processed.
                const f3 lp = p.smp == 0 ? p.cur.prev_pos : p.cur.pos;
                const uint32_t base_level = K > 0 ? mc_adaptive_base_level(P, U, lp) : 0u;
                p.score_sum = 0.0f; p.mc_index = MQ_NIL; p.mc_id = 0; p.mc_sum_w = 0.0f;
                MCS sel = {};
                uint32_t bi_next = 0, h16_next = 0; bool adapt_next = false; float xsel_next = 0.0f;
                MCS st_next = {};
#pragma nounroll
                for (int i = -1; i < K; i++) { // iteration i requests state i + 1, then processes state i
                    MCS st = st_next;
                    const uint32_t bi = bi_next, h16 = h16_next; const bool adapt = adapt_next; const float xsel = xsel_next;
                    if (i + 1 < K) { // RNG draws stay in stream order: [grid, level, cell | cell, select] per lookup
                        adapt_next = xorshift(p.rng) < P.mc_samples_adaptive_prob;
                        if (adapt_next) mc_adaptive_buffer_index_at(P, base_level, p.rng, lp, p.cur.normal, bi_next, h16_next);
                        else mc_static_buffer_index(P, p.rng, lp, bi_next, h16_next);
                        xsel_next = xorshift(p.rng);
                        st_next = mc_load(F.mc, bi_next);
                    }
                    if (i < 0) continue;
                    if (COUNT) ctr.mc_reads++;
                    mc_finalize_load(U, st, h16, !adapt, p.cur.pos, p.cur.normal);
                    p.score_sum += st.sum_w;
                    f3 d = mc_state_dir(st, p.cur.pos); float kk = mc_state_kappa(P, st, p.cur.pos);
                    float* li = lobes + (6 * i) * 64;
                    const float nrm = vmf_norm(kk);
                    if (xsel < st.sum_w / p.score_sum) { // NaN compares false; selected lobe moves to slot 0 (mcpg.comp:99-105)
                        sel = st; p.mc_index = bi;
                        if (i > 0) { li[0] = lobes[0]; li[64] = lobes[64]; li[128] = lobes[128]; li[192] = lobes[192]; li[256] = lobes[256]; li[320] = lobes[320]; }
                        lobes[0] = st.sum_w; lobes[64] = d.x; lobes[128] = d.y; lobes[192] = d.z; lobes[256] = kk; lobes[320] = nrm;
                    } else { li[0] = st.sum_w; li[64] = d.x; li[128] = d.y; li[192] = d.z; li[256] = kk; li[320] = nrm; }
                }
                PLAP(ctr, 5);
                if (p.score_sum == 0.0f || xorshift(p.rng) < P.surf_bsdf_p) { // :113-117
                    float x0 = xorshift(p.rng), x1 = xorshift(p.rng), x2 = xorshift(p.rng);
                    wo = bsdf_sample(p.cur.wi, p.cur.normal, alpha, x0, x1, x2);
                    sel = mc_state_new(p.rng);
                    p.mc_index = MQ_NIL;
                } else {
                    float x0 = xorshift(p.rng), x1 = xorshift(p.rng);
                    wo = vmf_sample(F3(lobes[64], lobes[128], lobes[192]), lobes[256], x0, x1);
                }
                PLAP(ctr, 6);
                p.wodotn = dot(wo, p.cur.normal);
                if (p.wodotn <= 1e-3f || dot(wo, decode_normal(p.cur.enc_geonormal)) <= 1e-3f) rejected = true;
                else {
                    float g = 0.0f;
                    if (p.score_sum > 0.0f) {
#pragma nounroll
                        for (int i = 0; i < K; i++) {
                            const float* li = lobes + (6 * i) * 64;
                            g += li[0] * vmf_pdf_normed(wo, F3(li[64], li[128], li[192]), li[256], li[320]);
                        }
                        g /= p.score_sum;
                    }
                    p.wo_p = (p.score_sum > 0.0f ? P.surf_bsdf_p : 1.0f) * bsdf_pdf(p.cur.wi, wo, p.cur.normal, alpha) + (1.0f - P.surf_bsdf_p) * g;
                    p.mc_id = sel.id; p.mc_sum_w = sel.sum_w;
                    // second half of mc_light_missing (mc.glsl:34-38), evaluated now so the state need not be kept
                    p.lm_dir_ok = false;
                    if (p.mc_index != MQ_NIL) p.lm_dir_ok = !(dot(wo, mc_state_dir(sel, p.cur.pos)) < 0.9f + 0.1f * mc_state_mean_cos(P, sel, p.cur.pos));
                }
            }
            PLAP(ctr, 7);
            if (rejected) sample_done = true; // `break` at mcpg.comp:63 / :125
            else {
                p.bsdf = bsdf_times_wodotn(p.cur.wi, wo, p.cur.normal, alpha, 0.02f); // :153 (pre-trace data only)
                p.wo = wo;
                PLAP(ctr, 8);
                return true;
            }
        }
        if (sample_done) { // mcpg.comp:193-198
            PLAP(ctr, 32);
            sample_done = false;
            f3 contrib = p.fval * (1.0f / p.pp);
            if (mfinite(contrib.x) && mfinite(contrib.y) && mfinite(contrib.z)) { p.irr = p.irr + contrib; float l = luminance(contrib); p.m2 += l * l; }
            p.smp++;
            PLAP(ctr, 33);
            if (p.smp < P.spp) {
                load_chit(F.hits + 10 * pidx, p.cur);
                p.thr = F3(1, 1, 1); p.fval = F3(0, 0, 0); p.pp = 1.0f; p.seg = 1;
                need_dir = true;
            } else { // :205-210
                float inv = 1.0f / (float)P.spp;
                float4 o4 = make_float4(p.irr.x * inv, p.irr.y * inv, p.irr.z * inv, p.m2 * inv);
                *(float4*)(F.irradiance + 4 * pidx) = o4;
                *(float4*)(F.tiles_out + 4 * (size_t)slot) = o4;
                if (P.debug_output_connected) F.debug_rng[pidx] = p.rng;
                PLAP(ctr, 34);
                return false;
            }
        }
    }
    return false;
}

// ---- sharded queues ------------------------------------------------------------------------------
// A queue has MQ_SHARDS tails; shard s owns every 16th BLOCK of 64 positions: entry k of shard s sits
// at ((k / 64) * 16 + s) * 64 + k % 64, so the (up to 64) entries one wave appends stay adjacent and
// a consumer wave reading 64 consecutive positions sees the rays of neighbouring pixels.  A wave
// appends to the shard of its wave id with ONE atomic (ballot + prefix popcount).  Consumers walk
// positions [0, n_eff) and skip the holes behind shorter shards.
struct QView { uint32_t cnt; uint32_t n_eff; }; // cnt: tail of shard (lane & 15) in every lane; n_eff = 1024 * ceil(max tail / 64)
MQ_DEV QView queue_view(const uint32_t* tails, uint32_t stride = MQ_SHARD_STRIDE) {
    const int lane = threadIdx.x & 63;
    QView v;
    v.cnt = tails[(lane & (MQ_SHARDS - 1)) * stride];
    uint32_t m = v.cnt;
#pragma unroll
    for (int off = 8; off > 0; off >>= 1) { uint32_t o = (uint32_t)__shfl_xor((int)m, off, 64); m = o > m ? o : m; }
    v.n_eff = ((m + 63u) >> 6) * (64u * MQ_SHARDS);
    return v;
}
MQ_DEV bool queue_valid(const QView& v, uint32_t q) {
    const uint32_t shard = (q >> 6) & (MQ_SHARDS - 1), k = ((q >> 10) << 6) | (q & 63u);
    return k < (uint32_t)__shfl((int)v.cnt, (int)shard, 64);
}
MQ_DEV uint32_t shard_append(uint32_t* tails, bool push) { // returns the interleaved position (valid where push)
    unsigned long long m = __ballot(push);
    uint32_t pos = 0;
    if (m) {
        const int lane = threadIdx.x & 63;
        const uint32_t shard = (blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)) & (MQ_SHARDS - 1);
        int leader = __ffsll((long long)m) - 1;
        uint32_t base = 0;
        if (lane == leader) base = atomicAdd(&tails[shard * MQ_SHARD_STRIDE], (uint32_t)__popcll(m));
        base = __shfl(base, leader, 64);
        pos = shard_pos(shard, base + (uint32_t)__popcll(m & ((1ull << lane) - 1ull)));
    }
    return pos;
}
MQ_DEV uint32_t queue_append(const MqFrame& F, int round, bool& push) {
    uint32_t q = shard_append(F.qctrl + MQ_QTAILS(round), push);
    if (push && q >= F.ray_cap) { atomicOr(&F.ctrl[0], 1u); push = false; } // cannot happen with the 2x margin; flagged, never silent
    return q;
}
// Rays of round r live in ray buffer r & 1 (origins, then directions: field-major like the path records).  Two
// buffers because a shading kernel reads the direction of the ray that just returned (position q of ITS round's
// queue) while other waves of the same launch already write rays of the next round -- whose queue positions start
// at 0 again: with one buffer a wave could read a direction another wave had just replaced.
MQ_DEV float4* ray_buffer(const MqFrame& F, int round) { return F.rays + (size_t)(round & 1) * 2u * F.ray_cap; }

MQ_DEV void emit_ray(const MqFrame& F, int round, uint32_t q, uint32_t slot, const Path& p) {
    f3 ro = p.cur.pos - p.cur.wi * 1e-3f; // mcpg.comp:144
    float4* rays = ray_buffer(F, round);
    rays[q] = make_float4(ro.x, ro.y, ro.z, 0.0f);
    rays[(size_t)F.ray_cap + q] = make_float4(p.wo.x, p.wo.y, p.wo.z, 0.0f);
    F.queue_slots[round & 1][q] = slot;
    store_path(F.paths + slot, F.n_slots, p);
}

// ---- camera rays: FRUSTUM PACKET traversal, one 8x8-pixel tile per wave -----------------------------------------------
// The 64 camera rays of a tile share their origin and span a narrow pyramid, so the wave walks the tree ONCE for all of
// them: one shared stack (no per-lane stacks), every node and triangle fetched once for the wave with coalesced /
// broadcast loads (no divergent gathers), and a child box is tested against the tile's pyramid instead of against 64 rays -- lane 8 r + c
// tests child c against side plane r (r = 0..3) or against the distance of the farthest closest hit so far (r = 4);
// a ballot combines them.  That is ~85 vector instructions per node visit instead of the ~210 of the per-ray slab test
// of eight children.  Only triangles are tested per ray (all 64 lanes, the triangle in SGPRs), with exactly the
// operations, acceptance rule and tie break of trav_tri, so the closest hits are bit-identical to the per-ray
// traversal: the closest hit of a ray is the minimum over ALL candidate triangles by (t, key), and culling is
// conservative -- a box is skipped only if it lies outside the pyramid of pixel-centre rays widened by half a pixel
// (relative slack 1e-5 on the plane test; the boxes themselves are padded, mq_bvh.cpp) or farther than every lane's hit.
// The host launches this kernel only for trees whose depth fits the shared stack (mq_scene_commit records the depth of
// both trees; deeper ones -- not seen so far -- take mq_primary_trace_lanes_kernel, the per-lane traversal).
#define MQ_PKT_STACK 48
// wave-wide OR / max by DPP (quad swaps, half-row and row mirrors, row broadcasts); the result is in lane 63
MQ_DEV uint32_t wave_or(uint32_t v) {
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true);  // quad_perm [1,0,3,2]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true);  // quad_perm [2,3,0,1]
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); // row_half_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); // row_mirror
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); // row_bcast:15 into rows 1 and 3
    v |= (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); // row_bcast:31 into rows 2 and 3
    return (uint32_t)__builtin_amdgcn_readlane((int)v, 63);
}
MQ_DEV float wave_max_nonneg(float x) { // x >= 0: the float order is the order of the bit patterns
    uint32_t v = __float_as_uint(x), o;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0xB1, 0xf, 0xf, true); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x4E, 0xf, 0xf, true); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x141, 0xf, 0xf, true); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x140, 0xf, 0xf, true); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x142, 0xa, 0xf, true); v = o > v ? o : v;
    o = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x143, 0xc, 0xf, true); v = o > v ? o : v;
    return __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)v, 63));
}
MQ_DEV uint32_t rfl(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }

// One node of the packet walk: which children does the tile's pyramid reach?  Every lane fetches what IT needs of the
// 80-byte node with ordinary vector loads -- the header (same address in all lanes: one request) and the seven bytes
// of its child; the whole node is two cache lines, so the wave's loads coalesce to two requests each.  (Scalar loads of
// the node were measured slower, 0.44 against 0.33 ms for the per-lane kernel: the scalar cache sustains only about a
// dozen misses in flight and every tile touches ~70 fresh lines.)  Returns the hit bits in box4's layout (internal
// child -> bit 24 + (slot ^ octant), leaf -> the bits of its triangles), the same in every lane.
MQ_DEV uint32_t pkt_node(const MqNode* nd, f3 o, f3 pn, int role, uint32_t child, uint32_t octu, float tmax2, uint32_t& child_base, uint32_t& tri_base, uint32_t& imask) {
    const uint4 h = *(const uint4*)nd;                     // origin, exponents | imask << 24
    const uint2 bases = *((const uint2*)nd + 2);           // child_base, tri_base
    const uint8_t* q = (const uint8_t*)nd + 24 + child;    // meta, then the six plane arrays, 8 bytes apart
    const uint32_t meta = q[0];
    const float qlx = (float)q[8], qly = (float)q[16], qlz = (float)q[24], qhx = (float)q[32], qhy = (float)q[40], qhz = (float)q[48];
    child_base = rfl(bases.x); tri_base = rfl(bases.y); imask = rfl(h.w) >> 24;
    // quantised box -> world box with the builder's own float expressions (mq_bvh.cpp:225-226), relative to the origin
    const float ex = __uint_as_float((h.w & 0xffu) << 23), ey = __uint_as_float(((h.w >> 8) & 0xffu) << 23), ez = __uint_as_float(((h.w >> 16) & 0xffu) << 23);
    const float bx = __uint_as_float(h.x), by = __uint_as_float(h.y), bz = __uint_as_float(h.z);
    const float ax = (bx + qlx * ex) - o.x, ay = (by + qly * ey) - o.y, az = (bz + qlz * ez) - o.z; // lo - origin
    const float cx = (bx + qhx * ex) - o.x, cy = (by + qhy * ey) - o.y, cz = (bz + qhz * ez) - o.z; // hi - origin
    // side planes (roles 0..3): the box corner farthest along the inward normal must not be behind the plane;
    // role 4: squared distance from the origin to the box against the farthest closest hit of the tile
    const float pv = (fmaxf(pn.x * ax, pn.x * cx) + fmaxf(pn.y * ay, pn.y * cy)) + fmaxf(pn.z * az, pn.z * cz);
    const float mag = (fabsf(pn.x) * fmaxf(fabsf(ax), fabsf(cx)) + fabsf(pn.y) * fmaxf(fabsf(ay), fabsf(cy))) + fabsf(pn.z) * fmaxf(fabsf(az), fabsf(cz));
    const float dx = fmaxf(fmaxf(ax, -cx), 0.0f), dy = fmaxf(fmaxf(ay, -cy), 0.0f), dz = fmaxf(fmaxf(az, -cz), 0.0f);
    const bool pass = role < 4 ? !(pv < -1e-5f * mag) : (role > 4 || !(((dx * dx + dy * dy) + dz * dz) * 0.9999f > tmax2));
    const unsigned long long pm = __ballot(pass);
    const uint32_t cm = (uint32_t)(pm & (pm >> 8) & (pm >> 16) & (pm >> 24) & (pm >> 32)) & 0xffu; // children inside all four planes and near enough
    const bool inner = (meta & 0x18u) == 0x18u;
    const uint32_t bidx = inner ? 24u + ((meta & 7u) ^ octu) : (meta & 31u);
    const uint32_t contrib = (role == 0 && ((cm >> child) & 1u)) ? (meta >> 5) << bidx : 0u;
    return wave_or(contrib);
}

__global__ __launch_bounds__(MQ_BLOCK, 5) void mq_primary_trace_kernel(MqSceneDev sc, MqFrame F, float fov_tan_alpha_half) {
    __shared__ uint2 s_pstack[MQ_WAVES][MQ_PKT_STACK];   // the wave's shared stack of node groups
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* const pstk = &s_pstack[wave][0];
    const mq_uniform& U = F.u;
    const float Wf = (float)F.W, Hf = (float)F.H;
    const f3 up = F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), fw = F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]);
    const f3 o = cam_pos(U);
    const uint32_t n_waves = gridDim.x * MQ_WAVES;
    const uint32_t first_tile = F.slot_begin >> 6, end_tile = F.slot_end >> 6;
    const int role = lane >> 3;                 // 0..3: side plane, 4: distance, 5..7: nothing to test
    const uint32_t child = (uint32_t)lane & 7u; // the child slot this lane tests
    for (uint32_t tile = first_tile + blockIdx.x * MQ_WAVES + (uint32_t)wave; tile < end_tile; tile += n_waves) {
        const uint32_t my = (tile << 6) | (uint32_t)lane;
        const uint32_t gtile = MQ_GTILE(F, tile);
        const uint32_t tx = gtile % F.tiles_x, ty = gtile / F.tiles_x;
        const uint32_t px = tx * 8u + ((uint32_t)lane & 7u), py = ty * 8u + ((uint32_t)lane >> 3);
        const bool valid = px < F.W && py < F.H;
        const f3 d = camera_ray_dir((float)px, (float)py, Wf, Hf, up, fw, fov_tan_alpha_half);
        // the tile's pyramid: pixel-centre rays widened by half a pixel and a bit; inward normals of its four sides
        const float x0 = (float)(tx * 8u) - 0.52f, x1 = (float)(tx * 8u) + 7.52f, y0 = (float)(ty * 8u) - 0.52f, y1 = (float)(ty * 8u) + 7.52f;
        const f3 c00 = camera_ray_dir(x0, y0, Wf, Hf, up, fw, fov_tan_alpha_half), c10 = camera_ray_dir(x1, y0, Wf, Hf, up, fw, fov_tan_alpha_half);
        const f3 c11 = camera_ray_dir(x1, y1, Wf, Hf, up, fw, fov_tan_alpha_half), c01 = camera_ray_dir(x0, y1, Wf, Hf, up, fw, fov_tan_alpha_half);
        const f3 cdir = (c00 + c11) + (c10 + c01);
        f3 pn = role == 0 ? cross(c00, c10) : (role == 1 ? cross(c10, c11) : (role == 2 ? cross(c11, c01) : cross(c01, c00)));
        if (dot(pn, cdir) < 0.0f) pn = -pn;
        const uint32_t octu = rfl((cdir.x < 0.0f ? 0u : 1u) | (cdir.y < 0.0f ? 0u : 2u) | (cdir.z < 0.0f ? 0u : 4u)); // visiting order only
        RayHit hit; hit.tri = MQ_NIL; hit.t = __uint_as_float(0x7f800000u); hit.u = 0.0f; hit.v = 0.0f;
        uint32_t best_key = MQ_NIL;
        if (sc.n_nodes != 0) {
            float tmax2 = trav_limit(MQ_T_MAX); tmax2 *= tmax2; // (farthest closest hit of the tile)^2, conservative
            uint32_t sp = 0;
            if (sc.dyn_root != MQ_NIL) { pstk[0] = make_uint2(sc.dyn_root, 0x80000000u); sp = 1; } // the per-frame tree: visited last (trav_defer)
            uint32_t Gx = 0u, Gy = 0x80000000u; // the group on top (in registers): first internal child, pending-children bits << 24 | imask; starts at the root
            for (;;) {
                while (!(Gy > 0x00ffffffu) && sp != 0) { sp--; const uint2 g = pstk[sp]; Gx = rfl(g.x); Gy = rfl(g.y); }
                if (!(Gy > 0x00ffffffu)) break;
                const uint32_t bit = 31u - (uint32_t)__clz((int)Gy);
                Gy &= ~(1u << bit);
                const uint32_t slot = (bit - 24u) ^ octu;
                const uint32_t node = Gx + (uint32_t)__popc(Gy & 0xffu & ((1u << slot) - 1u));
                uint32_t cbase, tbase, imask;
                const uint32_t hm = pkt_node(sc.nodes + node, o, pn, role, child, octu, tmax2, cbase, tbase, imask);
                if (Gy > 0x00ffffffu) { // the rest of the group waits on the stack
                    if (sp >= MQ_PKT_STACK) { atomicOr(&F.ctrl[0], 4u); break; } // cannot happen (the host checked the depth); flagged, never silent
                    if (lane == 0) pstk[sp] = make_uint2(Gx, Gy);
                    sp++;
                }
                Gx = cbase; Gy = (hm & 0xff000000u) | imask;
                uint32_t tmask = hm & 0x00ffffffu;
                bool any_new = false;
                while (tmask) { // every lane tests the leaf record's triangles (one broadcast fetch) with its own ray: trav_tri's operations, acceptance rule and tie break
                    const uint32_t k = (uint32_t)__ffs((int)tmask) - 1u;
                    tmask &= tmask - 1u;
                    const LeafTris L = load_leaf(sc, tbase + k);
                    float tt = 0.0f, uu = 0.0f, vv = 0.0f;
                    bool accept = tri_isect(o, d, L.a0, L.a1, L.a2, tt, uu, vv);
                    accept = accept && (tt < MQ_T_MAX) && (tt < hit.t || (tt == hit.t && L.key0 < best_key));
                    if (accept && (L.sel & 0x10000u)) accept = anyhit_confirm(sc, L.tri0, uu, vv);
                    if (accept) { hit.t = tt; hit.u = uu; hit.v = vv; hit.tri = L.tri0; best_key = L.key0; }
                    any_new = any_new || accept;
                    if (L.sel & MQ_LEAF_HAS_B) { // (wave-uniform: every lane holds the same record)
                        accept = tri_isect(o, d, L.b0, L.b1, L.b2, tt, uu, vv);
                        accept = accept && (tt < MQ_T_MAX) && (tt < hit.t || (tt == hit.t && L.key1 < best_key));
                        if (accept && (L.sel & 0x20000u)) accept = anyhit_confirm(sc, L.tri0 + 1u, uu, vv);
                        if (accept) { hit.t = tt; hit.u = uu; hit.v = vv; hit.tri = L.tri0 + 1u; best_key = L.key1; }
                        any_new = any_new || accept;
                    }
                }
                if (__ballot(any_new) != 0ull) { // the farthest closest hit of the tile's (valid) rays bounds what is still worth visiting
                    const float m = wave_max_nonneg(valid ? fminf(hit.t, MQ_T_MAX) : 0.0f);
                    const float lim = trav_limit(m);
                    tmax2 = lim * lim;
                }
            }
        }
        if (valid) F.prim_hits[my] = make_uint4(hit.tri, __float_as_uint(hit.t), __float_as_uint(hit.u), __float_as_uint(hit.v));
    }
}

// ---- camera rays, per-lane traversal: for trees too deep for the packet kernel's shared stack ------------------------
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_TRACE) void mq_primary_trace_lanes_kernel(MqSceneDev sc, MqFrame F, float fov_tan_alpha_half) {
    __shared__ uint2 s_stack[MQ_WAVES][MQ_STACK_LDS][64];
    const int lane = threadIdx.x & 63;
    uint2* stk = &s_stack[threadIdx.x >> 6][0][lane];
    const mq_uniform& U = F.u;
    const uint32_t total = F.slot_end; // this pipeline's pixel slots: [slot_begin, slot_end)
    const float Wf = (float)F.W, Hf = (float)F.H;
    Ctr ctr = {};
    for (uint32_t my = F.slot_begin + blockIdx.x * MQ_BLOCK + threadIdx.x; my < total; my += gridDim.x * MQ_BLOCK) { // (the host launches one wave per tile: one trip)
        unsigned long long* spill = F.cam_spill + (size_t)my * MQ_SPILL_ENTRIES; // per pixel slot: whatever else runs beside this launch has its own area
        const uint32_t gtile = MQ_GTILE(F, my >> 6), within = my & 63u;
        const uint32_t px = (gtile % F.tiles_x) * 8u + (within & 7u), py = (gtile / F.tiles_x) * 8u + (within >> 3);
        if (px >= F.W || py >= F.H) continue;
        const f3 up = F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), fw = F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]);
        RayHit rhit;
        traverse<false>(sc, cam_pos(U), camera_ray_dir((float)px, (float)py, Wf, Hf, up, fw, fov_tan_alpha_half), rhit, stk, spill, ctr);
        F.prim_hits[my] = make_uint4(rhit.tri, __float_as_uint(rhit.t), __float_as_uint(rhit.u), __float_as_uint(rhit.v));
    }
}

// ---- first hit: gbuffer.comp:75-131 + start of mcpg.comp:39-57 --------------------------------
template <bool GUIDED, bool COUNT>
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_primary_kernel(MqSceneDev sc, MqParams P, MqFrame F) {
    // one LDS region per wave, used first as traversal stack, then as lobe storage of the direction choice.
    // (Tracing the primary rays in mq_trace_queue_kernel instead was measured: 0.36 ms there against
    // 0.28 ms here -- a wave of this kernel is one 8x8 tile whose rays stay coherent to the end, while
    // the dynamic fetch of the queue kernel mixes tiles -- plus one more launch per frame.)
    extern __shared__ uint2 s_dyn[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint2* stk = s_dyn + (size_t)wave * F.lds_rows2 * 64 + lane;
    float* lobes = GUIDED ? (float*)(s_dyn + (size_t)wave * F.lds_rows2 * 64) + lane : nullptr;
    unsigned long long* spill = F.stack_spill + (size_t)(blockIdx.x * blockDim.x + threadIdx.x) * MQ_SPILL_ENTRIES; // blockDim.x: 256, fewer when many Markov-chain samples need more LDS per wave (shade_block)
    const uint32_t total = F.slot_end; // this pipeline's pixel slots: [slot_begin, slot_end)
    const mq_uniform& U = F.u;
    const float Wf = (float)F.W, Hf = (float)F.H;
    const f3 gb_sun = P.gbuffer_hide_sun ? F3(0.0f, 0.0f, 0.0f) : F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]);
    Ctr ctr = {};
    if (blockIdx.x == 0 && threadIdx.x < MQ_SHARDS && F.slot_begin == 0u && !F.gbuffer_only) F.active_ctrl[threadIdx.x * MQ_SHARD_STRIDE] = 0u; // the link pass of this frame lists from 0 (the last frame's apply pass is done)
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t rounds = (total - F.slot_begin + stride - 1) / stride;
    PSTART(ctr);
    for (uint32_t it = 0; it < rounds; it++) {
        const uint32_t my = F.slot_begin + it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool cont = false;
        Path p = {};
        PLAP(ctr, 0);
        if (my < total) {
            uint32_t ltile = my >> 6, within = my & 63u;
            uint32_t gtile = MQ_GTILE(F, ltile);
            uint32_t tx = gtile % F.tiles_x, ty = gtile / F.tiles_x;
            p.px = tx * 8u + (within & 7u); p.py = ty * 8u + (within >> 3);
            if (p.px < F.W && p.py < F.H) {
                if (COUNT) ctr.pixels++;
                const size_t pidx = (size_t)p.py * F.W + p.px;
                p.rng = pcg4d16(p.px, p.py, U.frame, P.seed); // mcpg.comp:40
                const f3 up = F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), fw = F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]);
                f3 ro = cam_pos(U);
                f3 rd = camera_ray_dir((float)p.px, (float)p.py, Wf, Hf, up, fw, P.fov_tan_alpha_half);
                RayHit rhit;
                if constexpr (!COUNT) { const uint4 hq = F.prim_hits[my]; rhit.tri = hq.x; rhit.t = __uint_as_float(hq.y); rhit.u = __uint_as_float(hq.z); rhit.v = __uint_as_float(hq.w); } // traced by mq_primary_trace_kernel
                else traverse<COUNT>(sc, ro, rd, rhit, stk, spill, ctr); // the counting instantiation traces inline: its counters price the camera rays
                PLAP(ctr, 1);
                Hit h; h.pos = ro; h.wi = rd; h.prev_pos = ro; h.normal = F3(0, 0, 1); h.enc_geonormal = 0; h.albedo = F3(0, 0, 0); h.roughness = 0.0f;
                f3 incident = F3(0, 0, 0), cthr = F3(1, 1, 1);
                const f3 r_x = camera_ray_dir((float)p.px + 1.0f, (float)p.py, Wf, Hf, up, fw, P.fov_tan_alpha_half); // gbuffer.comp:92-93
                const f3 r_y = camera_ray_dir((float)p.px, (float)p.py + 1.0f, Wf, Hf, up, fw, P.fov_tan_alpha_half);
                shade_hit<true>(sc, P, U, rhit, cthr, incident, h, gb_sun, r_x, r_y);
                PLAP(ctr, 2);
                *(uint2*)(F.gb_irr + 4 * pidx) = make_uint2((uint32_t)f2h(incident.x) | ((uint32_t)f2h(incident.y) << 16), (uint32_t)f2h(incident.z) | (0x3c00u << 16));
                float keep = (incident.x >= 1e-5f || incident.y >= 1e-5f || incident.z >= 1e-5f) ? 0.0f : 1.0f; // gbuffer.comp:107
                h.albedo = rh3(rh3(h.albedo * keep) * cthr);
                *(uint2*)(F.gb_albedo + 4 * pidx) = make_uint2((uint32_t)f2h(h.albedo.x) | ((uint32_t)f2h(h.albedo.y) << 16), (uint32_t)f2h(h.albedo.z) | (0x3c00u << 16));
                { // gbuffer.comp:111-115
                    f3 old_dir = h.prev_pos - F3(U.prev_cam_x[0], U.prev_cam_x[1], U.prev_cam_x[2]);
                    float opx, opy;
                    camera_pixel(old_dir, Wf, Hf, F3(U.prev_cam_u[0], U.prev_cam_u[1], U.prev_cam_u[2]), F3(U.prev_cam_w[0], U.prev_cam_w[1], U.prev_cam_w[2]), P.fov_tan_alpha_half, opx, opy);
                    *(uint32_t*)(F.gb_mv + 2 * pidx) = (uint32_t)f2h(opx - (float)p.px) | ((uint32_t)f2h(opy - (float)p.py) << 16);
                }
                __attribute__((aligned(8))) uint32_t rec[10];
                store_chit(rec, h);
                { uint2* d2 = (uint2*)(F.hits + 10 * pidx); const uint2* s2 = (const uint2*)rec; d2[0] = s2[0]; d2[1] = s2[1]; d2[2] = s2[2]; d2[3] = s2[3]; d2[4] = s2[4]; }
                { // gbuffer.comp:123-130
                    f3 gn = decode_normal(h.enc_geonormal);
                    f3 cp = cam_pos(U);
                    float lz = length(cp - h.pos);
                    float num = dot(gn, h.pos - cp);
                    uint32_t g0 = f2h(num / dot(gn, r_x - h.wi) - lz), g1 = f2h(num / dot(gn, r_y - h.wi) - lz);
                    float vz = length(F3(U.prev_cam_x[0], U.prev_cam_x[1], U.prev_cam_x[2]) - h.prev_pos) - lz;
                    *(uint4*)(F.gbuffer + 4 * pidx) = make_uint4(encode_normal(h.normal), __float_as_uint(lz), g0 | (g1 << 16), __float_as_uint(vz));
                }
                PLAP(ctr, 3);
                // mcpg.comp:44: pixels whose first hit carries no albedo get zero irradiance
                if (F.gbuffer_only) {} // a row band's g-buffer for the ReSTIR node / the post chain of a rank: the MCPG outputs are not this launch's
                else if ((h.albedo.x >= 1e-7f || h.albedo.y >= 1e-7f || h.albedo.z >= 1e-7f) && P.spp > 0 && P.max_path_length > 1) {
                    load_chit(rec, p.cur); // the surface pass starts from the COMPRESSED first hit (mcpg.comp:46-47)
                    p.thr = F3(1, 1, 1); p.fval = F3(0, 0, 0); p.pp = 1.0f; p.seg = 1; p.smp = 0; p.irr = F3(0, 0, 0); p.m2 = 0.0f;
                    cont = advance_path<GUIDED, COUNT>(P, F, p, my, true, false, lobes, ctr);
                } else {
                    float4 o4 = make_float4(0.0f, 0.0f, 0.0f, 0.0f);
                    *(float4*)(F.irradiance + 4 * pidx) = o4;
                    *(float4*)(F.tiles_out + 4 * (size_t)my) = o4;
                    if (P.debug_output_connected) F.debug_rng[pidx] = p.rng;
                }
            }
        }
        PLAP(ctr, 9);
        uint32_t q = queue_append(F, 0, cont);
        if (cont) emit_ray(F, 0, q, my, p);
        PLAP(ctr, 10);
    }
    if (COUNT) flush_counters(F.counters, ctr);
    PFLUSH(F.counters, ctr);
}

// ---- closest hit for every queued ray ------------------------------------------------------------
// Persistent waves with dynamic ray fetch: a lane whose ray is finished takes the next ray of its
// wave's pool instead of idling until the longest traversal of the wave ends; pools are refilled
// MQ_TRACE_BLOCKS blocks of its home shard at a time (one atomic on one of 16 per-shard heads), moving on
// to the other shards when the home shard is drained, so no single word sees more than ~n/2048 atomics.
#ifndef MQ_TRACE_BLOCKS
#define MQ_TRACE_BLOCKS 1u
#endif
// most 64-entry blocks fetched per refill.  One block: with 256-ray refills the ~7400 chunks of a 1080p
// round spread over 6144 resident waves as "one or two each", i.e. a 2x makespan imbalance (measured:
// 44 of 64 lanes busy on average, trace rounds 0.474 + 0.225 ms against 0.448 + 0.211 ms with 64-ray refills).
#ifndef MQ_REFILL_MIN
#define MQ_REFILL_MIN 8u
#endif
#ifndef MQ_TRI_VOTE
#define MQ_TRI_VOTE 24u
#endif
#ifndef MQ_SHARE_MIN_IDLE
#define MQ_SHARE_MIN_IDLE 8u // a hand-over round costs ~60 instructions of the whole wave: how many idle lanes make it worth it
#endif
#ifndef MQ_SHARE
#define MQ_SHARE 1 // idle lanes adopt subtrees of busy lanes once the queue is exhausted (see the kernel)
#endif
#ifndef MQ_OCC_TRACEQ
#define MQ_OCC_TRACEQ MQ_OCC_TRACE
#endif
template <bool COUNT>
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_TRACEQ) void mq_trace_queue_kernel(MqSceneDev sc, MqFrame F, int round) {
    __shared__ uint2 s_stack[MQ_WAVES][MQ_STACK_LDS][64];
    const int lane = threadIdx.x & 63;
#ifdef MQ_PROF
    __shared__ uint32_t s_hist[MQ_WAVES][64]; // rays by loop iterations, bins of 8
    s_hist[threadIdx.x >> 6][lane] = 0u;
#endif
    uint2* stk = &s_stack[threadIdx.x >> 6][0][lane];
    const uint32_t gid = blockIdx.x * MQ_BLOCK + threadIdx.x;
    unsigned long long* spill = F.stack_spill + (size_t)gid * MQ_SPILL_ENTRIES;
    const uint32_t* tails = F.qctrl + MQ_QTAILS(round);
    uint32_t* heads = F.qctrl + MQ_QHEADS(round);
    // the control words of the other round parity: read last by the shading launch before this one, written next by the one behind it
    if (blockIdx.x == 0 && threadIdx.x < MQ_SHARDS) { F.qctrl[MQ_QTAILS(round + 1) + threadIdx.x * MQ_SHARD_STRIDE] = 0u; F.qctrl[MQ_QHEADS(round + 1) + threadIdx.x * MQ_SHARD_STRIDE] = 0u; }
    Ctr ctr = {};
    const uint32_t wave_id = blockIdx.x * MQ_WAVES + (threadIdx.x >> 6);
    const uint32_t n_eff = queue_view(tails).n_eff;
    if (sc.n_nodes == 0) { // empty scene: every ray misses
        const uint32_t lim = n_eff < F.ray_cap ? n_eff : F.ray_cap;
        for (uint32_t q = gid; q < lim; q += gridDim.x * MQ_BLOCK) F.ray_hits[q] = make_uint4(MQ_NIL, 0x7f800000u, 0u, 0u);
        return;
    }
    // The wave's pool: `pool_len` entries of shard `pool_s`, starting at entry 64 * pool_j of that shard
    // (a run of MQ_TRACE_BLOCKS 64-entry blocks); `pool_i` entries are already handed out.  All wave-uniform.
    uint32_t pool_s = wave_id & (MQ_SHARDS - 1), pool_j = 0, pool_i = 0, pool_len = 0;
    // No more waves than 64-ray blocks in the queue (a small queue: late rounds, a rank of a partitioned frame): a wave
    // that starts with half its lanes empty finishes no earlier and its gathers queue up with everybody else's.
    const uint32_t n_waves_all = gridDim.x * MQ_WAVES, n_waves = n_eff / 64u < n_waves_all ? (n_eff / 64u > 0u ? n_eff / 64u : 1u) : n_waves_all;
    if (wave_id >= n_waves) return;
    const uint32_t per_wave = n_eff / n_waves;
    const uint32_t nblk = per_wave >= 64u * MQ_TRACE_BLOCKS ? MQ_TRACE_BLOCKS : (per_wave >= 128u && MQ_TRACE_BLOCKS >= 2u ? 2u : 1u);
    bool exhausted = false;
    uint32_t q = MQ_NIL; // queue position of the lane's ray; MQ_NIL = the lane is idle
#define busy (q != MQ_NIL)
#ifdef MQ_PROF
    uint32_t ray_iters = 0;
#endif
    Trav t;
    trav_init(t, F3(0, 0, 0), F3(0, 0, 1));
    // The counting instantiation runs without work sharing: its node / triangle counters price the algorithm's visits,
    // not the extra ones of helper lanes (which cull with an older limit).
    constexpr bool SHARE = MQ_SHARE != 0 && !COUNT;
    // Work sharing in the drain (queue exhausted, lanes running dry): an idle lane adopts the OLDEST pending stack
    // entry of a busy lane -- the same ray, another subtree -- traverses it with the ray's current closest hit as
    // its limit and hands its own closest hit back to the ray's owner lane when done.  The closest hit of a ray is
    // the minimum over its subtrees by (t, key), so the result is the one the owner alone would have found.
    // owner: lane that holds the ray and writes its result (-1: this lane is an owner); fin: this lane's subtree is
    // finished; s_pend[owner lane]: helper lanes of that owner still running.
    __shared__ uint32_t s_pend[MQ_WAVES][64];
    uint32_t* const pend = &s_pend[threadIdx.x >> 6][0];
    pend[lane] = 0u;
    int owner = -1;
    bool fin = false;
    PSTART(ctr);
#ifdef MQ_PROF
    const uint32_t prof_t0 = ctr.pt;
#endif
    for (;;) {
        unsigned long long idle = __ballot(!busy);
        PLAP(ctr, 24);
        // Refill in batches: fetching rays (a round trip to memory, three divisions per ray) costs the whole wave the same
        // for one idle lane as for sixteen, and a few lanes finish in every step.
        if ((uint32_t)__popcll(idle) >= (exhausted ? 1u : MQ_REFILL_MIN)) {
            while (pool_i == pool_len && !exhausted) { // refill: next run of blocks of the current shard
                uint32_t cnt = 0, head = 0;
                if (lane == 0) { cnt = tails[pool_s * MQ_SHARD_STRIDE]; head = __hip_atomic_load(&heads[pool_s * MQ_SHARD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
                cnt = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt); head = (uint32_t)__builtin_amdgcn_readfirstlane((int)head);
                bool got = false;
                if (head * 64u < cnt) { // look before bumping: a failed atomic on a shared line is the expensive case
                    uint32_t j = 0;
                    if (lane == 0) j = atomicAdd(&heads[pool_s * MQ_SHARD_STRIDE], nblk);
                    j = (uint32_t)__builtin_amdgcn_readfirstlane((int)j);
                    if (j * 64u < cnt) {
                        pool_j = j; pool_i = 0;
                        const uint32_t rest = cnt - j * 64u;
                        pool_len = rest < 64u * nblk ? rest : 64u * nblk;
                        got = true;
                    }
                }
                if (!got) { // this shard is drained: probe all 16 shards in one round trip (lanes 0..15) and move to the next one with rays
                    const uint32_t sh = (uint32_t)lane & (MQ_SHARDS - 1);
                    const uint32_t c16 = tails[sh * MQ_SHARD_STRIDE];
                    const uint32_t h16 = __hip_atomic_load(&heads[sh * MQ_SHARD_STRIDE], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    const uint32_t has = (uint32_t)(__ballot(h16 * 64u < c16) & 0xffffull);
                    if (!has) exhausted = true;
                    else {
                        const uint32_t rot = ((has >> pool_s) | (has << (MQ_SHARDS - pool_s))) & 0xffffu; // bit i = shard (pool_s + i) & 15
                        pool_s = (pool_s + (uint32_t)__ffs((int)rot) - 1u) & (MQ_SHARDS - 1);
                    }
                }
            }
            PLAP(ctr, 25);
            const uint32_t avail = pool_len - pool_i;
            const uint32_t rank = (uint32_t)__popcll(idle & ((1ull << lane) - 1ull));
            // (a position past the buffer was never written -- queue_append drops such a ray and raises the overflow flag --: skipped)
            if (!busy && rank < avail && shard_pos(pool_s, (pool_j << 6) + pool_i + rank) < F.ray_cap) {
                const uint32_t i = pool_i + rank;
                q = shard_pos(pool_s, (pool_j << 6) + i); // the lane is busy from here on
                const float4* rays = ray_buffer(F, round);
                float4 o = rays[q], d = rays[(size_t)F.ray_cap + q];
                trav_init(t, F3(o.x, o.y, o.z), F3(d.x, d.y, d.z));
                trav_defer(sc, t, stk);
                fin = false;
                if (COUNT) ctr.rays++;
#ifdef MQ_PROF
                ray_iters = 0;
#endif
            }
            const uint32_t n_idle = (uint32_t)__popcll(idle);
            pool_i += n_idle < avail ? n_idle : avail;
            if (SHARE && exhausted && pool_i == pool_len) { // no ray left to fetch: idle lanes help busy ones
                const unsigned long long idles = __ballot(!busy);
                const bool can_give = busy && !fin && t.sp > t.sb && t.sb < MQ_STACK_LDS;
                const unsigned long long givers = __ballot(can_give);
                if ((uint32_t)__popcll(idles) >= MQ_SHARE_MIN_IDLE && givers) {
                    const unsigned long long lt = (1ull << lane) - 1ull;
                    const uint32_t g_rank = (uint32_t)__popcll(givers & lt), i_rank = (uint32_t)__popcll(idles & lt);
                    const uint32_t n_pair = (uint32_t)(__popcll(givers) < __popcll(idles) ? __popcll(givers) : __popcll(idles));
                    // giver of rank r posts its lane id to lane r; the idle lane of rank r then reads it from lane r
                    const int posted = __builtin_amdgcn_ds_permute((int)((can_give ? g_rank : 63u) << 2), lane);
                    const int from = __shfl(posted, (int)(i_rank < n_pair ? i_rank : 0u), 64);
                    const bool take = !busy && i_rank < n_pair;
                    const bool give = can_give && g_rank < n_pair;
                    // every lane reads its partner's registers (lanes that do not take read lane `from` = some giver: harmless)
                    const float ox = __shfl(t.o.x, from, 64), oy = __shfl(t.o.y, from, 64), oz = __shfl(t.o.z, from, 64);
                    const float dx = __shfl(t.d.x, from, 64), dy = __shfl(t.d.y, from, 64), dz = __shfl(t.d.z, from, 64);
                    const float ix = __shfl(t.idx, from, 64), iy = __shfl(t.idy, from, 64), iz = __shfl(t.idz, from, 64);
                    const uint32_t oc = (uint32_t)__shfl((int)t.oct4, from, 64), fq = (uint32_t)__shfl((int)q, from, 64);
                    const float ht = __shfl(t.hit.t, from, 64), tl = __shfl(t.tlim, from, 64);
                    const uint32_t bk = (uint32_t)__shfl((int)t.best_key, from, 64);
                    const int f_sb = __shfl(t.sb, from, 64), f_owner = __shfl(owner, from, 64);
                    if (take) {
                        t.o = F3(ox, oy, oz); t.d = F3(dx, dy, dz); t.idx = ix; t.idy = iy; t.idz = iz; t.oct4 = oc;
                        t.tlim = tl; t.hit.tri = MQ_NIL; t.hit.t = ht; t.hit.u = 0.0f; t.hit.v = 0.0f; t.best_key = bk; // only hits that beat the ray's current one are kept
                        t.G = (stk - lane)[f_sb * 64 + from]; // the giver's oldest entry
                        t.sp = 0; t.sb = 0; t.tmask = 0; t.tbase = 0;
                        q = fq; fin = false;
                        owner = f_owner >= 0 ? f_owner : from;
                        __hip_atomic_fetch_add(&pend[owner], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    if (give) t.sb++;
                }
            }
        }
        PLAP(ctr, 26);
        if (__ballot(busy) == 0ull) { if (exhausted) break; else continue; }
        // node phase: lanes without pending triangles visit one node
        const bool want_node = busy && !fin && t.tmask == 0;
#ifdef MQ_PROF
        { const uint32_t nn = (uint32_t)__popcll(__ballot(want_node)), nb = (uint32_t)__popcll(__ballot(busy)); if (nn) { ctr.prof[12]++; ctr.prof[13] += nn; } ctr.prof[11]++; ctr.prof[30] += nb;
          if (exhausted && pool_i == pool_len) { ctr.prof[22]++; ctr.prof[23] += nb; } }
#endif
        if (want_node) trav_node<COUNT>(sc, t, stk, spill, ctr);
        PLAP(ctr, 27);
        // triangle phase, by wave vote: run it only when enough lanes have triangles pending (or no
        // lane could use another node phase), so the expensive test executes at useful occupancy
        const bool has_tri = busy && t.tmask != 0;
        const unsigned long long tv = __ballot(has_tri);
        const uint32_t ntri = (uint32_t)__popcll(tv);
        const uint32_t nwork = (uint32_t)__popcll(__ballot(busy && !fin));
        if (ntri >= MQ_TRI_VOTE || ntri == nwork) {
#ifdef MQ_PROF
            if (ntri) { ctr.prof[14]++; ctr.prof[15] += ntri; }
#endif
            if (has_tri) trav_tri<COUNT>(sc, t, ctr);
        }
        PLAP(ctr, 28);
#ifdef MQ_PROF
        if (busy) ray_iters++;
#endif
        if (busy && !fin && t.tmask == 0) fin = trav_next(t, stk, spill);
        if (SHARE) { // finished helpers hand their closest hit to the owner lane, one at a time (wave-uniform loop)
            unsigned long long hm = __ballot(busy && fin && owner >= 0);
            while (hm) {
                const int l = __ffsll((long long)hm) - 1; hm &= hm - 1ull;
                const int root = __builtin_amdgcn_readlane(owner, l);
                const uint32_t h_tri = (uint32_t)__builtin_amdgcn_readlane((int)t.hit.tri, l), h_key = (uint32_t)__builtin_amdgcn_readlane((int)t.best_key, l);
                const float h_t = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t.hit.t), l));
                const float h_u = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t.hit.u), l));
                const float h_v = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(t.hit.v), l));
                // the owner takes the hit if it is closer; the ray's other helpers take it as their new limit
                if ((lane == root || owner == root) && h_tri != MQ_NIL && (h_t < t.hit.t || (h_t == t.hit.t && h_key < t.best_key))) {
                    if (lane == root) { t.hit.tri = h_tri; t.hit.u = h_u; t.hit.v = h_v; } else t.hit.tri = MQ_NIL; // a helper's own farther hit is obsolete
                    t.hit.t = h_t; t.best_key = h_key; t.tlim = trav_limit(h_t);
                }
                if (lane == l) { __hip_atomic_fetch_sub(&pend[root], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); q = MQ_NIL; owner = -1; fin = false; }
            }
        }
        if (busy && fin && owner < 0 && (!SHARE || __hip_atomic_load(&pend[lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT) == 0u)) {
            F.ray_hits[q] = make_uint4(t.hit.tri, __float_as_uint(t.hit.t), __float_as_uint(t.hit.u), __float_as_uint(t.hit.v));
            q = MQ_NIL; fin = false;
#ifdef MQ_PROF
            __hip_atomic_fetch_add(&s_hist[threadIdx.x >> 6][ray_iters / 8u < 63u ? ray_iters / 8u : 63u], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); // per wave in LDS: one global atomic per ray serialises the whole kernel on four cache lines
#endif
        }
        PLAP(ctr, 29);
    }
#ifdef MQ_PROF
    ctr.prof[31] = 1u; ctr.prof[37] = (uint32_t)__builtin_readcyclecounter() - prof_t0; // waves and their lifetimes: the section clocks of a wave must add up to its lifetime
    { const uint32_t h = s_hist[threadIdx.x >> 6][lane]; if (h) atomicAdd(&F.counters->ray_hist[lane], (unsigned long long)h); }
#endif
    PFLUSH(F.counters, ctr);
    if (COUNT) { // counted separately so the trace kernel's own algorithmic bytes can be priced
        flush_counters(F.counters, ctr);
        uint32_t v[3] = {ctr.rays, ctr.nodes, ctr.tris};
        for (int i = 0; i < 3; i++) {
            uint32_t x = v[i];
            for (int off = 32; off > 0; off >>= 1) x += __shfl_down(x, off, 64);
            if (lane == 0 && x) atomicAdd(&F.counters->q_rays + i, (unsigned long long)x);
        }
    }
}
#undef busy

// ---- a bounce ray returned: mcpg.comp:141-189, then the next direction ---------------------------
template <bool GUIDED, bool COUNT>
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_bounce_kernel(MqSceneDev sc, MqParams P, MqFrame F, int round) {
    extern __shared__ uint2 s_dyn[];
    float* lobes = GUIDED ? (float*)(s_dyn + (size_t)(threadIdx.x >> 6) * F.lds_rows2 * 64) + (threadIdx.x & 63) : nullptr;
    const mq_uniform& U = F.u;
    const QView qv = queue_view(F.qctrl + MQ_QTAILS(round));
    const uint32_t n = qv.n_eff < F.ray_cap ? qv.n_eff : F.ray_cap; // (positions past the buffer were never written, see queue_append)
    const f3 sun_color = F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]);
    Ctr ctr = {};
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t iters = (n + stride - 1) / stride;
    PSTART(ctr);
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t q = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool cont = false;
        uint32_t slot = 0;
        Path p = {};
        PLAP(ctr, 16);
        const bool valid = queue_valid(qv, q < n ? q : 0u);
        if (q < n && valid) {
            slot = F.queue_slots[round & 1][q];
            load_path(F.paths + slot, F.n_slots, ray_buffer(F, round)[(size_t)F.ray_cap + q], p);
            uint4 hq = F.ray_hits[q];
            RayHit rhit; rhit.tri = hq.x; rhit.t = __uint_as_float(hq.y); rhit.u = __uint_as_float(hq.z); rhit.v = __uint_as_float(hq.w);
            Hit next; next.wi = p.wo; next.pos = p.cur.pos - p.cur.wi * 1e-3f; next.prev_pos = next.pos; next.normal = F3(0, 0, 1); next.enc_geonormal = 0; next.albedo = F3(0, 0, 0); next.roughness = 0.0f;
            f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
            PLAP(ctr, 17);
            shade_hit(sc, P, U, rhit, throughput, incident, next, sun_color);
            PLAP(ctr, 18);
            f3 lc_incident; // mcpg.comp:149
            if ((incident.x > 0.0f || incident.y > 0.0f || incident.z > 0.0f) || (P.use_light_cache_tail == 0 && P.max_path_length == 2)) lc_incident = incident;
            else { lc_incident = rh3(throughput * light_cache_get(P, U, F.lc, p.rng, next.pos, next.normal)); if (COUNT) ctr.lc++; }
            PLAP(ctr, 19);
            p.thr = p.thr * p.bsdf;
            if (P.use_light_cache_tail) p.fval = p.thr * (p.seg < P.max_path_length - 1 ? incident : lc_incident);
            else p.fval = p.thr * incident;
            p.pp *= p.wo_p;
            p.thr = p.thr * throughput;
            if (GUIDED) { // mcpg.comp:165-181
                float mc_f = luminance((lc_incident * p.bsdf) * (1.0f / p.wo_p));
                if (mfinite(mc_f)) {
                    float den = P.quirk_lc_max_wo_p ? mmax(p.wo_p, 10.0f) : mmax(p.wo_p, 1e-6f);
                    light_cache_update(P, F, p.rng, p.cur.pos, p.cur.normal, ((lc_incident * (p.cur.albedo * MQ_INV_PI)) * p.wodotn) * (1.0f / den), ctr);
                    if (COUNT) ctr.lc++;
                    PLAP(ctr, 20);
                    if (xorshift(p.rng) * p.score_sum < mc_f * (float)P.mc_samples) {
                        f3 mv = rh3((next.pos - next.prev_pos) * (1.0f / U.cam_w[3]));
                        enqueue_update(P, F, p.rng, p.mc_index, p.mc_id, p.cur.pos, mc_f, next.pos, mv, p.cur.normal, ctr);
                    } else if (P.mc_fast_recovery && p.mc_index != MQ_NIL && !(mc_f > 1e-3f * p.mc_sum_w) && p.lm_dir_ok) {
                        if (P.log_learning) learn_log_simple(F, 3u, p.mc_index, 0u, 0u, 0u, 0u);
                        if (!P.freeze_learning) F.mc[p.mc_index].sum_w = 0.0f; // mcpg.comp:177
                    }
                }
            }
            PLAP(ctr, 21);
            p.thr = p.thr * next.albedo; // :184
            p.cur = next;
            PLAP(ctr, 35);
            bool need_dir = false, sample_done = false;
            if ((p.thr.x < 1e-7f && p.thr.y < 1e-7f && p.thr.z < 1e-7f) || (p.fval.x > 1e-7f || p.fval.y > 1e-7f || p.fval.z > 1e-7f)) sample_done = true;
            else { p.seg++; if (p.seg < P.max_path_length) need_dir = true; else sample_done = true; }
            PLAP(ctr, 36);
            cont = advance_path<GUIDED, COUNT>(P, F, p, slot, need_dir, sample_done, lobes, ctr);
        }
        PLAP(ctr, 9);
        uint32_t qn = queue_append(F, round + 1, cont);
        if (cont) emit_ray(F, round + 1, qn, slot, p);
        PLAP(ctr, 10);
    }
    if (COUNT) flush_counters(F.counters, ctr);
    PFLUSH(F.counters, ctr);
}


// ------------------------------------------------------------------------------------------------
// Single-scatter volume estimator: volume.comp:34-238, mc_distance.glsl, volume_forward_project.comp.
// Same wavefront scheme: per sample  mq_volume_sample_kernel (camera-distance + direction choice for
// every pixel)  ->  mq_trace_queue_kernel  ->  mq_volume_shade_kernel (contribution + learning),
// then mq_volume_finish_kernel.  Per-pixel state: six uint4 in the path record of the pixel slot.
// ------------------------------------------------------------------------------------------------
struct VPath {
    f3 irr, wo;
    float m2, t, pd, wo_p, score_sum, dist_score_sum, mc_sum_w;
    MqDistMC ds;
    uint32_t rng, mc_index, mc_id, px, py;
    bool lm_dir_ok;
};
MQ_DEV void store_vpath(uint4* d /* paths + slot */, size_t n, const VPath& v) {
    d[0 * n] = make_uint4(__float_as_uint(v.irr.x), __float_as_uint(v.irr.y), __float_as_uint(v.irr.z), __float_as_uint(v.m2));
    d[1 * n] = make_uint4(v.rng, __float_as_uint(v.t), __float_as_uint(v.pd), __float_as_uint(v.wo_p));
    d[2 * n] = make_uint4(__float_as_uint(v.wo.x), __float_as_uint(v.wo.y), __float_as_uint(v.wo.z), __float_as_uint(v.score_sum));
    d[3 * n] = make_uint4(__float_as_uint(v.dist_score_sum), __float_as_uint(v.ds.sum_w), v.ds.N, __float_as_uint(v.ds.m0));
    d[4 * n] = make_uint4(__float_as_uint(v.ds.m1), v.mc_index, v.mc_id, __float_as_uint(v.mc_sum_w));
    d[5 * n] = make_uint4(v.px | (v.py << 16), v.lm_dir_ok ? 1u : 0u, 0u, 0u);
}
MQ_DEV void load_vpath(const uint4* s /* paths + slot */, size_t n, VPath& v) {
    uint4 a = s[0], b = s[n], c = s[2 * n], d = s[3 * n], e = s[4 * n], f = s[5 * n];
    v.irr = F3(__uint_as_float(a.x), __uint_as_float(a.y), __uint_as_float(a.z)); v.m2 = __uint_as_float(a.w);
    v.rng = b.x; v.t = __uint_as_float(b.y); v.pd = __uint_as_float(b.z); v.wo_p = __uint_as_float(b.w);
    v.wo = F3(__uint_as_float(c.x), __uint_as_float(c.y), __uint_as_float(c.z)); v.score_sum = __uint_as_float(c.w);
    v.dist_score_sum = __uint_as_float(d.x); v.ds.sum_w = __uint_as_float(d.y); v.ds.N = d.z; v.ds.m0 = __uint_as_float(d.w);
    v.ds.m1 = __uint_as_float(e.x); v.mc_index = e.y; v.mc_id = e.z; v.mc_sum_w = __uint_as_float(e.w);
    v.px = f.x & 0xffffu; v.py = f.x >> 16; v.lm_dir_ok = (f.y & 1u) != 0;
}

MQ_DEV void distance_normal_dist(const MqDistMC& s, float& mu, float& sigma) { // mc_distance.glsl:10-16
    float den = s.sum_w > 0.0f ? s.sum_w : 1.0f;
    float m0 = s.m0 / den, m1 = s.m1 / den;
    float sg = sqrtf(mmax(m1 - m0 * m0, 0.0f));
    float n2 = (float)(s.N * s.N);
    mu = m0; sigma = (n2 * sg + 0.2f) / (n2 + 0.2f);
}
MQ_DEV void distance_add_sample(MqDistMC& s, float dist, float w) { // mc_distance.glsl:19-27
    s.N = s.N + 1 < 1024u ? s.N + 1 : 1024u;
    float alpha = mmax(1.0f / (float)s.N, 0.01f);
    s.sum_w = mmix(s.sum_w, w, alpha);
    s.m0 = mmix(s.m0, w * dist, alpha); s.m1 = mmix(s.m1, w * (dist * dist), alpha);
}
MQ_DEV uint32_t distance_mc_index(const MqParams& P, const MqFrame& F, uint32_t& rng, float px, float py, uint32_t grid_max_x) { // mc_distance.glsl:29-44
    float inv = 1.0f / (float)P.distance_mc_grid_width;
    float xi = xorshift(rng);
    int gx = (int)floorf(px * inv + xi), gy = (int)floorf(py * inv + xi);
    uint32_t st = (uint32_t)(xorshift(rng) * (float)P.distance_mc_vertex_state_count);
    uint32_t idx = ((uint32_t)gx + (grid_max_x + 1u) * (uint32_t)gy) * 10u + st;
    return idx < F.dist_mc_n ? idx : F.dist_mc_n - 1u;
}
MQ_DEV MqDistMC distance_mc_load(const MqFrame& F, uint32_t i) {
    float4 v = F.dist_mc[i];
    MqDistMC s; s.sum_w = v.x; s.N = __float_as_uint(v.y); s.m0 = v.z; s.m1 = v.w;
    return s;
}

// volume_forward_project.comp:17-53 is a SCATTER: every pixel projects last frame's scatter point into this frame and writes
// `pixel - target` at the target; pixels that collide race (the reference does not order them either).  Here the collisions are
// resolved: pass 1 keeps, per target, the LARGEST linear index of the pixels that hit it (one atomicMax), pass 2 writes that
// pixel's vector -- the image a sequential row-major sweep leaves (the last writer wins), i.e. the oracle's, bit for bit, so
// that the guided volume estimator, whose distance lookups read `volume_mv`, is deterministic with forward projection ON too.
__global__ void mq_forward_project_kernel(MqParams P, MqFrame F) {
    const mq_uniform& U = F.u;
    const float Wf = (float)F.W, Hf = (float)F.H;
    const uint32_t total = F.n_local_tiles * 64u;
    for (uint32_t my = blockIdx.x * blockDim.x + threadIdx.x; my < total; my += gridDim.x * blockDim.x) {
        uint32_t gtile = MQ_GTILE(F, my >> 6), within = my & 63u;
        uint32_t px = (gtile % F.tiles_x) * 8u + (within & 7u), py = (gtile / F.tiles_x) * 8u + (within >> 3);
        if (px >= F.W || py >= F.H) continue;
        float prev_depth = h2f(F.prev_volume_depth[(size_t)py * F.W + px]);
        f3 pwi = camera_ray_dir((float)px, (float)py, Wf, Hf, F3(U.prev_cam_u[0], U.prev_cam_u[1], U.prev_cam_u[2]), F3(U.prev_cam_w[0], U.prev_cam_w[1], U.prev_cam_w[2]), P.fov_tan_alpha_half);
        f3 ppos = F3(U.prev_cam_x[0], U.prev_cam_x[1], U.prev_cam_x[2]) + pwi * prev_depth;
        float fx, fy;
        camera_pixel(ppos - cam_pos(U), Wf, Hf, F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]), P.fov_tan_alpha_half, fx, fy);
        float rx = floorf(fx + 0.5f), ry = floorf(fy + 0.5f);
        if (!(rx >= 0.0f && ry >= 0.0f && rx < Wf && ry < Hf)) continue;
        if (prev_depth < 50.0f) continue;
        atomicMax(&F.fp_winner[(size_t)(int)ry * F.W + (size_t)(int)rx], py * F.W + px + 1u);
    }
}
__global__ void mq_forward_project_resolve_kernel(MqFrame F) { // every pixel: a rank of a partitioned frame scatters from its own pixels only, but onto any target
    const size_t n = (size_t)F.W * F.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t w = F.fp_winner[i];
        if (w == 0u) continue;
        F.fp_winner[i] = 0u; // ready for the next frame
        const uint32_t src = w - 1u, sx = src % F.W, sy = src / F.W, tx = (uint32_t)(i % F.W), ty = (uint32_t)(i / F.W);
        *(uint32_t*)(F.volume_mv + 2 * i) = (uint32_t)f2h((float)sx - (float)tx) | ((uint32_t)f2h((float)sy - (float)ty) << 16);
    }
}

// camera-distance sampling + direction choice (volume.comp:54-179) for sample `smp` of every pixel
template <bool COUNT>
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_volume_sample_kernel(MqSceneDev sc, MqParams P, MqFrame F, int smp, int round) {
    extern __shared__ uint2 s_dyn[]; // F.lds_rows2 rows of 64 x 8 bytes per wave: the candidates' (score, mu, sigma), then the lobes
    float* lobes = (float*)(s_dyn + (size_t)(threadIdx.x >> 6) * F.lds_rows2 * 64) + (threadIdx.x & 63);
    const mq_uniform& U = F.u;
    const float Wf = (float)F.W, Hf = (float)F.H;
    const uint32_t total = F.n_local_tiles * 64u;
    const uint32_t grid_max_x = F.W / (uint32_t)P.distance_mc_grid_width + 1u;
    const float mu_t = U.cam_x[3];
    const int KD = P.distance_mc_samples < MQ_MAX_MC_SAMPLES ? P.distance_mc_samples : MQ_MAX_MC_SAMPLES;
    const int K = P.mc_samples < MQ_MAX_MC_SAMPLES ? P.mc_samples : MQ_MAX_MC_SAMPLES;
    Ctr ctr = {};
    const uint32_t stride = gridDim.x * blockDim.x;
    const uint32_t iters = (total + stride - 1) / stride;
    for (uint32_t it = 0; it < iters; it++) {
        const uint32_t my = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        bool cont = false;
        VPath v = {};
        f3 first_wi = F3(0, 0, 1);
        if (my < total) {
            uint32_t gtile = MQ_GTILE(F, my >> 6), within = my & 63u;
            uint32_t px = (gtile % F.tiles_x) * 8u + (within & 7u), py = (gtile / F.tiles_x) * 8u + (within >> 3);
            if (px < F.W && py < F.H) {
                const size_t pidx = (size_t)py * F.W + px;
                if (smp == 0) { v.px = px; v.py = py; v.irr = F3(0, 0, 0); v.m2 = 0.0f; v.rng = pcg4d16(px, py, U.frame, P.seed); } // :45
                else load_vpath(F.paths + my, F.n_slots, v);
                uint4 gb = *(const uint4*)(F.gbuffer + 4 * pidx);
                const float linear_z = __uint_as_float(gb.y);
                first_wi = camera_ray_dir((float)px, (float)py, Wf, Hf, F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]), P.fov_tan_alpha_half);
                const uint32_t mvp = *(const uint32_t*)(F.volume_mv + 2 * pidx);
                const float mvx = h2f((uint16_t)(mvp & 0xffffu)), mvy = h2f((uint16_t)(mvp >> 16));
                const float tmax_v = mmin(linear_z, P.volume_max_t);
                bool skip = false;
                float pd = 0.0f, t = 0.0f;
                MqDistMC dstate; dstate.sum_w = 0.0f; dstate.N = 0; dstate.m0 = 0.0f; dstate.m1 = 0.0f;
                float dist_score_sum = 0.0f;
                { // :58-104 camera-distance sampling; (score, mu, sigma) of the candidates in LDS rows 0..3K-1
                    const float xi_max = transmittance_xi_max(tmax_v, mu_t);
#pragma nounroll
                    for (int i = 0; i < KD; i++) {
                        MqDistMC st; float nmu, nsg;
                        if (smp == 0) {
                            float qx = mclamp((float)px + mvx, 0.0f, Wf - 1.0f), qy = mclamp((float)py + mvy, 0.0f, Hf - 1.0f);
                            st = distance_mc_load(F, distance_mc_index(P, F, v.rng, qx, qy, grid_max_x));
                            distance_normal_dist(st, nmu, nsg);
                            nmu -= dot(cam_pos(U) - F3(U.prev_cam_x[0], U.prev_cam_x[1], U.prev_cam_x[2]), first_wi);
                        } else {
                            st = distance_mc_load(F, distance_mc_index(P, F, v.rng, (float)px, (float)py, grid_max_x));
                            distance_normal_dist(st, nmu, nsg);
                        }
                        float score = st.sum_w * (st.sum_w > 0.0f ? 1.0f : 0.0f) * (nmu < linear_z ? 1.0f : 0.0f);
                        lobes[(3 * i) * 64] = score; lobes[(3 * i + 1) * 64] = nmu; lobes[(3 * i + 2) * 64] = nsg;
                        dist_score_sum += score;
                        if (xorshift(v.rng) < score / dist_score_sum) {
                            dstate = st;
                            float x0 = xorshift(v.rng), x1 = xorshift(v.rng);
                            t = sample_normal_box_muller(nmu, nsg, x0, x1);
                        }
                    }
                    if (P.dist_guide_p < xorshift(v.rng) || dist_score_sum == 0.0f) t = transmittance_sample2(mu_t, xorshift(v.rng), xi_max);
                    else if (t >= tmax_v || t <= 0.0f) skip = true; // `continue` at :93
                    if (!skip) {
                        if (dist_score_sum > 0.0f) {
#pragma nounroll
                            for (int i = 0; i < KD; i++) pd += lobes[(3 * i) * 64] * sample_normal_pdf(lobes[(3 * i + 1) * 64], lobes[(3 * i + 2) * 64], t);
                            pd /= dist_score_sum;
                        }
                        pd = (dist_score_sum > 0.0f ? (1.0f - P.dist_guide_p) : 1.0f) * transmittance_pdf2(t, mu_t, xi_max) + P.dist_guide_p * pd;
                    }
                }
                if (!skip) { // :106-179 direction: Markov-chain lobes MIS'd with the Draine phase function
                    const f3 cur_pos = cam_pos(U) + first_wi * t;
                    const f3 nrm = -first_wi;
                    float wo_p = 0.0f, score_sum = 0.0f;
                    MCS sel = {}; uint32_t mc_index = MQ_NIL;
                    const uint32_t base_level = K > 0 ? mc_adaptive_base_level(P, U, cur_pos) : 0u;
#pragma nounroll
                    for (int i = 0; i < K; i++) {
                        const bool adapt = xorshift(v.rng) < P.mc_samples_adaptive_prob;
                        uint32_t bi, h16;
                        if (adapt) mc_adaptive_buffer_index_at(P, base_level, v.rng, cur_pos, nrm, bi, h16);
                        else mc_static_buffer_index(P, v.rng, cur_pos, bi, h16);
                        MCS st = mc_load(F.mc, bi);
                        if (COUNT) ctr.mc_reads++;
                        mc_finalize_load(U, st, h16, false, cur_pos, nrm); // volume lookups skip the below-surface test (mc.glsl:123-128)
                        score_sum += st.sum_w;
                        f3 d = mc_state_dir(st, cur_pos); float kk = mc_state_kappa(P, st, cur_pos);
                        float* li = lobes + (6 * i) * 64;
                        const float nm = vmf_norm(kk);
                        if (xorshift(v.rng) < st.sum_w / score_sum) {
                            sel = st; mc_index = bi;
                            if (i > 0) { li[0] = lobes[0]; li[64] = lobes[64]; li[128] = lobes[128]; li[192] = lobes[192]; li[256] = lobes[256]; li[320] = lobes[320]; }
                            lobes[0] = st.sum_w; lobes[64] = d.x; lobes[128] = d.y; lobes[192] = d.z; lobes[256] = kk; lobes[320] = nm;
                        } else { li[0] = st.sum_w; li[64] = d.x; li[128] = d.y; li[192] = d.z; li[256] = kk; li[320] = nm; }
                    }
                    f3 wo;
                    if (score_sum == 0.0f || xorshift(v.rng) < P.volume_phase_p) {
                        float x0 = xorshift(v.rng), x1 = xorshift(v.rng);
                        wo = draine_sample(x0, x1, first_wi, P.draine_g, P.draine_a);
                        sel = mc_state_new(v.rng);
                        mc_index = MQ_NIL;
                    } else {
                        float x0 = xorshift(v.rng), x1 = xorshift(v.rng);
                        wo = vmf_sample(F3(lobes[64], lobes[128], lobes[192]), lobes[256], x0, x1);
                    }
                    if (score_sum > 0.0f) {
#pragma nounroll
                        for (int i = 0; i < K; i++) {
                            const float* li = lobes + (6 * i) * 64;
                            wo_p += li[0] * vmf_pdf_normed(wo, F3(li[64], li[128], li[192]), li[256], li[320]);
                        }
                        wo_p /= score_sum;
                    }
                    wo_p = (score_sum > 0.0f ? P.volume_phase_p : 1.0f) * draine_eval(dot(first_wi, wo), P.draine_g, P.draine_a) + (1.0f - P.volume_phase_p) * wo_p;
                    v.t = t; v.pd = pd * wo_p; v.wo = wo; v.wo_p = wo_p; v.score_sum = score_sum; v.dist_score_sum = dist_score_sum; v.ds = dstate;
                    v.mc_index = mc_index; v.mc_id = sel.id; v.mc_sum_w = sel.sum_w;
                    v.lm_dir_ok = false;
                    if (mc_index != MQ_NIL) v.lm_dir_ok = !(dot(wo, mc_state_dir(sel, cur_pos)) < 0.9f + 0.1f * mc_state_mean_cos(P, sel, cur_pos));
                    cont = true;
                }
                store_vpath(F.paths + my, F.n_slots, v);
            }
        }
        uint32_t q = queue_append(F, round, cont);
        if (cont) {
            f3 ro = cam_pos(U) + first_wi * v.t;
            float4* rays = ray_buffer(F, round);
            rays[q] = make_float4(ro.x, ro.y, ro.z, 0.0f);
            rays[(size_t)F.ray_cap + q] = make_float4(v.wo.x, v.wo.y, v.wo.z, 0.0f);
            F.queue_slots[round & 1][q] = my;
        }
    }
    if (COUNT) flush_counters(F.counters, ctr);
}

// the scattered ray returned: contribution + learning (volume.comp:181-230)
template <bool COUNT>
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_volume_shade_kernel(MqSceneDev sc, MqParams P, MqFrame F, int smp, int round) {
    const mq_uniform& U = F.u;
    const float Wf = (float)F.W, Hf = (float)F.H;
    const QView qv = queue_view(F.qctrl + MQ_QTAILS(round));
    const uint32_t n = qv.n_eff < F.ray_cap ? qv.n_eff : F.ray_cap; // (positions past the buffer were never written, see queue_append)
    const uint32_t grid_max_x = F.W / (uint32_t)P.distance_mc_grid_width + 1u;
    const f3 sun_color = F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]);
    const f3 mu_s = F3(U.prev_cam_x[3], U.prev_cam_w[3], U.prev_cam_u[3]);
    const float mu_t = U.cam_x[3];
    Ctr ctr = {};
    const uint32_t vstride = gridDim.x * MQ_BLOCK;
    for (uint32_t it = 0; it < (n + vstride - 1) / vstride; it++) {
        const uint32_t q = it * vstride + blockIdx.x * MQ_BLOCK + threadIdx.x;
        const bool valid = queue_valid(qv, q < n ? q : 0u);
        if (!(q < n && valid)) continue;
        const uint32_t slot = F.queue_slots[round & 1][q];
        VPath v;
        load_vpath(F.paths + slot, F.n_slots, v);
        const size_t pidx = (size_t)v.py * F.W + v.px;
        const f3 first_wi = camera_ray_dir((float)v.px, (float)v.py, Wf, Hf, F3(U.cam_u[0], U.cam_u[1], U.cam_u[2]), F3(U.cam_w[0], U.cam_w[1], U.cam_w[2]), P.fov_tan_alpha_half);
        const f3 cur_pos = cam_pos(U) + first_wi * v.t;
        uint4 hq = F.ray_hits[q];
        RayHit rhit; rhit.tri = hq.x; rhit.t = __uint_as_float(hq.y); rhit.u = __uint_as_float(hq.z); rhit.v = __uint_as_float(hq.w);
        Hit next; next.wi = v.wo; next.pos = cur_pos; next.prev_pos = cur_pos; next.normal = F3(0, 0, 1); next.enc_geonormal = 0; next.albedo = F3(0, 0, 0); next.roughness = 0.0f;
        f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
        shade_hit(sc, P, U, rhit, throughput, incident, next, sun_color);
        if (P.volume_use_light_cache && !(incident.x > 0.0f || incident.y > 0.0f || incident.z > 0.0f)) { // :188-192
            incident = rh3(throughput * light_cache_get(P, U, F.lc, v.rng, next.pos, next.normal));
            if (COUNT) ctr.lc++;
        }
        const float phase = draine_eval(dot(first_wi, v.wo), P.draine_g, P.draine_a);
        const float tr = transmittance(v.t, mu_t, P.volume_max_t);
        f3 contrib = ((incident * phase) * mu_s) * (tr / v.pd); // :195
        if (mfinite(contrib.x) && mfinite(contrib.y) && mfinite(contrib.z)) {
            v.irr = v.irr + contrib;
            float l = luminance(contrib);
            v.m2 += l * l;
            distance_add_sample(v.ds, v.t, l); // :202
            if (smp == P.volume_spp - 1) {
                uint4 gb = *(const uint4*)(F.gbuffer + 4 * pidx);
                F.volume_depth[pidx] = f2h(v.ds.sum_w > 0.0f ? v.ds.m0 / v.ds.sum_w : __uint_as_float(gb.y));
            }
            if (xorshift(v.rng) < l / (v.dist_score_sum / (float)P.distance_mc_samples)) { // :213
                const uint32_t di = distance_mc_index(P, F, v.rng, (float)v.px, (float)v.py, grid_max_x);
                if (P.log_learning) learn_log_simple(F, 4u, di, __float_as_uint(v.ds.sum_w), v.ds.N, __float_as_uint(v.ds.m0), __float_as_uint(v.ds.m1));
                if (!P.freeze_learning) F.dist_mc[di] = make_float4(v.ds.sum_w, __uint_as_float(v.ds.N), v.ds.m0, v.ds.m1);
            }
            const float mc_f = luminance((incident * phase) * (1.0f / v.wo_p)); // :218
            if (xorshift(v.rng) < mc_f / (v.score_sum / (float)P.mc_samples)) {
                float x0 = xorshift(v.rng), x1 = xorshift(v.rng);
                f3 jn = sample_cos_frame(-first_wi, x0, x1);
                f3 mv = rh3((next.pos - next.prev_pos) * (1.0f / U.cam_w[3]));
                enqueue_update(P, F, v.rng, v.mc_index, v.mc_id, cur_pos, mc_f, next.pos, mv, jn, ctr);
            } else if (P.mc_fast_recovery && v.mc_index != MQ_NIL && !(mc_f > 1e-3f * v.mc_sum_w) && v.lm_dir_ok) {
                if (P.log_learning) learn_log_simple(F, 3u, v.mc_index, 0u, 0u, 0u, 0u);
                if (!P.freeze_learning) F.mc[v.mc_index].sum_w = 0.0f; // :228
            }
        }
        store_vpath(F.paths + slot, F.n_slots, v);
    }
    if (COUNT) flush_counters(F.counters, ctr);
}

__global__ void mq_volume_finish_kernel(MqParams P, MqFrame F) { // volume.comp:237
    const uint32_t total = F.n_local_tiles * 64u;
    const float inv = 1.0f / (float)(P.volume_spp > 1 ? P.volume_spp : 1);
    for (uint32_t my = blockIdx.x * blockDim.x + threadIdx.x; my < total; my += gridDim.x * blockDim.x) {
        uint32_t gtile = MQ_GTILE(F, my >> 6), within = my & 63u;
        uint32_t px = (gtile % F.tiles_x) * 8u + (within & 7u), py = (gtile / F.tiles_x) * 8u + (within >> 3);
        if (px >= F.W || py >= F.H) continue;
        uint4 a = F.paths[my];
        const float4 o4 = make_float4(__uint_as_float(a.x) * inv, __uint_as_float(a.y) * inv, __uint_as_float(a.z) * inv, __uint_as_float(a.w) * inv);
        *(float4*)(F.volume + 4 * ((size_t)py * F.W + px)) = o4;
        *(float4*)(F.volume_tiles_out + 4 * (size_t)my) = o4;
        F.vdepth_tiles_out[my] = F.volume_depth[(size_t)py * F.W + px]; // (a pixel without a finite sample keeps its old depth: the tile copy mirrors the image)
    }
}

// ------------------------------------------------------------------------------------------------
// update application: compute_updates.comp:41-124 over the compact queue
// ------------------------------------------------------------------------------------------------
MQ_DEV void mc_update(MCS& s, f3 pos, float w, f3 target, const uint16_t* mv) { // :41-54
    s.N = s.N + 1 < MQ_ML_MAX_N ? s.N + 1 : MQ_ML_MAX_N;
    float alpha = mmax(1.0f / (float)s.N, MQ_ML_MIN_ALPHA);
    s.sum_w = mmix(s.sum_w, w, alpha);
    s.w_tgt = F3(mmix(s.w_tgt.x, w * target.x, alpha), mmix(s.w_tgt.y, w * target.y, alpha), mmix(s.w_tgt.z, w * target.z, alpha));
    float co = mmax(0.0f, dot(normalize(target - pos), mc_state_dir(s, pos)));
    s.w_cos = mmin(mmix(s.w_cos, w * co, alpha), s.sum_w);
    s.mv[0] = mv[0]; s.mv[1] = mv[1]; s.mv[2] = mv[2];
}

// pass A: chain the queue entries of each slot (newest first) through `next`
__global__ __launch_bounds__(256) void mq_link_kernel(MqFrame F) {
    const QView qv = queue_view(F.ctrl + MQ_CTRL_UPDATES);
    const uint32_t n = qv.n_eff < F.queue_cap ? qv.n_eff : F.queue_cap;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t it = 0; it < (n + stride - 1) / stride; it++) {
        const uint32_t i = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = queue_valid(qv, i < n ? i : 0u);
        bool first = false; uint32_t slot = MQ_NIL;
        if (i < n && valid) {
            uint32_t* e3 = (uint32_t*)(F.queue + i) + 12;
            slot = e3[2];
            if (slot != MQ_NIL) { // (MQ_NIL: an update dropped by the per-slot cap)
                const uint32_t prev = atomicExch(&F.upd_head[slot], i + 1u);
                e3[3] = prev;
                first = prev == 0u; // the slot's first linked entry lists the slot for the apply pass
            }
        }
        const uint32_t at = shard_append(F.active_ctrl, first); // wave-wide; 16 tails: one counter would serialise ~40 k appends per frame
        if (first) { if (at < F.queue_cap) F.active[at] = slot; else atomicOr(&F.ctrl[0], 2u); } // cannot happen (the list is sized like the queue, whose entries it indexes); flagged, never silent
    }
}

// One slot of the update pass, compute_updates.comp:56-124: `head` = index of the newest queue entry of `slot` (the chain
// built by mq_link_kernel).  Returns the number of updates applied.
MQ_DEV uint32_t apply_slot(const MqParams& P, const MqFrame& F, uint32_t slot, uint32_t i /* head entry */) {
    const mq_uniform& U = F.u;
    uint32_t n_applied = 0;
    // The slot's entries by ARRIVAL RANK (the value the slot counter returned at enqueue, kept in the entry): the
    // chain links them in the order the link pass happened to see them, the replay below follows the ranks --
    // compute_updates.comp:73 walks update.ids[0 .. count) in exactly that order.  Only ranks < MQ_MAX_UPDATES
    // were ever written (mc.glsl:169-184); `present` tolerates holes (an entry lost to a full queue).
    uint32_t chain[MQ_MAX_UPDATES];
    uint32_t present = 0u;
    for (uint32_t at = i + 1u; at != 0u;) {
        const uint4 q3 = ((const uint4*)(F.queue + (at - 1u)))[3];
        const uint32_t r = q3.y >> 16;
        if (r < MQ_MAX_UPDATES) { chain[r] = at - 1u; present |= 1u << r; }
        at = q3.w;
    }
    const uint32_t count = (uint32_t)__popc(present);
    n_applied = count;
    uint32_t rng = pcg4d16(slot, 0u, U.frame, P.seed); // :62
    MCS mc_state = mc_load(F.mc, slot);
    float sum = 0.0f, upd_T = 0.0f;
    f3 pos = F3(0, 0, 0), normal = F3(0, 0, 0);
    MCS new_state = {}; bool picked = false;
    for (uint32_t m = present; m; m &= m - 1u) {
        const uint32_t r = (uint32_t)__ffs((int)m) - 1u;
        const uint4* q = (const uint4*)(F.queue + chain[r]);
        uint4 q0 = q[0], q1 = q[1], q2 = q[2], q3 = q[3];
        f3 upos = F3(__uint_as_float(q0.x), __uint_as_float(q0.y), __uint_as_float(q0.z));
        f3 utgt = F3(__uint_as_float(q1.x), __uint_as_float(q1.y), __uint_as_float(q1.z));
        uint16_t mv[3] = {(uint16_t)(q3.x & 0xffffu), (uint16_t)(q3.x >> 16), (uint16_t)(q3.y & 0xffffu)};
        if (m == present) upd_T = __uint_as_float(q2.w); // the first arrival stamps the slot (mc.glsl:171-173)
        MCS st = mc_state;
        if (mc_state.id != q1.w) st = mc_state_new(rng);
        mc_update(st, upos, __uint_as_float(q0.w), utgt, mv);
        if (mc_state.id == st.id) mc_state = st;
        sum += st.sum_w;
        if (xorshift(rng) < st.sum_w / sum) { new_state = st; pos = upos; normal = F3(__uint_as_float(q2.x), __uint_as_float(q2.y), __uint_as_float(q2.z)); picked = true; }
    }
    new_state.T = upd_T;
    if (picked) for (uint32_t k = 0; k < count; k++) { // :94-119
        { uint32_t bi, h16; mc_static_buffer_index(P, rng, pos, bi, h16);
          new_state.hash = h16; MCS old = mc_load(F.mc, bi);
          if (old.id == new_state.id || xorshift(rng) < new_state.sum_w / (new_state.sum_w + old.sum_w)) mc_store(F.mc, bi, new_state); }
        { uint32_t bi, h16; mc_adaptive_buffer_index(P, U, rng, pos, normal, bi, h16);
          new_state.hash = h16; MCS old = mc_load(F.mc, bi);
          if (old.id == new_state.id || xorshift(rng) < new_state.sum_w / (new_state.sum_w + old.sum_w)) mc_store(F.mc, bi, new_state); }
    }
    if (F.last_upd_count) F.last_upd_count[slot] = n_applied; // :121
    F.upd_head[slot] = 0u; // :121-122
    F.upd_count[slot] = 0u;
    return n_applied;
}

// The apply pass works from the list of slots the link pass left (F.active) and reads no queue tail, so
// its first block can zero the live control words of every queue of the frame right away -- the ray queues were consumed
// by the bounce kernels before it, the update queue by the link pass -- and the next frame needs no fill launch.  (A
// "last block done" counter would do too, but 2048 blocks bumping one address cost +70 us on this eight-die part.)
MQ_DEV void reset_queue_control(const MqFrame& F) {
    if (blockIdx.x == 0) for (uint32_t w = MQ_CTRL_UPDATES + threadIdx.x; w < F.ctrl_words; w += blockDim.x) F.ctrl[w] = 0u;
}

// pass B: the newest entry of a slot leads and replays compute_updates.comp:56-124 for the slot's
// first MQ_MAX_UPDATES arrivals (the reference drops later arrivals at enqueue time, mc.glsl:169-184)
__global__ __launch_bounds__(256) void mq_apply_kernel(MqParams P, MqFrame F) {
    // one thread per slot the link pass listed (every thread has work: a frame with volume samples queues millions of
    // entries for a third as many slots, and walking the entries to find each slot's newest one took 3x the rounds)
    const QView av = queue_view(F.active_ctrl);
    const uint32_t n_active = av.n_eff < F.queue_cap ? av.n_eff : F.queue_cap;
    uint32_t accepted = 0;
    const uint32_t stride = gridDim.x * blockDim.x;
    for (uint32_t it = 0; it < (n_active + stride - 1) / stride; it++) {
        const uint32_t j = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = queue_valid(av, j < n_active ? j : 0u); // (wave-wide)
        if (!(j < n_active && valid)) continue;
        const uint32_t slot = F.active[j];
        const uint32_t head = F.upd_head[slot];
        if (head != 0u) accepted += apply_slot(P, F, slot, head - 1u);
    }
    // statistics: updates applied this frame (those dropped by the cap are counted where they are dropped, at enqueue)
    for (int off = 32; off > 0; off >>= 1) accepted += __shfl_down(accepted, off, 64);
    if (F.count_stats && (threadIdx.x & 63) == 0 && accepted) atomicAdd(&F.counters->mc_updates_accepted, (unsigned long long)accepted);
    reset_queue_control(F);
}

// The same pass in the reference's own dispatch order, serialised (property "debug: sequential update pass", a test
// hook): compute_updates.comp runs one thread per slot of the whole table; here ONE lane walks the table's chain
// heads in ascending slot order, so what a slot reads of the table is what all lower slots left -- deterministic, where
// the parallel pass (like the reference's) lets slots that write the same cells race.
__global__ __launch_bounds__(64) void mq_apply_seq_kernel(MqParams P, MqFrame F, uint32_t mc_total) {
    const int lane = threadIdx.x & 63;
    uint32_t accepted = 0;
    for (uint32_t base = 0; base < mc_total; base += 64u) {
        const uint32_t s = base + (uint32_t)lane;
        const uint32_t head = s < mc_total ? F.upd_head[s] : 0u;
        unsigned long long m = __ballot(head != 0u);
        while (m) {
            const int l = __ffsll((long long)m) - 1; m &= m - 1ull;
            const uint32_t h = (uint32_t)__shfl((int)head, l, 64);
            if (lane == 0) accepted += apply_slot(P, F, base + (uint32_t)l, h - 1u);
        }
    }
    if (F.count_stats && lane == 0 && accepted) atomicAdd(&F.counters->mc_updates_accepted, (unsigned long long)accepted);
    reset_queue_control(F);
}

// ------------------------------------------------------------------------------------------------
// debug views, mcpg.comp:212-277 (grid_idx_closest(p, w) = floor(p / w + 0.5); acos / oklch_to_rgb: mq_device.h)
// ------------------------------------------------------------------------------------------------
__global__ void mq_debug_view_kernel(MqParams P, MqFrame F) {
    const mq_uniform& U = F.u;
    const uint32_t total = F.n_local_tiles * 64u;
    for (uint32_t my = blockIdx.x * blockDim.x + threadIdx.x; my < total; my += gridDim.x * blockDim.x) {
        uint32_t gtile = MQ_GTILE(F, my >> 6), within = my & 63u;
        uint32_t px = (gtile % F.tiles_x) * 8u + (within & 7u), py = (gtile / F.tiles_x) * 8u + (within >> 3);
        if (px >= F.W || py >= F.H) continue;
        const size_t pidx = (size_t)py * F.W + px;
        Hit h; load_chit(F.hits + 10 * pidx, h);
        uint32_t rng = F.debug_rng[pidx];
        const float4 irr4 = *(const float4*)(F.irradiance + 4 * pidx);
        const f3 irr = F3(irr4.x, irr4.y, irr4.z);
        f3 out = F3(0, 0, 0);
        MCS st = {};
        const int sel = P.debug_output_selector;
        if (sel == 1 || sel == 2 || sel == 6 || sel == 7 || sel == 8) { // mc_adaptive_load, mc.glsl:98-103
            uint32_t bi, h16;
            mc_adaptive_buffer_index(P, U, rng, h.pos, h.normal, bi, h16);
            st = mc_load(F.mc, bi);
            mc_finalize_load(U, st, h16, false, h.pos, h.normal);
        }
        switch (sel) {
        case 0: out = light_cache_get(P, U, F.lc, rng, h.pos, h.normal) * 5.0f; break;
        case 1: out = F3(st.sum_w * 0.1f, st.sum_w * 0.1f, st.sum_w * 0.1f); break;
        case 2: { f3 d = mc_state_dir(st, h.pos); out = F3((d.x + 1.0f) / 2.0f, (d.y + 1.0f) / 2.0f, (d.z + 1.0f) / 2.0f); break; }
        case 3: {
            uint32_t level = grid_level(P.adaptive_grid_type, P.mc_adaptive_grid_steps_per_unit_size, P.mc_adaptive_grid_tan_alpha_half, P.mc_adaptive_grid_min_width, P.mc_log_power, P.mc_inv_power, cam_pos(U), h.pos);
            i3 g = grid_idx_interpolate(h.pos, mc_inv_width(P, level), 0.5f);
            uint32_t seed = hash2_grid(g);
            float x0 = xorshift(seed), x1 = xorshift(seed);
            float L = mq_exp(0.001f * -length(h.pos - cam_pos(U))) * (0.0f + x0 * 1.0f) + 0.2f;
            out = oklch_to_rgb(F3(L, 0.2f, 6.28318548202514648f * x1));
            break; }
        case 4: out = irr; break;
        case 5: out = F3(luminance(irr), irr4.w, 0.0f); break;
        case 6: { float v = st.sum_w > 0.0f ? 1.0f - mclamp(mq_acos(st.w_cos / st.sum_w) * MQ_INV_PI, 0.0f, 1.0f) : 0.0f; out = F3(v, v, v); break; }
        case 7: { float v = (float)st.N / (float)MQ_ML_MAX_N; out = F3(v, v, v); break; }
        case 8: out = F3(h2f(st.mv[0]), h2f(st.mv[1]), h2f(st.mv[2])); break;
        default: break;
        }
        *(uint2*)(F.debug + 4 * pidx) = make_uint2((uint32_t)f2h(out.x) | ((uint32_t)f2h(out.y) << 16), (uint32_t)f2h(out.z) | (0x3c00u << 16));
    }
}

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
__global__ void mq_clear_kernel(MqFrame F) { // clear.comp:15-23, gbuffer.comp:83-90
    size_t n = (size_t)F.W * F.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        *(float4*)(F.irradiance + 4 * i) = make_float4(0, 0, 0, 0);
        *(uint2*)(F.gb_albedo + 4 * i) = make_uint2(0, 0);
        *(uint2*)(F.gb_irr + 4 * i) = make_uint2(0, 0);
        *(uint32_t*)(F.gb_mv + 2 * i) = 0;
        *(uint4*)(F.gbuffer + 4 * i) = make_uint4(0, 0, 0, 0);
        *(float4*)(F.volume + 4 * i) = make_float4(0, 0, 0, 0);
    }
    size_t nt = (size_t)F.n_local_tiles * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nt; i += (size_t)gridDim.x * blockDim.x)
        *(float4*)(F.tiles_out + 4 * i) = make_float4(0, 0, 0, 0);
}

// gathered: [rank][local tile][64 px][4] -> image
__global__ void mq_untile_kernel(const float4* gathered, float4* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank) {
    size_t n = (size_t)n_tiles * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t gtile = (uint32_t)(i >> 6), within = (uint32_t)(i & 63);
        uint32_t r = gtile % world, lt = gtile / world;
        uint32_t x = (gtile % tiles_x) * 8 + (within & 7), y = (gtile / tiles_x) * 8 + (within >> 3);
        if (x < W && y < H) image[(size_t)y * W + x] = gathered[((size_t)r * tiles_per_rank + lt) * 64 + within];
    }
}

// the same for a 16-bit image (volume_depth)
__global__ void mq_untile16_kernel(const uint16_t* gathered, uint16_t* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank) {
    size_t n = (size_t)n_tiles * 64;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint32_t gtile = (uint32_t)(i >> 6), within = (uint32_t)(i & 63);
        uint32_t r = gtile % world, lt = gtile / world;
        uint32_t x = (gtile % tiles_x) * 8 + (within & 7), y = (gtile / tiles_x) * 8 + (within >> 3);
        if (x < W && y < H) image[(size_t)y * W + x] = gathered[((size_t)r * tiles_per_rank + lt) * 64 + within];
    }
}

__global__ __launch_bounds__(MQ_BLOCK) void mq_trace_kernel(MqSceneDev sc, const float* org, const float* dir, uint32_t n, uint32_t* prim, float* t_out, float* uv, unsigned long long* spill_base) {
    __shared__ uint2 s_stack[MQ_WAVES][MQ_STACK_LDS][64];
    uint32_t gid = blockIdx.x * MQ_BLOCK + threadIdx.x;
    uint2* stk = &s_stack[threadIdx.x >> 6][0][threadIdx.x & 63];
    Ctr ctr = {};
    for (uint32_t i = gid; i < n; i += gridDim.x * MQ_BLOCK) {
        RayHit h;
        traverse<false>(sc, F3(org[3 * i], org[3 * i + 1], org[3 * i + 2]), F3(dir[3 * i], dir[3 * i + 1], dir[3 * i + 2]), h, stk, spill_base + (size_t)gid * MQ_SPILL_ENTRIES, ctr);
        prim[i] = h.tri == MQ_NIL ? MQ_NIL : sc.tris[h.tri].key;
        t_out[i] = h.tri == MQ_NIL ? MQ_T_MAX : h.t;
        if (uv) { uv[2 * i] = h.u; uv[2 * i + 1] = h.v; }
    }
}

// op codes mirror oracle/mq_oracle.h ORC_OP_* (tests map them one to one)
__global__ void mq_math_kernel(MqSceneDev sc, MqParams P, int op, int ni, int no, const float* in, float* out, uint32_t n) {
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const float* a = in + (size_t)k * ni; float* o = out + (size_t)k * no;
    switch (op) {
    case 0: o[0] = mq_exp2(a[0]); break;
    case 1: o[0] = mq_log2(a[0]); break;
    case 2: mq_sincos2pi(a[0], o[0], o[1]); break;
    case 3: o[0] = mq_pow(a[0], a[1]); break;
    case 4: o[0] = rh(a[0]); break;
    case 5: { uint32_t e = encode_normal(F3(a[0], a[1], a[2])); f3 d = decode_normal(e); o[0] = d.x; o[1] = d.y; o[2] = d.z; o[3] = __uint_as_float(e); break; }
    case 6: { f3 wi = F3(a[0], a[1], a[2]), nn = F3(a[3], a[4], a[5]); float al = roughness_to_alpha(a[6]);
        f3 wo = bsdf_sample(wi, nn, al, a[7], a[8], a[9]);
        o[0] = wo.x; o[1] = wo.y; o[2] = wo.z; o[3] = bsdf_pdf(wi, wo, nn, al); o[4] = bsdf_times_wodotn(wi, wo, nn, al, 0.02f); break; }
    case 7: { f3 mu = F3(a[0], a[1], a[2]); f3 w = vmf_sample(mu, a[3], a[4], a[5]); o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = vmf_pdf(w, mu, a[3]); break; }
    case 8: { uint32_t s = __float_as_uint(a[0]); for (int i = 0; i < 4; i++) o[i] = xorshift(s); break; }
    case 9: o[0] = __uint_as_float(pcg4d16(__float_as_uint(a[0]), __float_as_uint(a[1]), __float_as_uint(a[2]), __float_as_uint(a[3]))); break;
    case 10: { mq_uniform U = {}; U.sky_lf_ft = 0xfffe; U.sky_rt_bk = 0xffffffffu; U.sky_up_dn = 0xffffffffu;
        f3 s = get_sky(sc, P, U, F3(a[0], a[1], a[2]), F3(P.sun_color[0], P.sun_color[1], P.sun_color[2])); o[0] = s.x; o[1] = s.y; o[2] = s.z; break; }
    case 11: { i3 g = grid_idx_interpolate(F3(a[0], a[1], a[2]), 1.0f / a[7], 0.5f); uint32_t lv = (uint32_t)a[6];
        o[0] = __uint_as_float(hash_grid_normal_level(g, F3(a[3], a[4], a[5]), lv, __float_as_uint(a[8]))); o[1] = __uint_as_float(hash2_grid_level(g, lv)); break; }
    case 12: { f3 r = ldr_to_hdr(F3(a[0], a[1], a[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; break; }
    case 13: { f3 fwd = F3(a[4], a[5], a[6]), up = F3(a[7], a[8], a[9]);
        f3 d = camera_ray_dir(a[0], a[1], a[2], a[3], up, fwd, a[10]); o[0] = d.x; o[1] = d.y; o[2] = d.z;
        camera_pixel(d, a[2], a[3], up, fwd, a[10], o[3], o[4]); break; }
    case 14: { f3 wi = F3(a[0], a[1], a[2]); f3 w = draine_sample(a[5], a[6], wi, a[3], a[4]); o[0] = w.x; o[1] = w.y; o[2] = w.z; o[3] = draine_eval(dot(wi, w), a[3], a[4]); break; }
    case 15: { float xm = transmittance_xi_max(a[1], a[0]); o[0] = transmittance_sample2(a[0], a[2], xm); o[1] = transmittance_pdf2(o[0], a[0], xm);
        o[2] = sample_normal_box_muller(a[3], a[4], a[5], a[6]); o[3] = sample_normal_pdf(a[3], a[4], o[2]); break; }
    case 17: { mq_uniform U = {}; U.sky_rt_bk = __float_as_uint(a[3]); U.sky_lf_ft = __float_as_uint(a[4]); U.sky_up_dn = __float_as_uint(a[5]); U.cl_time = a[6];
        f3 s = get_sky(sc, P, U, F3(a[0], a[1], a[2]), F3(P.sun_color[0], P.sun_color[1], P.sun_color[2])); o[0] = s.x; o[1] = s.y; o[2] = s.z; break; }
    case 18: { uint32_t tn = (uint32_t)a[0]; if (tn > MQ_MAX_GLTEXTURES - 1) tn = MQ_MAX_GLTEXTURES - 1;
        f4 x = tex_sample_grad_desc(sc, sc.tex[tn], a[1], a[2], a[3], a[4], a[5], a[6]); o[0] = x.r; o[1] = x.g; o[2] = x.b; o[3] = x.a; break; }
    case 16: { f4 x = tex_sample(sc, (uint32_t)a[0], a[1], a[2]); o[0] = x.r; o[1] = x.g; o[2] = x.b; o[3] = x.a; break; }
    }
}

#include "mq_restir.h"

// ------------------------------------------------------------------------------------------------
// host-callable launchers (C++ linkage; used by mq_api.cpp)
// ------------------------------------------------------------------------------------------------
// dynamic LDS bytes of a shading block: F.lds_rows2 rows of 64 x 8 bytes per wave
static size_t shade_lds(const MqFrame& F) { return (size_t)F.lds_rows2 * 64 * 8 * (F.shade_block / 64); }
int mq_launch_primary(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, bool guided, bool count, int grid, hipStream_t s) {
    const size_t lds = shade_lds(F);

    if (guided) { if (count) mq_primary_kernel<true, true><<<grid, F.shade_block, lds, s>>>(sc, P, F); else mq_primary_kernel<true, false><<<grid, F.shade_block, lds, s>>>(sc, P, F); }
    else { if (count) mq_primary_kernel<false, true><<<grid, F.shade_block, lds, s>>>(sc, P, F); else mq_primary_kernel<false, false><<<grid, F.shade_block, lds, s>>>(sc, P, F); }
    return (int)hipGetLastError();
}
int mq_launch_primary_trace(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, bool packet, int grid, hipStream_t s) {
    if (packet) mq_primary_trace_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, F, P.fov_tan_alpha_half);
    else mq_primary_trace_lanes_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, F, P.fov_tan_alpha_half);
    return (int)hipGetLastError();
}
int mq_packet_stack_entries() { return MQ_PKT_STACK; }
int mq_launch_trace_queue(const MqSceneDev& sc, const MqFrame& F, int round, bool count, int grid, hipStream_t s) {
    if (count) mq_trace_queue_kernel<true><<<grid, MQ_BLOCK, 0, s>>>(sc, F, round); else mq_trace_queue_kernel<false><<<grid, MQ_BLOCK, 0, s>>>(sc, F, round);
    return (int)hipGetLastError();
}
int mq_launch_bounce(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int round, bool guided, bool count, int grid, hipStream_t s) {
    const size_t lds = guided ? shade_lds(F) : 0;
    if (guided) { if (count) mq_bounce_kernel<true, true><<<grid, F.shade_block, lds, s>>>(sc, P, F, round); else mq_bounce_kernel<true, false><<<grid, F.shade_block, lds, s>>>(sc, P, F, round); }
    else { if (count) mq_bounce_kernel<false, true><<<grid, F.shade_block, lds, s>>>(sc, P, F, round); else mq_bounce_kernel<false, false><<<grid, F.shade_block, lds, s>>>(sc, P, F, round); }
    return (int)hipGetLastError();
}
int mq_launch_apply(const MqParams& P, const MqFrame& F, int grid, uint32_t sequential_mc_total, hipStream_t s) {
    mq_link_kernel<<<grid, 256, 0, s>>>(F);
    if (sequential_mc_total) mq_apply_seq_kernel<<<1, 64, 0, s>>>(P, F, sequential_mc_total); // test hook: ascending slot order, one lane
    else mq_apply_kernel<<<grid, 256, 0, s>>>(P, F);
    return (int)hipGetLastError();
}
int mq_launch_debug_view(const MqParams& P, const MqFrame& F, int grid, hipStream_t s) {
    mq_debug_view_kernel<<<grid, 256, 0, s>>>(P, F);
    return (int)hipGetLastError();
}
int mq_launch_clear(const MqFrame& F, hipStream_t s) {
    mq_clear_kernel<<<1024, 256, 0, s>>>(F);
    return (int)hipGetLastError();
}
int mq_launch_untile(const void* gathered, void* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank, hipStream_t s) {
    mq_untile_kernel<<<1024, 256, 0, s>>>((const float4*)gathered, (float4*)image, W, H, tiles_x, n_tiles, world, tiles_per_rank);
    return (int)hipGetLastError();
}
int mq_launch_untile16(const void* gathered, void* image, uint32_t W, uint32_t H, uint32_t tiles_x, uint32_t n_tiles, uint32_t world, uint32_t tiles_per_rank, hipStream_t s) {
    mq_untile16_kernel<<<1024, 256, 0, s>>>((const uint16_t*)gathered, (uint16_t*)image, W, H, tiles_x, n_tiles, world, tiles_per_rank);
    return (int)hipGetLastError();
}
int mq_launch_trace(const MqSceneDev& sc, const float* org, const float* dir, uint32_t n, uint32_t* prim, float* t, float* uv, unsigned long long* spill, int grid, hipStream_t s) {
    mq_trace_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, org, dir, n, prim, t, uv, spill);
    return (int)hipGetLastError();
}
int mq_launch_math(const MqSceneDev& sc, const MqParams& P, int op, int ni, int no, const float* in, float* out, uint32_t n, hipStream_t s) {
    mq_math_kernel<<<(n + 255) / 256, 256, 0, s>>>(sc, P, op, ni, no, in, out, n);
    return (int)hipGetLastError();
}
int mq_launch_forward_project(const MqParams& P, const MqFrame& F, int grid, hipStream_t s) {
    mq_forward_project_kernel<<<grid, 256, 0, s>>>(P, F);
    mq_forward_project_resolve_kernel<<<grid, 256, 0, s>>>(F);
    return (int)hipGetLastError();
}
int mq_launch_volume_sample(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int smp, int round, bool count, int grid, hipStream_t s) {
    const size_t lds = shade_lds(F);
    if (count) mq_volume_sample_kernel<true><<<grid, F.shade_block, lds, s>>>(sc, P, F, smp, round); else mq_volume_sample_kernel<false><<<grid, F.shade_block, lds, s>>>(sc, P, F, smp, round);
    return (int)hipGetLastError();
}
int mq_launch_volume_shade(const MqSceneDev& sc, const MqParams& P, const MqFrame& F, int smp, int round, bool count, int grid, hipStream_t s) {
    if (count) mq_volume_shade_kernel<true><<<grid, MQ_BLOCK, 0, s>>>(sc, P, F, smp, round); else mq_volume_shade_kernel<false><<<grid, MQ_BLOCK, 0, s>>>(sc, P, F, smp, round);
    return (int)hipGetLastError();
}
int mq_launch_volume_finish(const MqParams& P, const MqFrame& F, int grid, hipStream_t s) {
    mq_volume_finish_kernel<<<grid, 256, 0, s>>>(P, F);
    return (int)hipGetLastError();
}
// streaming-read micro-benchmark ("achievable peak" beside the 8 TB/s spec figure): every lane
// reads 16 bytes per step, grid-stride, four independent accumulators
typedef uint32_t mq_u4v __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void mq_stream_read_kernel(const mq_u4v* __restrict__ src, size_t n16, uint32_t* sink) {
    mq_u4v a = {0, 0, 0, 0}, b = a, c = a, d = a;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    for (; i + 3 * stride < n16; i += 4 * stride) {
        a ^= __builtin_nontemporal_load(src + i); b ^= __builtin_nontemporal_load(src + i + stride);
        c ^= __builtin_nontemporal_load(src + i + 2 * stride); d ^= __builtin_nontemporal_load(src + i + 3 * stride);
    }
    for (; i < n16; i += stride) a ^= src[i];
    a ^= b; c ^= d; a ^= c;
    const uint32_t r = a.x ^ a.y ^ a.z ^ a.w;
    if (r == 0x9e3779b9u) *sink = r; // practically never: keeps the loads alive
}
int mq_launch_stream_read(const void* src, size_t bytes, uint32_t* sink, int grid, hipStream_t s) {
    mq_stream_read_kernel<<<grid, 256, 0, s>>>((const mq_u4v*)src, bytes / 16, sink);
    return (int)hipGetLastError();
}
int mq_render_block_size() { return MQ_BLOCK; }
int mq_spill_entries() { return MQ_SPILL_ENTRIES; }
int mq_stack_lds_entries() { return MQ_STACK_LDS; }
// resident blocks per CU of the three frame kernels at the given dynamic LDS size: {primary, trace, bounce}
int mq_resident_blocks(bool guided, size_t shade_lds_bytes, int shade_block, int out[4]) {
    int a = 0, b = 0, c = 0, d = 0;
    hipError_t e;
    if (guided) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, mq_primary_kernel<true, false>, shade_block, shade_lds_bytes);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&a, mq_primary_kernel<false, false>, shade_block, shade_lds_bytes);
    if (e != hipSuccess) return (int)e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&b, mq_trace_queue_kernel<false>, MQ_BLOCK, 0);
    if (e != hipSuccess) return (int)e;
    e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&d, mq_primary_trace_lanes_kernel, MQ_BLOCK, 0);
    if (e != hipSuccess) return (int)e;
    if (guided) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, mq_bounce_kernel<true, false>, shade_block, shade_lds_bytes);
    else e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&c, mq_bounce_kernel<false, false>, shade_block, 0);
    if (e != hipSuccess) return (int)e;
    out[0] = a; out[1] = b; out[2] = c; out[3] = d; // first hit, trace, bounce, camera rays
    return 0;
}
