// mq_restir.h -- ReSTIR DI render node, included by mq_kernels.hip after the traversal and
// shading code it shares with the MCPG node.
//
//   reference: src/render_restir/renderer_restir.cpp:104-251 (pass order, ping-pong of the reservoir buffers),
//   res/shader/render_restir/restir_di.glsl:41-156 (reservoir arithmetic), restir_di_common.glsl:7-18 (target
//   function), restir_di_generate_samples_bsdf.comp, restir_di_temporal_reuse.comp, restir_di_spatial_reuse.comp,
//   restir_di_shade.comp, restir_di_clear.comp.
//
// One thread per pixel, one wave per 8x8 tile (the reference's workgroup: the boiling filter averages over it).
// Rays are traced inline with the per-lane traversal (this node is a "next" row: correctness first).
// DEFINITIONS for what the reference takes from absent headers: first_hit = the decompressed
// CompressedHit of the g-buffer node (the snapshot declares `Hit hits[]`, layout.glsl:68-70, while gbuffer.comp writes
// CompressedHit -- the MCPG node's decompress_hit is used); reprojection_valid(n, n', cos, z, vel_z, z', r) =
// dot(n, n') >= cos && |z + vel_z - z'| <= r * max(z + vel_z, z'); round(x) = floor(x + 0.5); pow(d, 2) = d * d;
// yuv_luminance_f16 = the luminance of the half-precision radiance, rounded to half; boiling-filter sums in lane order.

struct Reservoir { // ReSTIRDIReservoir, restir_di_reservoir.glsl.h:8-27: 64 bytes in the scalar layout
    uint32_t M; float w; float p_target;
    f3 pos, normal, mv; float T;
    uint16_t rad[3]; uint32_t flags;
};
MQ_DEV Reservoir res_init() { Reservoir r; r.M = 0; r.w = 0.0f; r.p_target = 0.0f; r.pos = r.normal = r.mv = F3(0, 0, 0); r.T = 0.0f; r.rad[0] = r.rad[1] = r.rad[2] = 0; r.flags = 0; return r; }
MQ_DEV Reservoir res_load(const uint4* p) {
    const uint4 a = p[0], b = p[1], c = p[2], d = p[3];
    Reservoir r;
    r.M = a.x; r.w = __uint_as_float(a.y); r.p_target = __uint_as_float(a.z);
    r.pos = F3(__uint_as_float(a.w), __uint_as_float(b.x), __uint_as_float(b.y));
    r.normal = F3(__uint_as_float(b.z), __uint_as_float(b.w), __uint_as_float(c.x));
    r.mv = F3(__uint_as_float(c.y), __uint_as_float(c.z), __uint_as_float(c.w));
    r.T = __uint_as_float(d.x); r.rad[0] = (uint16_t)(d.y & 0xffffu); r.rad[1] = (uint16_t)(d.y >> 16); r.rad[2] = (uint16_t)(d.z & 0xffffu); r.flags = d.w;
    return r;
}
MQ_DEV void res_store(uint4* p, const Reservoir& r) {
    p[0] = make_uint4(r.M, __float_as_uint(r.w), __float_as_uint(r.p_target), __float_as_uint(r.pos.x));
    p[1] = make_uint4(__float_as_uint(r.pos.y), __float_as_uint(r.pos.z), __float_as_uint(r.normal.x), __float_as_uint(r.normal.y));
    p[2] = make_uint4(__float_as_uint(r.normal.z), __float_as_uint(r.mv.x), __float_as_uint(r.mv.y), __float_as_uint(r.mv.z));
    p[3] = make_uint4(__float_as_uint(r.T), (uint32_t)r.rad[0] | ((uint32_t)r.rad[1] << 16), (uint32_t)r.rad[2], r.flags);
}
MQ_DEV f3 res_radiance(const Reservoir& r) { return F3(h2f(r.rad[0]), h2f(r.rad[1]), h2f(r.rad[2])); }
MQ_DEV void res_discard(Reservoir& r) { r.w = 0.0f; r.flags = 0; r.rad[0] = r.rad[1] = r.rad[2] = 0; } // restir_di.glsl:54-58

// restir_di_common.glsl:7-18
MQ_DEV float restir_target_pdf(const Reservoir& y, const Hit& surface) {
    const f3 dv = y.pos - surface.pos;
    const f3 wo = normalize(dv);
    const float wodotn = dot(wo, surface.normal);
    if (wodotn <= 0.0f) return 0.0f;
    const float bsdf = bsdf_times_wodotn(surface.wi, wo, surface.normal, roughness_to_alpha(surface.roughness), 0.02f);
    const float dist = length(dv);
    return ((mmax(dot(y.normal, -wo), 0.0f) / (dist * dist)) * bsdf) * rh(luminance(res_radiance(y)));
}
// restir_di.glsl:69-88 (sample = the fields of `x` other than M / w / p_target)
MQ_DEV bool res_add_sample(Reservoir& r, uint32_t& rng, const Reservoir& x, float p_sample, float p_target) {
    const float w = p_target / p_sample;
    r.w += w; r.M += 1;
    if (xorshift(rng) * r.w < w) { r.p_target = p_target; r.pos = x.pos; r.normal = x.normal; r.mv = x.mv; r.T = x.T; r.rad[0] = x.rad[0]; r.rad[1] = x.rad[1]; r.rad[2] = x.rad[2]; r.flags = x.flags; return true; }
    return false;
}
// restir_di.glsl:125-141
MQ_DEV bool res_combine_finalized(Reservoir& r, uint32_t& rng, const Reservoir& o, float p_target_x_y) {
    r.M += o.M;
    const float w = (p_target_x_y * o.w) * (float)o.M;
    r.w += w;
    if (xorshift(rng) * r.w < w) { r.p_target = p_target_x_y; r.pos = o.pos; r.normal = o.normal; r.mv = o.mv; r.T = o.T; r.rad[0] = o.rad[0]; r.rad[1] = o.rad[1]; r.rad[2] = o.rad[2]; r.flags = o.flags; return true; }
    return false;
}
MQ_DEV void res_finalize(Reservoir& r) { const float den = (float)r.M * r.p_target; r.w = den > 0.0f ? r.w / den : 0.0f; }                                   // :146-149
MQ_DEV void res_finalize_custom(Reservoir& r, float num, float den) { den *= r.p_target; r.w = den > 0.0f ? (r.w * num) / den : 0.0f; }                      // :153-156
MQ_DEV bool reprojection_valid(f3 n, f3 pn, float cos_reject, float z, float vel_z, float pz, float depth_reject) {
    const float ze = z + vel_z;
    return dot(n, pn) >= cos_reject && fabsf(ze - pz) <= depth_reject * mmax(ze, pz);
}

// trace_ray of raytrace.glsl:156-311 from (hit.pos, hit.wi): closest hit + shading, as the MCPG kernels do it in two launches
MQ_DEV void restir_trace_ray(const MqSceneDev& sc, const MqParams& P, const mq_uniform& U, f3& throughput, f3& contribution, Hit& hit, uint2* stk, unsigned long long* spill, Ctr& ctr) {
    RayHit rhit;
    traverse<false>(sc, hit.pos, hit.wi, rhit, stk, spill, ctr);
    hit.prev_pos = hit.pos; hit.normal = F3(0, 0, 1); hit.enc_geonormal = 0; hit.albedo = F3(0, 0, 0); hit.roughness = 0.0f;
    shade_hit(sc, P, U, rhit, throughput, contribution, hit, F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]));
}
// trace_visibility of raytrace.glsl:66-150: nothing but sky between `from` and `to` (tmin 1e-3, tmax |to - from| - 2e-3)
MQ_DEV bool restir_trace_visibility(const MqSceneDev& sc, f3 from, f3 to, uint2* stk, unsigned long long* spill, Ctr& ctr) {
    const f3 wo = to - from;
    const float len = length(wo);
    RayHit rhit;
    traverse_range<false>(sc, from, normalize(wo), 1e-3f, mmax(1e-3f, len - 2.0f * 1e-3f), rhit, stk, spill, ctr);
    if (rhit.tri == MQ_NIL) return true;
    mq_ext e; MqTexDesc a, b;
    load_shade(sc, rhit.tri, e, a, b);
    return (e.texnum_fb_flags >> 12) == MQ_MAT_FLAGS_SKY;
}

struct RestirPixel { uint32_t px, py; size_t idx; bool inside; };
MQ_DEV RestirPixel restir_pixel(const MqRestirFrame& F, uint32_t tile, int lane) {
    RestirPixel p;
    p.px = (tile % F.tiles_x) * 8u + ((uint32_t)lane & 7u); p.py = (tile / F.tiles_x) * 8u + ((uint32_t)lane >> 3);
    p.inside = p.px < F.W && p.py < F.H;
    p.idx = (size_t)p.py * F.W + p.px;
    return p;
}
#define MQ_RESTIR_SETUP \
    __shared__ uint2 s_stack[MQ_WAVES][MQ_STACK_LDS][64]; \
    const int lane = threadIdx.x & 63; \
    uint2* stk = &s_stack[threadIdx.x >> 6][0][lane]; \
    unsigned long long* spill = F.stack_spill + (size_t)(blockIdx.x * MQ_BLOCK + threadIdx.x) * MQ_SPILL_ENTRIES; \
    const mq_uniform& U = F.u; \
    Ctr ctr = {}; (void)ctr; (void)stk; (void)spill; (void)U; \
    const uint32_t n_waves = gridDim.x * MQ_WAVES;

// restir_di_generate_samples_bsdf.comp:23-62
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_generate_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F) {
    MQ_RESTIR_SETUP
    for (uint32_t tile = F.tile_begin + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6); tile < F.tile_end; tile += n_waves) {
        const RestirPixel p = restir_pixel(F, tile, lane);
        if (!p.inside) continue;
        uint32_t rng = pcg4d16(p.px, p.py, U.frame * 4u + 0u, R.seed);
        Reservoir r = res_init();
        Hit first; load_chit(F.hits + 10 * p.idx, first);
        if (first.albedo.x >= 1e-7f || first.albedo.y >= 1e-7f || first.albedo.z >= 1e-7f)
            for (int s = 0; s < R.spp; s++) {
                const float alpha = roughness_to_alpha(first.roughness);
                const float x0 = xorshift(rng), x1 = xorshift(rng), x2 = xorshift(rng);
                const f3 wo = bsdf_sample(first.wi, first.normal, alpha, x0, x1, x2);
                const float wodotn = dot(wo, first.normal);
                if (dot(wo, decode_normal(first.enc_geonormal)) <= 1e-3f || wodotn <= 1e-3f) continue;
                Hit next; next.wi = wo; next.pos = first.pos - first.wi * 1e-3f;
                f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
                restir_trace_ray(sc, P, U, throughput, incident, next, stk, spill, ctr);
                const float dist = length(next.pos - first.pos);
                const float geo = mmax(dot(next.normal, -wo), 0.0f) / (dist * dist);
                Reservoir x = res_init();
                x.pos = next.pos; x.normal = next.normal; x.mv = (next.pos - next.prev_pos) * (1.0f / U.cam_w[3]); x.T = U.cl_time;
                x.rad[0] = f2h(incident.x); x.rad[1] = f2h(incident.y); x.rad[2] = f2h(incident.z); x.flags = 1u;
                res_add_sample(r, rng, x, geo * bsdf_pdf(first.wi, wo, first.normal, alpha), restir_target_pdf(x, first));
            }
        res_finalize(r);
        res_store(F.res_a + 4 * p.idx, r);
    }
}

// ---- the same two passes as a wavefront: request the ray -> mq_trace_queue_kernel -> finish ---------------------
// The generate and shade passes each trace one incoherent closest-hit ray per pixel.  Inline (kernels above / below) the
// ray is traversed at the register budget of the shading code, one ray per lane to the end of the longest ray of the wave;
// split, the rays go through the MCPG node's queues (FQ: its ray / hit buffers, slot lists and counters, free between
// its frames) and its persistent traversal kernel (refill, triangle vote, work sharing).  Pixel slot = tile * 64 + lane,
// as the MCPG kernels number them.  Arithmetic and the order of the random numbers are those of the inline kernels: the
// direction is stored and read back as floats, everything else is recomputed from the same inputs.
// scratch: one u32 per pixel slot (the random-number state between the two halves), in the MCPG node's path records.
MQ_DEV void restir_emit(const MqFrame& FQ, int round, bool push, uint32_t slot, f3 o, f3 d) {
    uint32_t q = queue_append(FQ, round, push); // wave-wide: every lane of the wave calls it
    if (push) {
        float4* rays = ray_buffer(FQ, round);
        rays[q] = make_float4(o.x, o.y, o.z, 0.0f);
        rays[(size_t)FQ.ray_cap + q] = make_float4(d.x, d.y, d.z, 0.0f);
        FQ.queue_slots[round & 1][q] = slot;
    }
}
// restir_di_generate_samples_bsdf.comp:23-62, sample `smp`: up to the ray
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_generate_a_kernel(MqParams P, MqRestirParams R, MqRestirFrame F, MqFrame FQ, int smp) {
    const int lane = threadIdx.x & 63;
    const mq_uniform& U = F.u;
    uint32_t* scratch = (uint32_t*)FQ.paths;
    const uint32_t n_waves = gridDim.x * MQ_WAVES;
    const uint32_t tiles_per_wave = (F.tile_end - F.tile_begin + n_waves - 1) / n_waves;
    for (uint32_t it = 0; it < tiles_per_wave; it++) { // every lane of a wave runs the same trips: the append is wave-wide
        const uint32_t tile = F.tile_begin + it * n_waves + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6);
        bool push = false; f3 ro = F3(0, 0, 0), wo = F3(0, 0, 1);
        const uint32_t slot = (tile - F.slot_tile0) * 64u + (uint32_t)lane; // pixel slots count from the first tile of the rank's rows
        if (tile < F.tile_end) {
            const RestirPixel p = restir_pixel(F, tile, lane);
            if (p.inside) {
                Hit first; load_chit(F.hits + 10 * p.idx, first);
                const bool lit = first.albedo.x >= 1e-7f || first.albedo.y >= 1e-7f || first.albedo.z >= 1e-7f;
                if (lit && smp < R.spp) {
                    uint32_t rng = smp == 0 ? pcg4d16(p.px, p.py, U.frame * 4u + 0u, R.seed) : scratch[slot];
                    const float alpha = roughness_to_alpha(first.roughness);
                    const float x0 = xorshift(rng), x1 = xorshift(rng), x2 = xorshift(rng);
                    wo = bsdf_sample(first.wi, first.normal, alpha, x0, x1, x2);
                    const float wodotn = dot(wo, first.normal);
                    push = !(dot(wo, decode_normal(first.enc_geonormal)) <= 1e-3f || wodotn <= 1e-3f);
                    ro = first.pos - first.wi * 1e-3f;
                    scratch[slot] = rng;
                }
                if (smp == 0) res_store(F.res_a + 4 * p.idx, res_init()); // the finishing half adds the sample to what is stored
                if (!push && (smp >= R.spp - 1 || !lit)) { // no sample to wait for: this half finishes the pixel
                    if (smp == 0 || lit) { Reservoir r = smp == 0 ? res_init() : res_load(F.res_a + 4 * p.idx); res_finalize(r); res_store(F.res_a + 4 * p.idx, r); }
                }
            }
        }
        restir_emit(FQ, smp, push, slot, ro, wo);
    }
}
// ... from the returned hit on
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_generate_b_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F, MqFrame FQ, int smp) {
    const mq_uniform& U = F.u;
    uint32_t* scratch = (uint32_t*)FQ.paths;
    const QView qv = queue_view(FQ.qctrl + MQ_QTAILS(smp));
    const uint32_t n = qv.n_eff < FQ.ray_cap ? qv.n_eff : FQ.ray_cap, stride = gridDim.x * blockDim.x;
    for (uint32_t it = 0; it < (n + stride - 1) / stride; it++) {
        const uint32_t q = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = queue_valid(qv, q < n ? q : 0u);
        if (!(q < n && valid)) continue;
        const uint32_t slot = FQ.queue_slots[smp & 1][q];
        const RestirPixel p = restir_pixel(F, (slot >> 6) + F.slot_tile0, (int)(slot & 63u));
        Hit first; load_chit(F.hits + 10 * p.idx, first);
        const float4 d4 = ray_buffer(FQ, smp)[(size_t)FQ.ray_cap + q];
        const f3 wo = F3(d4.x, d4.y, d4.z);
        const uint4 hq = FQ.ray_hits[q];
        RayHit rhit; rhit.tri = hq.x; rhit.t = __uint_as_float(hq.y); rhit.u = __uint_as_float(hq.z); rhit.v = __uint_as_float(hq.w);
        const float alpha = roughness_to_alpha(first.roughness);
        Hit next; next.wi = wo; next.pos = first.pos - first.wi * 1e-3f;
        next.prev_pos = next.pos; next.normal = F3(0, 0, 1); next.enc_geonormal = 0; next.albedo = F3(0, 0, 0); next.roughness = 0.0f;
        f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
        shade_hit(sc, P, U, rhit, throughput, incident, next, F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]));
        const float dist = length(next.pos - first.pos);
        const float geo = mmax(dot(next.normal, -wo), 0.0f) / (dist * dist);
        Reservoir x = res_init();
        x.pos = next.pos; x.normal = next.normal; x.mv = (next.pos - next.prev_pos) * (1.0f / U.cam_w[3]); x.T = U.cl_time;
        x.rad[0] = f2h(incident.x); x.rad[1] = f2h(incident.y); x.rad[2] = f2h(incident.z); x.flags = 1u;
        uint32_t rng = scratch[slot];
        Reservoir r = res_load(F.res_a + 4 * p.idx);
        res_add_sample(r, rng, x, geo * bsdf_pdf(first.wi, wo, first.normal, alpha), restir_target_pdf(x, first));
        scratch[slot] = rng;
        if (smp >= R.spp - 1) res_finalize(r);
        res_store(F.res_a + 4 * p.idx, r);
    }
}
// restir_di_shade.comp:21-62 up to the ray
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_shade_a_kernel(MqRestirParams R, MqRestirFrame F, MqFrame FQ, int round) {
    const int lane = threadIdx.x & 63;
    const uint32_t n_waves = gridDim.x * MQ_WAVES;
    const uint32_t tiles_per_wave = (F.tile_end - F.tile_begin + n_waves - 1) / n_waves;
    for (uint32_t it = 0; it < tiles_per_wave; it++) {
        const uint32_t tile = F.tile_begin + it * n_waves + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6);
        bool push = false; f3 ro = F3(0, 0, 0), wo = F3(0, 0, 1);
        const uint32_t slot = (tile - F.slot_tile0) * 64u + (uint32_t)lane; // pixel slots count from the first tile of the rank's rows
        if (tile < F.tile_end) {
            const RestirPixel p = restir_pixel(F, tile, lane);
            if (p.inside) {
                const Reservoir r = res_load(F.res_a + 4 * p.idx);
                if (r.flags & 1u) {
                    Hit first; load_chit(F.hits + 10 * p.idx, first);
                    wo = normalize(r.pos - first.pos);
                    ro = first.pos - first.wi * 1e-3f;
                    push = true;
                } else { F.irradiance[p.idx] = make_float4(0.0f, 0.0f, 0.0f, 1.0f); F.moments[p.idx] = make_float2(0.0f, 0.0f); }
            }
        }
        restir_emit(FQ, round, push, slot, ro, wo);
    }
}
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_shade_b_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F, MqFrame FQ, int round) {
    const mq_uniform& U = F.u;
    const QView qv = queue_view(FQ.qctrl + MQ_QTAILS(round));
    const uint32_t n = qv.n_eff < FQ.ray_cap ? qv.n_eff : FQ.ray_cap, stride = gridDim.x * blockDim.x;
    for (uint32_t it = 0; it < (n + stride - 1) / stride; it++) {
        const uint32_t q = it * stride + blockIdx.x * blockDim.x + threadIdx.x;
        const bool valid = queue_valid(qv, q < n ? q : 0u);
        if (!(q < n && valid)) continue;
        const uint32_t slot = FQ.queue_slots[round & 1][q];
        const RestirPixel p = restir_pixel(F, (slot >> 6) + F.slot_tile0, (int)(slot & 63u));
        Reservoir r = res_load(F.res_a + 4 * p.idx);
        Hit first; load_chit(F.hits + 10 * p.idx, first);
        const f3 dv = r.pos - first.pos;
        const float4 d4 = ray_buffer(FQ, round)[(size_t)FQ.ray_cap + q];
        const f3 wo = F3(d4.x, d4.y, d4.z);
        const uint4 hq = FQ.ray_hits[q];
        RayHit rhit; rhit.tri = hq.x; rhit.t = __uint_as_float(hq.y); rhit.u = __uint_as_float(hq.z); rhit.v = __uint_as_float(hq.w);
        Hit next; next.wi = wo; next.pos = first.pos - first.wi * 1e-3f;
        next.prev_pos = next.pos; next.normal = F3(0, 0, 1); next.enc_geonormal = 0; next.albedo = F3(0, 0, 0); next.roughness = 0.0f;
        f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
        shade_hit(sc, P, U, rhit, throughput, incident, next, F3(P.sun_color[0], P.sun_color[1], P.sun_color[2]));
        f3 irr = F3(0, 0, 0);
        const float d_sample = length(dv), d_hit = length(first.pos - next.pos);
        if (R.visibility_shade && fabsf(d_sample - d_hit) / mmax(d_sample, d_hit) > 0.1f) { res_discard(r); res_store(F.res_a + 4 * p.idx, r); } // not visible
        const float bsdf = bsdf_times_wodotn(first.wi, wo, first.normal, roughness_to_alpha(first.roughness), 0.02f);
        if (mfinite(r.w)) irr = ((res_radiance(r) * bsdf) * r.w) * (mmax(dot(r.normal, -wo), 0.0f) / (d_sample * d_sample));
        F.irradiance[p.idx] = make_float4(irr.x, irr.y, irr.z, 1.0f);
        const float l = luminance(irr);
        F.moments[p.idx] = make_float2(l, l * l);
    }
}

// restir_di_temporal_reuse.comp:71-146 (+ the boiling filter, :37-69, over the 8x8 tile = this wave)
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_temporal_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F) {
    MQ_RESTIR_SETUP
    for (uint32_t tile = F.tile_begin + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6); tile < F.tile_end; tile += n_waves) {
        const RestirPixel p = restir_pixel(F, tile, lane);
        bool active = p.inside;
        Reservoir r = res_init();
        if (active) {
            uint32_t rng = pcg4d16(p.px, p.py, U.frame * 4u + 1u, R.seed);
            const Reservoir cur = res_load(F.res_a + 4 * p.idx);
            res_combine_finalized(r, rng, cur, cur.p_target);
            const uint32_t m = F.mv[p.idx];
            const float qx = floorf(((float)p.px + h2f((uint16_t)(m & 0xffffu))) + 0.5f), qy = floorf(((float)p.py + h2f((uint16_t)(m >> 16))) + 0.5f);
            active = qx >= 0.0f && qy >= 0.0f && qx < (float)F.W && qy < (float)F.H; // :82-84 (a NaN motion vector leaves the image too)
            if (active && !((uint32_t)qy >= F.row_lo && (uint32_t)qy < F.row_hi)) { atomicOr(F.flags, 8u); active = false; } // a rank of a row partition: beyond the rows it holds (flagged: "restir: reprojection halo" is too small for this motion)
            if (active) {
                const size_t q = (size_t)(uint32_t)qy * F.W + (uint32_t)qx;
                const uint4 g = F.gbuffer[p.idx], pg = F.prev_gbuffer[q];
                active = reprojection_valid(decode_normal(g.x), decode_normal(pg.x), R.temporal_normal_reject_cos, __uint_as_float(g.y), __uint_as_float(g.w), __uint_as_float(pg.y), R.temporal_depth_reject);
                if (active) {
                    Hit center; load_chit(F.hits + 10 * p.idx, center);
                    Reservoir prev = res_load(F.prev_reservoirs + 4 * q);
                    if (R.apply_mv == 1) { prev.pos = prev.pos + prev.mv * (U.cl_time - prev.T); prev.T = U.cl_time; }
                    if (R.temporal_clamp_m > 0) prev.M = prev.M < (uint32_t)R.temporal_clamp_m ? prev.M : (uint32_t)R.temporal_clamp_m;
                    const bool selected_prev = res_combine_finalized(r, rng, prev, restir_target_pdf(prev, center));
                    if (R.temporal_bias_correction == 0) res_finalize(r);
                    else { // :110-138
                        float pi = r.p_target, pi_sum = r.p_target * (float)cur.M;
                        Hit psurf; load_chit(F.hits + 10 * q, psurf); // surface_at(prev_pixel): THIS frame's record at that pixel, as the reference reads it
                        float temporal_p = restir_target_pdf(r, psurf);
                        if (temporal_p > 0.0f) {
                            if (R.temporal_bias_correction == 2 && !restir_trace_visibility(sc, center.pos, r.pos, stk, spill, ctr)) temporal_p = 0.0f;
                            if (R.temporal_bias_correction == 3) temporal_p = 0.0f; // "prev bvh currently unsupported"
                        }
                        pi = selected_prev ? temporal_p : pi;
                        pi_sum += temporal_p * (float)prev.M;
                        res_finalize_custom(r, pi, pi_sum);
                    }
                }
            }
        }
        if (R.boiling_filter_strength > 1e-6f) { // lanes that returned early in the reference take no part
            const float mult = 10.0f / R.boiling_filter_strength - 9.0f;
            unsigned long long am = __ballot(active);
            float sum = 0.0f; uint32_t count = 0;
            while (am) { // lane order: the definition of the group sum
                const int l = __ffsll((long long)am) - 1; am &= am - 1ull;
                const float wl = __uint_as_float((uint32_t)__builtin_amdgcn_readlane((int)__float_as_uint(r.w), l));
                sum += wl; count += wl > 0.0f ? 1u : 0u;
            }
            const float avg = count > 0 ? sum / (float)count : 0.0f;
            if (active && r.w > avg * mult) res_discard(r);
        }
        if (active) res_store(F.res_a + 4 * p.idx, r);
    }
}

#define MQ_RESTIR_MAX_NEIGHBORS 7 // renderer_restir.cpp:301: config_int("spatial reuse iterations", ..., 0, 7)
// restir_di_spatial_reuse.comp:26-101
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_spatial_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F) {
    MQ_RESTIR_SETUP
    const int NI = R.spatial_reuse_iterations < 1 ? 1 : (R.spatial_reuse_iterations > MQ_RESTIR_MAX_NEIGHBORS ? MQ_RESTIR_MAX_NEIGHBORS : R.spatial_reuse_iterations);
    for (uint32_t tile = F.tile_begin + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6); tile < F.tile_end; tile += n_waves) {
        const RestirPixel p = restir_pixel(F, tile, lane);
        if (!p.inside) continue;
        uint32_t rng = pcg4d16(p.px, p.py, U.frame * 4u + 2u, R.seed);
        Reservoir r = res_init();
        const Reservoir cur = res_load(F.res_read + 4 * p.idx);
        res_combine_finalized(r, rng, cur, cur.p_target);
        Hit center; load_chit(F.hits + 10 * p.idx, center);
        const uint4 g = F.gbuffer[p.idx];
        int selected = -1;
        uint32_t nq[MQ_RESTIR_MAX_NEIGHBORS]; // linear pixel index of neighbour i, MQ_NIL if rejected
        for (int i = 0; i < NI; i++) {
            const float x0 = xorshift(rng), x1 = xorshift(rng);
            const float nx = floorf(((float)p.px + (float)R.spatial_radius * (2.0f * x0 - 1.0f)) + 0.5f), ny = floorf(((float)p.py + (float)R.spatial_radius * (2.0f * x1 - 1.0f)) + 0.5f);
            nq[i] = MQ_NIL;
            if (!(nx >= 0.0f && ny >= 0.0f && nx < (float)F.W && ny < (float)F.H)) continue;
            if (!((uint32_t)ny >= F.row_lo && (uint32_t)ny < F.row_hi)) { atomicOr(F.flags, 8u); continue; } // (cannot happen: the rows a rank holds include the spatial radius)
            const uint32_t q = (uint32_t)ny * F.W + (uint32_t)nx;
            const uint4 ng = F.gbuffer[q];
            if (!reprojection_valid(decode_normal(g.x), decode_normal(ng.x), R.spatial_normal_reject_cos, __uint_as_float(g.y), __uint_as_float(g.w), __uint_as_float(ng.y), R.spatial_depth_reject)) continue;
            nq[i] = q;
            const Reservoir nb = res_load(F.res_read + 4 * (size_t)q);
            if (res_combine_finalized(r, rng, nb, restir_target_pdf(nb, center))) selected = i;
        }
        if (R.spatial_bias_correction == 0) res_finalize(r);
        else { // :75-97
            float pi = r.p_target, pi_sum = r.p_target * (float)cur.M;
            for (int i = 0; i < NI; i++) {
                if (nq[i] == MQ_NIL) continue;
                Hit ns; load_chit(F.hits + 10 * (size_t)nq[i], ns);
                float spatial_p = restir_target_pdf(r, ns);
                if (R.spatial_bias_correction == 2 && spatial_p > 0.0f && !restir_trace_visibility(sc, ns.pos, r.pos, stk, spill, ctr)) spatial_p = 0.0f;
                pi = selected == i ? spatial_p : pi;
                pi_sum += spatial_p * (float)F.res_read[4 * (size_t)nq[i]].x; // read_reservoir(neighbors[i]).M
            }
            res_finalize_custom(r, pi, pi_sum);
        }
        res_store(F.res_a + 4 * p.idx, r);
    }
}

// restir_di_shade.comp:21-62
__global__ __launch_bounds__(MQ_BLOCK, MQ_OCC_SHADE) void mq_restir_shade_kernel(MqSceneDev sc, MqParams P, MqRestirParams R, MqRestirFrame F) {
    MQ_RESTIR_SETUP
    for (uint32_t tile = F.tile_begin + blockIdx.x * MQ_WAVES + (threadIdx.x >> 6); tile < F.tile_end; tile += n_waves) {
        const RestirPixel p = restir_pixel(F, tile, lane);
        if (!p.inside) continue;
        Reservoir r = res_load(F.res_a + 4 * p.idx);
        f3 irr = F3(0, 0, 0);
        if (r.flags & 1u) {
            Hit first; load_chit(F.hits + 10 * p.idx, first);
            const f3 dv = r.pos - first.pos;
            const f3 wo = normalize(dv);
            Hit next; next.wi = wo; next.pos = first.pos - first.wi * 1e-3f;
            f3 incident = F3(0, 0, 0), throughput = F3(1, 1, 1);
            restir_trace_ray(sc, P, U, throughput, incident, next, stk, spill, ctr);
            const float d_sample = length(dv), d_hit = length(first.pos - next.pos);
            if (R.visibility_shade && fabsf(d_sample - d_hit) / mmax(d_sample, d_hit) > 0.1f) { res_discard(r); res_store(F.res_a + 4 * p.idx, r); } // not visible
            const float bsdf = bsdf_times_wodotn(first.wi, wo, first.normal, roughness_to_alpha(first.roughness), 0.02f);
            if (mfinite(r.w)) irr = ((res_radiance(r) * bsdf) * r.w) * (mmax(dot(r.normal, -wo), 0.0f) / (d_sample * d_sample));
        }
        F.irradiance[p.idx] = make_float4(irr.x, irr.y, irr.z, 1.0f);
        const float l = luminance(irr);
        F.moments[p.idx] = make_float2(l, l * l);
    }
}

// restir_di_clear.comp:8-16
__global__ void mq_restir_clear_kernel(MqRestirFrame F) {
    const size_t n = (size_t)F.W * F.H;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        F.res_a[4 * i] = F.res_a[4 * i + 1] = F.res_a[4 * i + 2] = F.res_a[4 * i + 3] = make_uint4(0, 0, 0, 0);
        F.irradiance[i] = make_float4(0, 0, 0, 0); F.moments[i] = make_float2(0, 0);
    }
}

// resident blocks per CU of the four pass kernels (they are grid-stride loops over tiles: the right grid holds exactly
// the blocks the chip keeps resident; with 2 per CU instead of 3 / 4 the node took 17 % longer)
int mq_restir_resident_blocks(int out[4]) {
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[0], mq_restir_generate_kernel, MQ_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[1], mq_restir_temporal_kernel, MQ_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[2], mq_restir_spatial_kernel, MQ_BLOCK, 0);
    if (e == hipSuccess) e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&out[3], mq_restir_shade_kernel, MQ_BLOCK, 0);
    return (int)e;
}
// the wavefront halves: which = 0 generate_a, 1 generate_b, 2 shade_a, 3 shade_b; `round` = the queue round they use
int mq_launch_restir_wavefront(const MqSceneDev& sc, const MqParams& P, const MqRestirParams& R, const MqRestirFrame& F, const MqFrame& FQ, int which, int round, int grid, hipStream_t s) {
    switch (which) {
    case 0: mq_restir_generate_a_kernel<<<grid, MQ_BLOCK, 0, s>>>(P, R, F, FQ, round); break;
    case 1: mq_restir_generate_b_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F, FQ, round); break;
    case 2: mq_restir_shade_a_kernel<<<grid, MQ_BLOCK, 0, s>>>(R, F, FQ, round); break;
    default: mq_restir_shade_b_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F, FQ, round); break;
    }
    return (int)hipGetLastError();
}
int mq_launch_restir(const MqSceneDev& sc, const MqParams& P, const MqRestirParams& R, const MqRestirFrame& F, int pass, int grid, hipStream_t s) {
    switch (pass) {
    case 0: mq_restir_generate_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F); break;
    case 1: mq_restir_temporal_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F); break;
    case 2: mq_restir_spatial_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F); break;
    case 3: mq_restir_shade_kernel<<<grid, MQ_BLOCK, 0, s>>>(sc, P, R, F); break;
    default: mq_restir_clear_kernel<<<1024, 256, 0, s>>>(F); break;
    }
    return (int)hipGetLastError();
}
