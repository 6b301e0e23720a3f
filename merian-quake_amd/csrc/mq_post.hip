// mq_post.hip -- the step right after the path: temporal accumulation of the radiance images
// with motion-vector reprojection, then albedo re-modulation and composition,
//
//     final = accum(irradiance) * albedo + accum(volume) + first-hit emission,
//
// i.e. the `accum` / `volume accum` (merian "Accumulate") nodes and the `add` node of the reference's graph with the
// albedo re-modulation its denoiser node performs in between (res/default_config.json:21-133,404-435,473-497; wiring:
// accum.src <- render_markovchain.irradiance, accum.mv <- gbuffer.mv, accum.gbuffer / prev_gbuffer <- gbuffer.gbuffer
// (delay 1), accum.prev_out / prev_history <- accum.out / history (delay 1); volume accum likewise with
// render_markovchain.volume / volume_mv; add.input_0..2 <- volume denoiser.out, denoiser.out, gbuffer.irradiance).
//
// merian's node sources are NOT in the reference tree (empty submodule), so the arithmetic is DEFINED here, by the
// property names of the shipped configuration ("post chain"):
//   reprojection   q = round(p + mv(p)) (mv = previous pixel - pixel, gbuffer.comp:111-115); "enable motion vectors"
//                  off: q = p; outside the image: rejected, or clamped to the border with "reuse border"
//   validation     normals: dot(n(p), n_prev(q)) >= cos("normal threshold");
//                  depth: |z(p) + vel_z(p) - z_prev(q)| <= "depth threshold" * max(z(p) + vel_z(p), z_prev(q))
//                  (linear_z and vel_z of the g-buffer, gbuffer.comp:123-130)
//   history        h = valid ? min(prev_history(q) + 1, "max history") : 1
//   blend          out = mix(prev_out(q), src, max(1 / h, 1 - "alpha")) on all four channels (rgb mean radiance, a = mean
//                  luminance^2); alpha = 1 gives the running mean the reference's convergence plots use (scripts/error_plot.py)
// Not taken over: the firefly filter, adaptive alpha ("adaptivity ..."), the stochastic bilinear / extended-search
// reprojection filters and the SVGF spatial filter (the shipped configuration runs the denoiser with "filter": "none").
#include "mq_device.h"

struct MqAccumParams {
    float alpha, max_history, cos_normal_threshold, depth_threshold;
    int32_t enable_mv, reuse_border;
};

__global__ __launch_bounds__(256) void mq_accumulate_kernel(MqAccumParams A, uint32_t W, uint32_t H, const float4* __restrict__ src, const uint32_t* __restrict__ mv,
                                                            const uint4* __restrict__ gb, const uint4* __restrict__ prev_gb, const float4* __restrict__ prev_out,
                                                            const float* __restrict__ prev_hist, float4* __restrict__ out, float* __restrict__ hist, int first,
                                                            uint32_t row_begin, uint32_t row_end, uint32_t row_lo, uint32_t row_hi, uint32_t* flags) {
    // rows [row_begin, row_end): what this rank accumulates (the whole image on one rank); [row_lo, row_hi): the rows whose
    // previous-frame state it holds -- a reprojected pixel beyond them is flagged (overflow bit 3) and starts a new history
    const size_t n = (size_t)W * row_end;
    for (size_t i = (size_t)W * row_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t ix = (uint32_t)(i % W), iy = (uint32_t)(i / W);
        const float4 s = src[i];
        float h = 1.0f;
        float4 o = s;
        if (!first) {
            float mx = 0.0f, my = 0.0f;
            if (A.enable_mv) { const uint32_t m = mv[i]; mx = h2f((uint16_t)(m & 0xffffu)); my = h2f((uint16_t)(m >> 16)); }
            float qx = floorf(((float)ix + mx) + 0.5f), qy = floorf(((float)iy + my) + 0.5f);
            bool valid = qx >= 0.0f && qy >= 0.0f && qx < (float)W && qy < (float)H; // NaN motion vectors fail here
            if (!valid && A.reuse_border && qx == qx && qy == qy) { qx = mclamp(qx, 0.0f, (float)W - 1.0f); qy = mclamp(qy, 0.0f, (float)H - 1.0f); valid = true; }
            if (valid && !((uint32_t)qy >= row_lo && (uint32_t)qy < row_hi)) { atomicOr(flags, 8u); valid = false; }
            if (valid) {
                const size_t q = (size_t)(uint32_t)qy * W + (uint32_t)qx;
                const uint4 g = gb[i], pg = prev_gb[q];
                const float ze = __uint_as_float(g.y) + __uint_as_float(g.w), zp = __uint_as_float(pg.y);
                valid = dot(decode_normal(g.x), decode_normal(pg.x)) >= A.cos_normal_threshold && fabsf(ze - zp) <= A.depth_threshold * mmax(ze, zp);
                if (valid) {
                    h = mmin(prev_hist[q] + 1.0f, A.max_history);
                    const float a = mmax(1.0f / h, 1.0f - A.alpha);
                    const float4 p = prev_out[q];
                    o = make_float4(mmix(p.x, s.x, a), mmix(p.y, s.y, a), mmix(p.z, s.z, a), mmix(p.w, s.w, a));
                }
            }
        }
        out[i] = o; hist[i] = h;
    }
}

// final = accum * albedo + volume accum + first-hit emission (alpha = 1); with `direct` (property "add: restir irradiance",
// BASELINE config 5 "ReSTIR DI + MCPG GI combined": one more input of the graph's `add` node): + direct * albedo
__global__ __launch_bounds__(256) void mq_compose_kernel(uint32_t W, uint32_t row_begin, uint32_t row_end, const float4* __restrict__ accum, const uint2* __restrict__ albedo, const float4* __restrict__ vol,
                                                         const uint2* __restrict__ emission, const float4* __restrict__ direct, float4* __restrict__ final_out) {
    const size_t n = (size_t)W * row_end;
    for (size_t i = (size_t)W * row_begin + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const float4 a = accum[i], v = vol[i];
        const uint2 al = albedo[i], em = emission[i];
        const float ar = h2f((uint16_t)(al.x & 0xffffu)), ag = h2f((uint16_t)(al.x >> 16)), ab = h2f((uint16_t)(al.y & 0xffffu));
        float4 o = make_float4((a.x * ar + v.x) + h2f((uint16_t)(em.x & 0xffffu)), (a.y * ag + v.y) + h2f((uint16_t)(em.x >> 16)), (a.z * ab + v.z) + h2f((uint16_t)(em.y & 0xffffu)), 1.0f);
        if (direct) { const float4 d = direct[i]; o.x = o.x + d.x * ar; o.y = o.y + d.y * ag; o.z = o.z + d.z * ab; }
        final_out[i] = o;
    }
}

int mq_launch_accumulate(const float* accum_params6, uint32_t W, uint32_t H, const void* src, const void* mv, const void* gb, const void* prev_gb, const void* prev_out, const void* prev_hist,
                         void* out, void* hist, int first, const uint32_t rows[4], uint32_t* flags, hipStream_t s) {
    MqAccumParams A;
    A.alpha = accum_params6[0]; A.max_history = accum_params6[1]; A.cos_normal_threshold = accum_params6[2]; A.depth_threshold = accum_params6[3];
    A.enable_mv = accum_params6[4] != 0.0f; A.reuse_border = accum_params6[5] != 0.0f;
    mq_accumulate_kernel<<<2048, 256, 0, s>>>(A, W, H, (const float4*)src, (const uint32_t*)mv, (const uint4*)gb, (const uint4*)prev_gb, (const float4*)prev_out, (const float*)prev_hist,
                                              (float4*)out, (float*)hist, first, rows[0], rows[1], rows[2], rows[3], flags);
    return (int)hipGetLastError();
}
int mq_launch_compose(uint32_t W, uint32_t row_begin, uint32_t row_end, const void* accum, const void* albedo, const void* vol, const void* emission, const void* direct, void* final_out, hipStream_t s) {
    mq_compose_kernel<<<2048, 256, 0, s>>>(W, row_begin, row_end, (const float4*)accum, (const uint2*)albedo, (const float4*)vol, (const uint2*)emission, (const float4*)direct, (float4*)final_out);
    return (int)hipGetLastError();
}
