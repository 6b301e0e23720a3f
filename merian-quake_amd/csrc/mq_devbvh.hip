// mq_devbvh.hip -- the tree of the PER-FRAME geometry built on the device.
//
// The reference hands its per-frame geometry (entities, particles: quake_node.cpp:896-983) to the Vulkan driver, which builds
// the acceleration structure on the GPU every frame.  The host builder of this library (mq_bvh.cpp: binned SAH, collapse to
// 8-wide nodes on a worker pool) keeps up with a device frame for some 10 k triangles per frame; beyond that the host is the
// bound of a frame.  This builder takes the flattened triangles and leaves everything else on the device:
//
//   bounds        centroid box + largest coordinate                      (wave reductions, ordered-integer atomics)
//   pairs         triangles 2k and 2k + 1 that share two vertices (the halves of a sprite's quad, two faces of a particle's tetrahedron,
//                 neighbours of a strip) become ONE primitive = one leaf record, as in the host builder; everything else is its own
//   codes         30-bit Morton code of every primitive's centre
//   sort          rocprim::radix_sort_pairs (code, primitive)
//   hierarchy     binary radix tree over the sorted codes (Karras 2012: one thread per internal node)
//   fit           boxes bottom-up: the second thread to arrive at a node merges its children (acq_rel counters at agent scope:
//                 the XCDs' L2s are not coherent with each other)
//   collapse      level by level from the root: a thread per wide node opens the internal child of the largest area until it
//                 has 8 children, assigns octant slots, quantises the child boxes conservatively and allocates its child block
//                 and its leaf records with two atomic counters -- the same 80-byte nodes and 64-byte leaf records the host
//                 builder writes, so the traversal kernels do not know who built a tree
//   records       leaf record (one or two triangles), triangles and shading records of every leaf, in leaf order
//
// Any closest-hit query returns the same (t, triangle) whatever tree it walks (ties go to the smaller key), so frames rendered
// on a device-built tree are bit-identical to frames on a host-built one (tests/test_gpu_devbvh.py).
#include "mq_device.h"
#include "mq_devbvh.h"

#include <algorithm>
#include <cstring>
#include <rocprim/device/device_radix_sort.hpp>

namespace {
__device__ __forceinline__ uint32_t ord(float f) { const uint32_t b = __float_as_uint(f); return (b & 0x80000000u) ? ~b : (b | 0x80000000u); } // order-preserving
__device__ __forceinline__ float unord(uint32_t k) { return __uint_as_float((k & 0x80000000u) ? (k & 0x7fffffffu) : ~k); }
__device__ __forceinline__ float exp2i(int e) { return __uint_as_float((uint32_t)(e + 127) << 23); }
__device__ __forceinline__ bool is_leaf(int m, int id) { return id >= m - 1; } // m primitives: ids [0, m - 1) internal, [m - 1, 2m - 1) leaves
#define MQ_DB_PAIRED 0x80000000u // primitive descriptor: first triangle | this flag when triangle + 1 belongs to it too
__device__ __forceinline__ bool same_vtx(const float* a, const float* b) { return __float_as_uint(a[0]) == __float_as_uint(b[0]) && __float_as_uint(a[1]) == __float_as_uint(b[1]) && __float_as_uint(a[2]) == __float_as_uint(b[2]); }
// vertex selectors of B against A's vertices (MqLeafRec::sel), -1 if B does not share two vertices with A; `fresh`: B's vertex that is not A's (-1: none)
__device__ __forceinline__ int pair_sel(const MqTri& A, const MqTri& B, int& fresh) {
    const float* av[3] = {A.v0, A.v1, A.v2}; const float* bv[3] = {B.v0, B.v1, B.v2};
    int sel[3], n_shared = 0; fresh = -1;
    for (int k = 0; k < 3; k++) {
        sel[k] = -1;
        for (int m = 0; m < 3; m++) if (same_vtx(bv[k], av[m])) { sel[k] = m; break; }
        if (sel[k] >= 0) n_shared++; else fresh = k;
    }
    if (n_shared < 2) return -1;
    if (fresh >= 0) sel[fresh] = 3;
    return sel[0] | (sel[1] << 2) | (sel[2] << 4);
}

__global__ void db_init(MqDevBvh A) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < MQ_DB_WORDS) {
        uint32_t v = 0;
        if (i == MQ_DB_Q0) v = 1u;          // the root waits in queue 0
        if (i == MQ_DB_NODES) v = 1u;       // node 0 of the region is the root
        if (i >= MQ_DB_LO && i < MQ_DB_LO + 3) v = 0xffffffffu;
        A.ctr[i] = v;
    }
    if (i == 0) { A.queue0[0] = make_uint2(0u, 0u); A.parent[0] = -1; }
}

__global__ __launch_bounds__(256) void db_bounds(MqDevBvh A) {
    float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY}, mx = 1.0f;
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
        const MqTri t = A.in[i];
        for (int a = 0; a < 3; a++) {
            const float l = fminf(t.v0[a], fminf(t.v1[a], t.v2[a])), h = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a]));
            const float c = 0.5f * (l + h);
            lo[a] = fminf(lo[a], c); hi[a] = fmaxf(hi[a], c);
            mx = fmaxf(mx, fmaxf(fabsf(l), fabsf(h)));
        }
    }
    for (int off = 32; off > 0; off >>= 1) {
        for (int a = 0; a < 3; a++) { lo[a] = fminf(lo[a], __shfl_down(lo[a], off, 64)); hi[a] = fmaxf(hi[a], __shfl_down(hi[a], off, 64)); }
        mx = fmaxf(mx, __shfl_down(mx, off, 64));
    }
    if ((threadIdx.x & 63) == 0) {
        for (int a = 0; a < 3; a++) { atomicMin(&A.ctr[MQ_DB_LO + a], ord(lo[a])); atomicMax(&A.ctr[MQ_DB_HI + a], ord(hi[a])); }
        atomicMax(&A.ctr[MQ_DB_MAXABS], __float_as_uint(mx)); // (positive floats order like their bits)
    }
}

__device__ __forceinline__ uint32_t spread10(uint32_t x) { x &= 0x3ffu; x = (x | (x << 16)) & 0x030000ffu; x = (x | (x << 8)) & 0x0300f00fu; x = (x | (x << 4)) & 0x030c30c3u; x = (x | (x << 2)) & 0x09249249u; return x; }
// primitives: (2k, 2k + 1) as one if they share two vertices, else one each; their order in the list is whatever the atomics make it (the sort follows)
__global__ __launch_bounds__(256) void db_pairs(MqDevBvh A) {
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; 2u * k < A.n; k += gridDim.x * blockDim.x) {
        const uint32_t t0 = 2u * k, t1 = t0 + 1u;
        int fresh;
        if (t1 < A.n && pair_sel(A.in[t0], A.in[t1], fresh) >= 0) A.vals0[atomicAdd(&A.ctr[MQ_DB_PRIMS], 1u)] = t0 | MQ_DB_PAIRED;
        else if (t1 < A.n) { const uint32_t at = atomicAdd(&A.ctr[MQ_DB_PRIMS], 2u); A.vals0[at] = t0; A.vals0[at + 1u] = t1; }
        else A.vals0[atomicAdd(&A.ctr[MQ_DB_PRIMS], 1u)] = t0;
    }
}
__device__ __forceinline__ void prim_box(const MqDevBvh& A, uint32_t desc, float b[6]) {
    const MqTri t = A.in[desc & ~MQ_DB_PAIRED];
    for (int a = 0; a < 3; a++) { b[a] = fminf(t.v0[a], fminf(t.v1[a], t.v2[a])); b[3 + a] = fmaxf(t.v0[a], fmaxf(t.v1[a], t.v2[a])); }
    if (desc & MQ_DB_PAIRED) {
        const MqTri u = A.in[(desc & ~MQ_DB_PAIRED) + 1u];
        for (int a = 0; a < 3; a++) { b[a] = fminf(b[a], fminf(u.v0[a], fminf(u.v1[a], u.v2[a]))); b[3 + a] = fmaxf(b[3 + a], fmaxf(u.v0[a], fmaxf(u.v1[a], u.v2[a]))); }
    }
}
__global__ __launch_bounds__(256) void db_codes(MqDevBvh A) {
    const uint32_t m = A.ctr[MQ_DB_PRIMS];
    float lo[3], inv[3];
    for (int a = 0; a < 3; a++) { lo[a] = unord(A.ctr[MQ_DB_LO + a]); const float e = unord(A.ctr[MQ_DB_HI + a]) - lo[a]; inv[a] = e > 0.0f ? 1024.0f / e : 0.0f; }
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < A.n; i += gridDim.x * blockDim.x) {
        if (i >= m) { A.keys0[i] = 0xffffffffu; A.vals0[i] = 0xffffffffu; continue; } // (the sort runs over n entries, whatever m turned out to be: the rest goes to the end)
        float b[6]; prim_box(A, A.vals0[i], b);
        uint32_t q[3];
        for (int a = 0; a < 3; a++) { const float f = (0.5f * (b[a] + b[3 + a]) - lo[a]) * inv[a]; q[a] = (uint32_t)fminf(fmaxf(f, 0.0f), 1023.0f); } // (NaN -> 0)
        A.keys0[i] = spread10(q[0]) | (spread10(q[1]) << 1) | (spread10(q[2]) << 2);
    }
}

// length of the common prefix of the codes at sorted positions i and j; equal codes are told apart by their positions
__device__ __forceinline__ int delta(const uint32_t* keys, int n, int i, int j) {
    if (j < 0 || j >= n) return -1;
    const uint32_t a = keys[i], b = keys[j];
    return a == b ? 32 + __clz((uint32_t)(i ^ j)) : __clz(a ^ b);
}
__global__ __launch_bounds__(256) void db_hierarchy(MqDevBvh A) {
    const int n = (int)A.ctr[MQ_DB_PRIMS];
    const uint32_t* keys = A.keys1;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n - 1; i += gridDim.x * blockDim.x) {
        const int d = delta(keys, n, i, i + 1) - delta(keys, n, i, i - 1) >= 0 ? 1 : -1;
        const int dmin = delta(keys, n, i, i - d);
        int lmax = 2;
        while (delta(keys, n, i, i + lmax * d) > dmin) lmax *= 2; // (ends: beyond the array delta is -1)
        int l = 0;
        for (int t = lmax >> 1; t >= 1; t >>= 1) if (delta(keys, n, i, i + (l + t) * d) > dmin) l += t;
        const int j = i + l * d;
        const int dnode = delta(keys, n, i, j);
        int s = 0, t = l;
        do { t = (t + 1) >> 1; if (delta(keys, n, i, i + (s + t) * d) > dnode) s += t; } while (t > 1);
        const int gamma = i + s * d + (d < 0 ? d : 0);
        const int lo_ = i < j ? i : j, hi_ = i < j ? j : i;
        const int left = lo_ == gamma ? n - 1 + gamma : gamma, right = hi_ == gamma + 1 ? n - 1 + gamma + 1 : gamma + 1;
        A.child[i] = make_int2(left, right);
        A.parent[left] = i; A.parent[right] = i;
    }
}

__global__ __launch_bounds__(256) void db_fit(MqDevBvh A) {
    const int n = (int)A.ctr[MQ_DB_PRIMS];
    for (int k = blockIdx.x * blockDim.x + threadIdx.x; k < n; k += gridDim.x * blockDim.x) {
        float b[6]; prim_box(A, A.vals1[k], b);
        int id = n - 1 + k;
        for (int a = 0; a < 6; a++) A.box[6 * (size_t)id + a] = b[a];
        int p = n > 1 ? A.parent[id] : -1;
        while (p >= 0) {
            // the first thread to arrive leaves; the second one has both children's boxes behind an acquire (agent scope: the other child may have been written through another XCD's L2)
            if (__hip_atomic_fetch_add(&A.flag[p], 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT) == 0u) break;
            const int2 c = A.child[p];
            for (int a = 0; a < 3; a++) {
                b[a] = fminf(__hip_atomic_load(&A.box[6 * (size_t)c.x + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&A.box[6 * (size_t)c.y + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
                b[3 + a] = fmaxf(__hip_atomic_load(&A.box[6 * (size_t)c.x + 3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(&A.box[6 * (size_t)c.y + 3 + a], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
            }
            for (int a = 0; a < 6; a++) A.box[6 * (size_t)p + a] = b[a];
            id = p; p = A.parent[p];
        }
    }
}

__device__ __forceinline__ float box_area(const float* b) {
    const float dx = b[3] - b[0], dy = b[4] - b[1], dz = b[5] - b[2];
    return dx < 0.0f ? 0.0f : 2.0f * (dx * dy + dy * dz + dz * dx);
}

__device__ __forceinline__ MqShadeRec shade_of(const MqDevBvh& A, uint32_t key) { // as shade_records() of mq_api.cpp
    MqShadeRec q; memset(&q, 0, sizeof q);
    const mq_ext* xp = A.sc.geo[key >> 28].ext;
    if (xp) {
        const mq_ext x = xp[key & 0x0fffffffu];
        memcpy(q.ext, &x, 28);
        const uint32_t ta = x.texnum_alpha & 0xfffu, tf = x.texnum_fb_flags & 0xfffu;
        q.albedo = A.sc.tex[ta < MQ_MAX_GLTEXTURES - 1 ? ta : MQ_MAX_GLTEXTURES - 1];
        if (tf < MQ_MAX_GLTEXTURES) q.fb = A.sc.tex[tf]; else q.fb.offset = MQ_NIL;
    }
    return q;
}

// One level of the collapse: every entry of this level's queue becomes an 80-byte node (mq_bvh.cpp, "collapse to 8-wide", is the
// host's version of the same steps); its internal children are queued for the next level.
__global__ __launch_bounds__(128) void db_collapse(MqDevBvh A, int level) {
    const uint32_t n_in = A.ctr[MQ_DB_Q0 + level % 3];
    const int m = (int)A.ctr[MQ_DB_PRIMS];
    uint32_t* n_out = &A.ctr[MQ_DB_Q0 + (level + 1) % 3];
    if (blockIdx.x == 0 && threadIdx.x == 0) { A.ctr[MQ_DB_Q0 + (level + 2) % 3] = 0u; if (n_in) atomicMax(&A.ctr[MQ_DB_DEPTH], (uint32_t)level + 1u); } // (last level's input: nobody reads it any more; it is the next level's output)
    const uint2* qin = (level & 1) ? A.queue1 : A.queue0;
    uint2* qout = (level & 1) ? A.queue0 : A.queue1;
    const float pad = fmaxf(1e-4f, __uint_as_float(A.ctr[MQ_DB_MAXABS]) * 4.76837158203125e-07f); // 2^-21 of the largest coordinate, as the host builder pads
    for (uint32_t w = blockIdx.x * blockDim.x + threadIdx.x; w < n_in; w += gridDim.x * blockDim.x) {
        const int bnode = (int)qin[w].x; const uint32_t out_index = qin[w].y;
        int ch[8]; int nc = 0;
        if (is_leaf(m, bnode)) ch[nc++] = bnode; // a tree of one primitive
        else { const int2 c = A.child[bnode]; ch[nc++] = c.x; ch[nc++] = c.y; }
        while (nc < 8) { // open the internal child of the largest area
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nc; i++) if (!is_leaf(m, ch[i])) { const float a = box_area(A.box + 6 * (size_t)ch[i]); if (a > ba) { ba = a; best = i; } }
            if (best < 0) break;
            const int2 c = A.child[ch[best]];
            ch[best] = c.x; ch[nc++] = c.y;
        }
        float pb[6];
        for (int a = 0; a < 6; a++) pb[a] = A.box[6 * (size_t)bnode + a];
        // octant-aware slots: slot bit k set = the child lies on the high side of axis k; greedy by the best (child, slot) pair
        const float pc[3] = {0.5f * (pb[0] + pb[3]), 0.5f * (pb[1] + pb[4]), 0.5f * (pb[2] + pb[5])};
        float dcen[8][3];
        for (int i = 0; i < nc; i++) { const float* cb = A.box + 6 * (size_t)ch[i]; for (int a = 0; a < 3; a++) dcen[i][a] = 0.5f * (cb[a] + cb[3 + a]) - pc[a]; }
        // a child takes the slot of its octant (slot bit k set = it lies on the high side of axis k) or, if that one is taken, the next free one: the
        // children farthest from the parent's centre choose first.  (The host builder runs the best-pair auction of the CWBVH paper, 512 evaluations per
        // node; one thread per node cannot afford that on the device: a level took 0.1 ms.)
        int slot_child[8]; for (int s = 0; s < 8; s++) slot_child[s] = -1;
        uint32_t used = 0;
        for (int it = 0; it < nc; it++) {
            int bi = 0; float bd = -1.0f;
            for (int i = 0; i < nc; i++) if (!((used >> i) & 1u)) { const float d = fabsf(dcen[i][0]) + fabsf(dcen[i][1]) + fabsf(dcen[i][2]); if (d > bd || bd < 0.0f) { bd = d; bi = i; } }
            used |= 1u << bi;
            int sl = (dcen[bi][0] > 0.0f ? 1 : 0) | (dcen[bi][1] > 0.0f ? 2 : 0) | (dcen[bi][2] > 0.0f ? 4 : 0);
            while (slot_child[sl] >= 0) sl = (sl + 1) & 7;
            slot_child[sl] = ch[bi];
        }
        MqNode node; memset(&node, 0, sizeof node);
        float lo[3], hi[3]; int ex[3];
        for (int a = 0; a < 3; a++) {
            lo[a] = pb[a] - pad; hi[a] = pb[3 + a] + pad;
            const float ext = fmaxf(hi[a] - lo[a], 1e-30f);
            int e = (int)ceilf(log2f(ext / 255.0f));
            e = e < -126 ? -126 : (e > 127 ? 127 : e);
            while (e < 127 && lo[a] + 255.0f * exp2i(e) < hi[a]) e++; // 255 steps of 2^e must cover the extent in float arithmetic
            ex[a] = e;
        }
        node.px = lo[0]; node.py = lo[1]; node.pz = lo[2];
        node.ex = (uint8_t)(ex[0] + 127); node.ey = (uint8_t)(ex[1] + 127); node.ez = (uint8_t)(ex[2] + 127);
        uint32_t n_internal = 0, n_leaf = 0, n_leaf_tris = 0;
        for (int s = 0; s < 8; s++) if (slot_child[s] >= 0) {
            if (is_leaf(m, slot_child[s])) { n_leaf++; n_leaf_tris += (A.vals1[slot_child[s] - (m - 1)] & MQ_DB_PAIRED) ? 2u : 1u; } else n_internal++;
        }
        const uint32_t child_rel = n_internal ? atomicAdd(&A.ctr[MQ_DB_NODES], n_internal) : 0u;
        const uint32_t leaf_rel = n_leaf ? atomicAdd(&A.ctr[MQ_DB_LEAVES], n_leaf) : 0u;
        const uint32_t tris_rel = n_leaf ? atomicAdd(&A.ctr[MQ_DB_TRIS], n_leaf_tris) : 0u;
        const bool room = child_rel + n_internal <= A.node_cap;
        if (!room) atomicOr(&A.ctr[MQ_DB_ERR], 1u); // cannot happen (a tree of n triangles has fewer than n wide nodes; the region holds n): the node keeps no children
        node.child_base = A.node_base + child_rel;
        node.tri_base = A.leaf_base + leaf_rel;
        uint8_t* qlo[3] = {node.qlox, node.qloy, node.qloz};
        uint8_t* qhi[3] = {node.qhix, node.qhiy, node.qhiz};
        uint32_t next_child = 0, rec_off = 0, tri_off = 0;
        const uint32_t q_at = (room && n_internal) ? atomicAdd(n_out, n_internal) : 0u;
        for (int s = 0; s < 8; s++) {
            const int c = slot_child[s];
            if (c < 0 || !room) { node.meta[s] = 0; for (int a = 0; a < 3; a++) { qlo[a][s] = 255; qhi[a][s] = 0; } continue; } // empty: inverted box, never hit
            const float* cb = A.box + 6 * (size_t)c;
            for (int a = 0; a < 3; a++) {
                const float e = exp2i(ex[a]);
                const float clo = cb[a] - pad, chi = cb[3 + a] + pad;
                int ql = (int)floorf((clo - lo[a]) / e), qh = (int)ceilf((chi - lo[a]) / e);
                ql = ql < 0 ? 0 : (ql > 255 ? 255 : ql); qh = qh < 0 ? 0 : (qh > 255 ? 255 : qh);
                while (ql > 0 && lo[a] + (float)ql * e > clo) ql--; // a decoded plane must not cut into the child
                while (qh < 255 && lo[a] + (float)qh * e < chi) qh++;
                qlo[a][s] = (uint8_t)ql; qhi[a][s] = (uint8_t)qh;
            }
            if (!is_leaf(m, c)) {
                node.imask |= (uint8_t)(1u << s);
                node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
                qout[q_at + next_child] = make_uint2((uint32_t)c, child_rel + next_child);
                next_child++;
            } else { // one leaf record of one or two triangles: the collapse only says WHERE (db_records writes them, a thread per primitive)
                const uint32_t desc = A.vals1[c - (m - 1)];
                A.leaf_at[c - (m - 1)] = make_uint2(A.leaf_base + leaf_rel + rec_off, A.tri_base + tris_rel + tri_off);
                tri_off += (desc & MQ_DB_PAIRED) ? 2u : 1u;
                node.meta[s] = (uint8_t)((1u << 5) | rec_off);
                rec_off++;
            }
        }
        if (out_index < A.node_cap) A.nodes[A.node_base + out_index] = node;
    }
}

// leaf record, triangles and shading records of every primitive, where the collapse put them
__global__ __launch_bounds__(256) void db_records(MqDevBvh A) {
    const uint32_t m = A.ctr[MQ_DB_PRIMS];
    for (uint32_t k = blockIdx.x * blockDim.x + threadIdx.x; k < m; k += gridDim.x * blockDim.x) {
        const uint2 at = A.leaf_at[k];
        if (at.x == MQ_NIL) continue; // (a leaf the collapse never reached: only in a tree deeper than MQ_DB_LEVELS, which is flagged)
        const uint32_t desc = A.vals1[k], ti = at.y;
        const MqTri t = A.in[desc & ~MQ_DB_PAIRED];
        MqLeafRec r; memset(&r, 0, sizeof r);
        for (int a = 0; a < 3; a++) { r.v[0][a] = t.v0[a]; r.v[1][a] = t.v1[a]; r.v[2][a] = t.v2[a]; r.v[3][a] = t.v0[a]; }
        r.key0 = t.key; r.key1 = MQ_NIL; r.tri0 = ti; r.sel = (t.flags & MQ_TRI_ANYHIT) ? 0x10000u : 0u;
        A.tris[ti] = t;
        A.shade[ti] = shade_of(A, t.key);
        if (desc & MQ_DB_PAIRED) {
            const MqTri u = A.in[(desc & ~MQ_DB_PAIRED) + 1u];
            int fresh; const int sel = pair_sel(t, u, fresh);
            if (fresh >= 0) { const float* bv[3] = {u.v0, u.v1, u.v2}; for (int a = 0; a < 3; a++) r.v[3][a] = bv[fresh][a]; }
            r.key1 = u.key;
            r.sel |= (uint32_t)sel | MQ_LEAF_HAS_B | ((u.flags & MQ_TRI_ANYHIT) ? 0x20000u : 0u);
            A.tris[ti + 1u] = u;
            A.shade[ti + 1u] = shade_of(A, u.key);
        }
        A.leaves[at.x] = r;
    }
}

__global__ void db_finish(MqDevBvh A) { // anything left in the queue after the last level: a tree deeper than MQ_DB_LEVELS (its unexpanded nodes are zeros: no children)
    if (blockIdx.x == 0 && threadIdx.x == 0 && A.ctr[MQ_DB_Q0 + MQ_DB_LEVELS % 3] != 0u) atomicOr(&A.ctr[MQ_DB_ERR], 2u);
}
} // namespace

size_t mq_device_bvh_sort_bytes(uint32_t n) {
    size_t bytes = 0;
    uint32_t* k = nullptr;
    (void)rocprim::radix_sort_pairs(nullptr, bytes, k, k, k, k, (size_t)n, 0u, 32u, (hipStream_t) nullptr);
    return bytes;
}

// Enqueues the whole build on `s`.  A.n >= 1; the node region [node_base, node_base + node_cap) must hold A.n nodes, the leaf / triangle /
// shading-record regions A.n entries.  The node region is zeroed first: a node nobody wrote has no children.
int mq_launch_device_bvh(const MqDevBvh& A, void* sort_tmp, size_t sort_bytes, hipStream_t s) {
    if (A.n == 0) return 0;
    const int grid = (int)std::min<uint32_t>((A.n + 255u) / 256u, 2048u);
    hipError_t e = hipMemsetAsync(A.nodes + A.node_base, 0, (size_t)std::min(A.n, A.node_cap) * sizeof(MqNode), s);
    if (e == hipSuccess && A.n > 1) e = hipMemsetAsync(A.flag, 0, (size_t)(A.n - 1) * 4, s);
    if (e == hipSuccess) e = hipMemsetAsync(A.leaf_at, 0xff, (size_t)A.n * 8, s);
    if (e != hipSuccess) return (int)e;
    db_init<<<1, 64, 0, s>>>(A);
    db_bounds<<<grid, 256, 0, s>>>(A);
    db_pairs<<<grid, 256, 0, s>>>(A);
    db_codes<<<grid, 256, 0, s>>>(A);
    e = rocprim::radix_sort_pairs(sort_tmp, sort_bytes, A.keys0, A.keys1, A.vals0, A.vals1, (size_t)A.n, 0u, 32u, s); // (n entries: the unused ones carry the largest key)
    if (e != hipSuccess) return (int)e;
    if (A.n > 1) db_hierarchy<<<grid, 256, 0, s>>>(A);
    db_fit<<<grid, 256, 0, s>>>(A);
    for (int level = 0; level < MQ_DB_LEVELS; level++) db_collapse<<<(int)std::min<uint32_t>((A.n + 127u) / 128u, 1024u), 128, 0, s>>>(A, level);
    db_records<<<grid, 256, 0, s>>>(A);
    db_finish<<<1, 64, 0, s>>>(A);
    return (int)hipGetLastError();
}
