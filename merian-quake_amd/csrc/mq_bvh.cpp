// mq_bvh.cpp -- host builder of the 8-wide compressed BVH consumed by mq_kernels.hip.
//
// Replaces merian's "Acceleration Structure Builder" node (res/default_config.json:3-20,400-403),
// which hands the geometry to the Vulkan driver.  Pipeline: binned-SAH binary BVH (leaves <= 3
// triangles) -> greedy collapse to 8 children by surface area -> octant-aware slot assignment ->
// 8-bit quantised child boxes (Ylitie, Karras, Laine 2017 node layout, 80 B/node).
// All vertices are world space (the reference applies transforms on the CPU,
// src/game/quake_helpers.cpp:410-417), so one BVH spans every geometry slot.
#include "mq_host.h"

#include <algorithm>
#include <functional>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <cstring>
#include <future>
#include <queue>
#include <deque>

namespace {

struct AABB {
    float lo[3], hi[3];
    void reset() { for (int a = 0; a < 3; a++) { lo[a] = INFINITY; hi[a] = -INFINITY; } }
    void grow(const AABB& o) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], o.lo[a]); hi[a] = std::max(hi[a], o.hi[a]); } }
    void grow(const float* p) { for (int a = 0; a < 3; a++) { lo[a] = std::min(lo[a], p[a]); hi[a] = std::max(hi[a], p[a]); } }
    float area() const {
        float d[3] = {hi[0] - lo[0], hi[1] - lo[1], hi[2] - lo[2]};
        if (d[0] < 0) return 0.0f;
        return 2.0f * (d[0] * d[1] + d[1] * d[2] + d[2] * d[0]);
    }
};

// most triangles per leaf of the binary tree (a compressed child slot addresses up to 3)
#ifndef MQ_BVH_LEAF
#define MQ_BVH_LEAF 3
#endif
struct BNode { AABB box; int left = -1, right = -1; uint32_t first = 0, count = 0 /* leaves only */, ntris = 0 /* below this node */; };

// A small pool of worker threads for the builder's subtree tasks (created on first use, lives as long as the process).  The
// thread that builds helps until its own tasks are done, so a build never waits for a worker that is busy elsewhere.
// Size: MQ_BVH_THREADS, else one per hardware thread up to 16 (a GPU box gives a job about 16 cores' worth of time).
struct TaskGroup { std::atomic<int> pending{0}; };
class TaskPool {
public:
    static TaskPool& get() { static TaskPool* p = new TaskPool; return *p; } // never destroyed: its threads sleep until the process ends (no join at exit, nothing to hang in a forked child)
    int threads() const { return (int)workers.size() + 1; }
    struct Busy { TaskPool& p; explicit Busy(TaskPool& pool) : p(pool) { p.active.fetch_add(1); } ~Busy() { p.active.fetch_sub(1); } };
    void spawn(TaskGroup& g, std::function<void()> f) {
        g.pending.fetch_add(1, std::memory_order_relaxed);
        { std::lock_guard<std::mutex> l(m); q.emplace_back(&g, std::move(f)); queued.fetch_add(1, std::memory_order_release); }
        cv.notify_one();
    }
    void wait(TaskGroup& g) { // run queued tasks (of any group) until this group has none left
        std::unique_lock<std::mutex> l(m);
        while (g.pending.load(std::memory_order_acquire) != 0) {
            if (!q.empty()) { run_one(l); continue; }
            done.wait_for(l, std::chrono::microseconds(200));
        }
    }
private:
    std::vector<std::thread> workers;
    std::mutex m; std::condition_variable cv, done;
    std::deque<std::pair<TaskGroup*, std::function<void()>>> q;
    bool stop = false;
    std::atomic<int> active{0}; // builds / parallel loops in progress: the workers do not go to sleep between their tasks (every wake-up is a system call on the thread that spawns)
    std::atomic<int> queued{0}; // mirrors q.size() for the workers' short spin before they sleep (a wake-up through the condition variable costs tens of microseconds, a per-frame build a few hundred)
    void run_one(std::unique_lock<std::mutex>& l) {
        auto t = std::move(q.front()); q.pop_front(); queued.fetch_sub(1, std::memory_order_relaxed);
        l.unlock();
        t.second();
        const bool last = t.first->pending.fetch_sub(1, std::memory_order_acq_rel) == 1;
        l.lock();
        if (last) done.notify_all();
    }
    TaskPool() {
        int n = getenv("MQ_BVH_THREADS") ? atoi(getenv("MQ_BVH_THREADS")) : (int)std::min(16u, std::max(1u, std::thread::hardware_concurrency()));
        for (int i = 1; i < n; i++) { workers.emplace_back([this] {
            std::unique_lock<std::mutex> l(m);
            for (;;) {
                if (q.empty() && !stop) {
                    l.unlock();
                    for (int i = 0; (i < 4000 || (active.load(std::memory_order_relaxed) > 0 && i < 2000000)) && queued.load(std::memory_order_acquire) == 0; i++) __builtin_ia32_pause();
                    l.lock();
                }
                cv.wait(l, [this] { return stop || !q.empty(); });
                if (stop) return;
                run_one(l);
            }
        }); workers.back().detach(); }
    }
};

struct Builder {
    const std::vector<MqTri>& in;
    std::vector<AABB>& tbox;
    std::vector<float>& cent; // 3 per tri
    std::vector<uint32_t>& order;
    std::vector<BNode>& nodes; // preallocated, 2 per triangle: a subtree over `count` triangles owns the ids [id, id + 2 * count - 1)
    TaskGroup group;
    uint32_t spawn_min = 0;   // subtrees of at least this many triangles become tasks of the pool (0: single-threaded build)

    Builder(const std::vector<MqTri>& t, std::vector<AABB>& tb, std::vector<float>& ce, std::vector<uint32_t>& od, std::vector<BNode>& nd) : in(t), tbox(tb), cent(ce), order(od), nodes(nd) {}

    // Binned SAH over the three axes, 16 bins, the three axes binned in ONE pass over the triangles.  The two halves of a split
    // touch disjoint ranges of `order` and their own node ids (known before they are built), so large halves are handed to the
    // pool; the tree does not depend on who built what.
    void build(uint32_t first, uint32_t count, int depth, int id) {
        for (;;) {
            AABB box; box.reset();
            AABB cbox; cbox.reset();
            const bool wide = spawn_min != 0 && count >= 16384; // the top of a large tree: the passes over its triangles run on the pool too (min / max / counts: the same bins in any order)
            if (wide) {
                std::mutex mm;
                mq_parallel_for(count, 4096, [&](size_t b0, size_t b1) {
                    AABB lb, lc; lb.reset(); lc.reset();
                    for (size_t i = first + b0; i < first + b1; i++) { lb.grow(tbox[order[i]]); lc.grow(&cent[3 * order[i]]); }
                    std::lock_guard<std::mutex> l(mm); box.grow(lb); cbox.grow(lc);
                });
            } else
            for (uint32_t i = first; i < first + count; i++) { box.grow(tbox[order[i]]); cbox.grow(&cent[3 * order[i]]); }
            nodes[id].box = box; nodes[id].ntris = count;
            if (count <= MQ_BVH_LEAF) { nodes[id].first = first; nodes[id].count = count; nodes[id].left = nodes[id].right = -1; return; }
            nodes[id].first = 0; nodes[id].count = 0; // (the node array is reused from build to build: every field is set)
            const int NB = 16;
            int best_axis = -1, best_split = 0; float best_cost = INFINITY;
            if (depth < 48) {
                AABB bb[3][NB]; uint32_t bc[3][NB];
                float scale[3]; bool use[3];
                for (int a = 0; a < 3; a++) {
                    const float ext = cbox.hi[a] - cbox.lo[a];
                    use[a] = ext > 0.0f; scale[a] = use[a] ? NB / ext : 0.0f;
                    for (int b = 0; b < NB; b++) { bb[a][b].reset(); bc[a][b] = 0; }
                }
                auto bin_range = [&](size_t i0, size_t i1, AABB (*xb)[NB], uint32_t (*xc)[NB]) {
                    for (size_t i = i0; i < i1; i++) {
                        const uint32_t t = order[i];
                        const AABB& tb = tbox[t];
                        for (int a = 0; a < 3; a++) if (use[a]) {
                            const int b = std::min(NB - 1, std::max(0, (int)((cent[3 * t + a] - cbox.lo[a]) * scale[a])));
                            xb[a][b].grow(tb); xc[a][b]++;
                        }
                    }
                };
                if (wide) {
                    std::mutex mm;
                    mq_parallel_for(count, 4096, [&](size_t b0, size_t b1) {
                        AABB lb[3][NB]; uint32_t lc[3][NB];
                        for (int a = 0; a < 3; a++) for (int b = 0; b < NB; b++) { lb[a][b].reset(); lc[a][b] = 0; }
                        bin_range(first + b0, first + b1, lb, lc);
                        std::lock_guard<std::mutex> l(mm);
                        for (int a = 0; a < 3; a++) for (int b = 0; b < NB; b++) { bb[a][b].grow(lb[a][b]); bc[a][b] += lc[a][b]; }
                    });
                } else bin_range(first, (size_t)first + count, bb, bc);
                for (int a = 0; a < 3; a++) {
                    if (!use[a]) continue;
                    float rarea[NB]; uint32_t rcount[NB];
                    AABB acc; acc.reset(); uint32_t cn = 0;
                    for (int b = NB - 1; b > 0; b--) { acc.grow(bb[a][b]); cn += bc[a][b]; rarea[b] = acc.area(); rcount[b] = cn; }
                    acc.reset(); cn = 0;
                    for (int b = 0; b < NB - 1; b++) {
                        acc.grow(bb[a][b]); cn += bc[a][b];
                        if (cn == 0 || rcount[b + 1] == 0) continue;
                        float cost = acc.area() * (float)cn + rarea[b + 1] * (float)rcount[b + 1];
                        if (cost < best_cost) { best_cost = cost; best_axis = a; best_split = b; }
                    }
                }
            }
            uint32_t mid;
            if (best_axis >= 0) {
                float ext = cbox.hi[best_axis] - cbox.lo[best_axis];
                float scale = NB / ext, lo = cbox.lo[best_axis];
                int a = best_axis, sp = best_split;
                auto it = std::partition(order.begin() + first, order.begin() + first + count, [&](uint32_t t) {
                    int b = std::min(NB - 1, std::max(0, (int)((cent[3 * t + a] - lo) * scale)));
                    return b <= sp;
                });
                mid = (uint32_t)(it - order.begin());
            } else mid = first; // force the median split below
            if (mid == first || mid == first + count) { // degenerate: median split on the widest axis
                int a = 0; float e0 = cbox.hi[0] - cbox.lo[0], e1 = cbox.hi[1] - cbox.lo[1], e2 = cbox.hi[2] - cbox.lo[2];
                if (e1 > e0 && e1 >= e2) a = 1; else if (e2 > e0 && e2 > e1) a = 2;
                mid = first + count / 2;
                std::nth_element(order.begin() + first, order.begin() + mid, order.begin() + first + count,
                                 [&](uint32_t x, uint32_t y) { float cx = cent[3 * x + a], cy = cent[3 * y + a]; return cx < cy || (cx == cy && x < y); });
            }
            const uint32_t lc = mid - first, rc = first + count - mid;
            const int l = id + 1, r = id + 2 * (int)lc;
            nodes[id].left = l; nodes[id].right = r;
            if (spawn_min && lc >= spawn_min) { const uint32_t f0 = first; const int d1 = depth + 1; TaskPool::get().spawn(group, [this, f0, lc, d1, l] { build(f0, lc, d1, l); }); }
            else build(first, lc, depth + 1, l);
            first = mid; count = rc; depth++; id = r; // the right half: this thread goes on
        }
    }
};

} // namespace
void mq_parallel_for(size_t n, size_t grain, const std::function<void(size_t, size_t)>& f) {
    static const bool serial = getenv("MQ_BVH_FORK_DEPTH") && atoi(getenv("MQ_BVH_FORK_DEPTH")) == 0;
    if (grain == 0) grain = 1;
    if (serial || n <= grain || TaskPool::get().threads() == 1) { if (n) f(0, n); return; }
    TaskPool::Busy busy(TaskPool::get());
    TaskGroup g;
    for (size_t b = 0; b < n; b += grain) { const size_t e = std::min(n, b + grain); TaskPool::get().spawn(g, [&f, b, e] { f(b, e); }); }
    TaskPool::get().wait(g);
}
namespace {
inline float exp2i(int e) { uint32_t b = (uint32_t)(e + 127) << 23; float f; memcpy(&f, &b, 4); return f; }

} // namespace

// Build into `out_nodes` / `out_tris` (triangles re-ordered into leaf order). `pad` widens every
// box so the float slab test stays conservative next to the exact triangle test.
// Leaf records of one leaf of the binary tree (at most MQ_BVH_LEAF triangles): triangles that share two vertices (bit-equal
// coordinates) go into one record, the others get a record each.  Appends the triangles to `out_tris` in record order
// (B right behind A) and returns the number of records.
static bool same_vtx(const float* a, const float* b) { return memcmp(a, b, 12) == 0; }
static uint32_t emit_leaf_records(const std::vector<MqTri>& tris, const uint32_t* order, uint32_t count, std::vector<MqTri>& out_tris, std::vector<MqLeafRec>& out_leaves) {
    bool used[8] = {false, false, false, false, false, false, false, false};
    uint32_t n_rec = 0;
    for (uint32_t i = 0; i < count; i++) {
        if (used[i]) continue;
        used[i] = true;
        const MqTri& A = tris[order[i]];
        const float* av[3] = {A.v0, A.v1, A.v2};
        MqLeafRec r; memset(&r, 0, sizeof r);
        for (int k = 0; k < 3; k++) memcpy(r.v[k], av[k], 12);
        memcpy(r.v[3], A.v0, 12);
        r.key0 = A.key; r.key1 = MQ_NIL; r.tri0 = (uint32_t)out_tris.size();
        r.sel = (A.flags & MQ_TRI_ANYHIT) ? 0x10000u : 0u;
        out_tris.push_back(A);
        for (uint32_t j = i + 1; j < count; j++) { // a partner: two of its vertices are vertices of A
            if (used[j]) continue;
            const MqTri& B = tris[order[j]];
            const float* bv[3] = {B.v0, B.v1, B.v2};
            int sel[3], fresh = -1, n_shared = 0;
            for (int k = 0; k < 3; k++) {
                sel[k] = -1;
                for (int m = 0; m < 3; m++) if (same_vtx(bv[k], av[m])) { sel[k] = m; break; }
                if (sel[k] >= 0) n_shared++; else fresh = k;
            }
            if (n_shared < 2) continue;
            if (fresh >= 0) { sel[fresh] = 3; memcpy(r.v[3], bv[fresh], 12); } // (n_shared == 3: a coincident triangle, no fourth vertex)
            r.key1 = B.key;
            r.sel |= (uint32_t)sel[0] | ((uint32_t)sel[1] << 2) | ((uint32_t)sel[2] << 4) | MQ_LEAF_HAS_B | ((B.flags & MQ_TRI_ANYHIT) ? 0x20000u : 0u);
            out_tris.push_back(B);
            used[j] = true;
            break;
        }
        out_leaves.push_back(r);
        n_rec++;
    }
    return n_rec;
}

bool mq_build_cwbvh(const std::vector<MqTri>& tris, std::vector<MqNode>& out_nodes, std::vector<MqTri>& out_tris, std::vector<MqLeafRec>& out_leaves, float* sah_cost, std::string& err, uint32_t* depth_out) {
    if (depth_out) *depth_out = 0;
    out_nodes.clear(); out_tris.clear(); out_leaves.clear();
    if (sah_cost) *sah_cost = 0.0f;
    const uint32_t n = (uint32_t)tris.size();
    if (n == 0) return true;
    // The builder's scratch arrays live as long as the calling thread (a game commits per-frame geometry every frame: zero-filling
    // 10 MB of scratch per build was a tenth of a 65 k-triangle commit); every element in use is written before it is read.  A build
    // of more than a million triangles gives its scratch back.
    TaskPool::Busy busy(TaskPool::get());
    struct Scratch { std::vector<AABB> tbox; std::vector<float> cent; std::vector<uint32_t> order; std::vector<BNode> nodes; };
    static thread_local Scratch scratch;
    Builder B(tris, scratch.tbox, scratch.cent, scratch.order, scratch.nodes);
    struct Release { Scratch& s; bool all; ~Release() { if (all) { Scratch e; std::swap(s, e); } } } release{scratch, n > 1000000u};
    if (B.tbox.size() < n) { B.tbox.resize(n); B.cent.resize(3 * (size_t)n); B.order.resize(n); }
    float maxabs = 1.0f;
    {
        std::mutex mm; bool bad = false;
        mq_parallel_for(n, 8192, [&](size_t b0, size_t b1) {
            float mx = 1.0f; bool nf = false;
            for (size_t i = b0; i < b1; i++) {
                AABB b; b.reset();
                b.grow(tris[i].v0); b.grow(tris[i].v1); b.grow(tris[i].v2);
                B.tbox[i] = b;
                for (int a = 0; a < 3; a++) {
                    B.cent[3 * i + a] = 0.5f * (b.lo[a] + b.hi[a]);
                    mx = std::max(mx, std::max(std::fabs(b.lo[a]), std::fabs(b.hi[a])));
                    if (!std::isfinite(b.lo[a]) || !std::isfinite(b.hi[a])) nf = true;
                }
                B.order[i] = (uint32_t)i;
            }
            std::lock_guard<std::mutex> l(mm); maxabs = std::max(maxabs, mx); bad = bad || nf;
        });
        if (bad) { err = "non-finite vertex"; return false; }
    }
    const float pad = std::max(1e-4f, maxabs * 4.76837158203125e-07f); // 2^-21 * extent
    if (B.nodes.size() < 2 * (size_t)n) B.nodes.resize(2 * (size_t)n);
    auto T0 = std::chrono::steady_clock::now();
    static const int fork_depth = getenv("MQ_BVH_FORK_DEPTH") ? atoi(getenv("MQ_BVH_FORK_DEPTH")) : 5; // 0: single-threaded build
    B.spawn_min = (fork_depth > 0 && n >= 2048 && TaskPool::get().threads() > 1) ? std::max(512u, n / 256u) : 0u;
    B.build(0, n, 0, 0);
    if (B.spawn_min) TaskPool::get().wait(B.group);
    const int root = 0;
    auto T1 = std::chrono::steady_clock::now();

    // ---- collapse to 8-wide ---------------------------------------------------------------------
    // Order in which nodes are expanded = order of their child blocks (and of their triangles and shading records) in
    // memory.  Depth first: a subtree's blocks are neighbours, so a ray walking down finds its next nodes in lines it
    // has just touched (-0.5 % frame time against breadth first, MQ_BVH_ORDER=bfs, which spreads a path over the levels).
    // Depth first also makes everything below a node ONE contiguous range of each output array, so subtrees of at most
    // COLLAPSE_PART triangles are collapsed by tasks of the pool into arrays of their own (indices relative to the subtree)
    // and spliced in where the walk over the top of the tree reaches them: the same arrays as one walk over everything.
    struct Work { int bnode; uint32_t out_index; uint32_t depth; };
    struct Collapsed { std::vector<MqNode> nodes; std::vector<MqLeafRec> leaves; std::vector<MqTri> tris; double sah = 0.0; uint32_t max_depth = 0; };
    static const bool dfs_order = !(getenv("MQ_BVH_ORDER") && !strcmp(getenv("MQ_BVH_ORDER"), "bfs"));
    const float root_area = std::max(B.nodes[root].box.area(), 1e-30f);
    // up to 8 children of a wide node: repeatedly open the internal child with the largest area
    auto gather = [&](int bnode, int ch[8]) -> int {
        int nc = 0;
        const BNode& bn = B.nodes[bnode];
        if (bn.count > 0) { ch[nc++] = bnode; } // root that is a leaf
        else { ch[nc++] = bn.left; ch[nc++] = bn.right; }
        for (;;) {
            if (nc >= 8) break;
            int best = -1; float ba = -1.0f;
            for (int i = 0; i < nc; i++) {
                const BNode& c = B.nodes[ch[i]];
                if (c.count == 0) { float a = c.box.area(); if (a > ba) { ba = a; best = i; } }
            }
            if (best < 0) break;
            int open = ch[best];
            ch[best] = B.nodes[open].left; ch[nc++] = B.nodes[open].right;
        }
        return nc;
    };
    const uint32_t COLLAPSE_PART = 1024;
    std::vector<int> part_root;            // binary nodes whose subtrees are collapsed on their own
    std::vector<Collapsed> parts;
    std::vector<int> part_of(dfs_order && n >= 4 * COLLAPSE_PART ? 2 * (size_t)n : 0, -1); // (the splice relies on the depth-first order)
    auto collapse = [&](int root_bnode, uint32_t depth0, Collapsed& out, bool splice) {
    std::vector<MqNode>& out_nodes = out.nodes; std::vector<MqLeafRec>& out_leaves = out.leaves; std::vector<MqTri>& out_tris = out.tris;
    double& sah = out.sah; uint32_t& max_depth = out.max_depth;
    std::deque<Work> q;
    out_nodes.emplace_back();
    q.push_back({root_bnode, 0, depth0});
    while (!q.empty()) {
        Work w;
        if (dfs_order) { w = q.back(); q.pop_back(); } else { w = q.front(); q.pop_front(); }
        if (splice && part_of[w.bnode] >= 0) { // a subtree collapsed on its own: its root into the slot its parent reserved, the rest behind everything so far
            const Collapsed& P = parts[part_of[w.bnode]];
            const uint32_t node_off = (uint32_t)out_nodes.size() - 1u, leaf_off = (uint32_t)out_leaves.size(), tri_off = (uint32_t)out_tris.size();
            for (size_t k = 0; k < P.nodes.size(); k++) {
                MqNode nd = P.nodes[k]; nd.child_base += node_off; nd.tri_base += leaf_off;
                if (k == 0) out_nodes[w.out_index] = nd; else out_nodes.push_back(nd);
            }
            for (MqLeafRec l : P.leaves) { l.tri0 += tri_off; out_leaves.push_back(l); }
            out_tris.insert(out_tris.end(), P.tris.begin(), P.tris.end());
            sah += P.sah; max_depth = std::max(max_depth, P.max_depth);
            continue;
        }
        max_depth = std::max(max_depth, w.depth);
        int ch[8];
        const BNode& bn = B.nodes[w.bnode];
        const int nc = gather(w.bnode, ch);
        // octant-aware slot assignment: slot bit k set = child lies on the high side of axis k
        AABB pb = bn.box;
        float pc[3] = {0.5f * (pb.lo[0] + pb.hi[0]), 0.5f * (pb.lo[1] + pb.hi[1]), 0.5f * (pb.lo[2] + pb.hi[2])};
        float cost[8][8];
        for (int i = 0; i < nc; i++) {
            const AABB& cb = B.nodes[ch[i]].box;
            float d[3] = {0.5f * (cb.lo[0] + cb.hi[0]) - pc[0], 0.5f * (cb.lo[1] + cb.hi[1]) - pc[1], 0.5f * (cb.lo[2] + cb.hi[2]) - pc[2]};
            for (int s = 0; s < 8; s++) cost[i][s] = ((s & 1) ? d[0] : -d[0]) + ((s & 2) ? d[1] : -d[1]) + ((s & 4) ? d[2] : -d[2]);
        }
        int slot_child[8]; for (int s = 0; s < 8; s++) slot_child[s] = -1;
        bool used[8] = {false, false, false, false, false, false, false, false};
        for (int it = 0; it < nc; it++) {
            int bi = -1, bs = -1; float bc = -INFINITY;
            for (int i = 0; i < nc; i++) if (!used[i]) for (int s = 0; s < 8; s++) if (slot_child[s] < 0 && cost[i][s] > bc) { bc = cost[i][s]; bi = i; bs = s; }
            used[bi] = true; slot_child[bs] = ch[bi];
        }
        // node origin / scale
        MqNode node; memset(&node, 0, sizeof node);
        float lo[3], hi[3];
        for (int a = 0; a < 3; a++) { lo[a] = pb.lo[a] - pad; hi[a] = pb.hi[a] + pad; }
        node.px = lo[0]; node.py = lo[1]; node.pz = lo[2];
        int ex[3];
        for (int a = 0; a < 3; a++) {
            float ext = std::max(hi[a] - lo[a], 1e-30f);
            int e = (int)std::ceil(std::log2((double)ext / 255.0));
            // make sure 255 * 2^e really covers the extent in float arithmetic
            while (lo[a] + 255.0f * exp2i(e) < hi[a]) e++;
            e = std::max(-126, std::min(127, e));
            ex[a] = e;
        }
        node.ex = (uint8_t)(ex[0] + 127); node.ey = (uint8_t)(ex[1] + 127); node.ez = (uint8_t)(ex[2] + 127);
        // children
        uint32_t n_internal = 0;
        for (int s = 0; s < 8; s++) if (slot_child[s] >= 0 && B.nodes[slot_child[s]].count == 0) n_internal++;
        node.child_base = (uint32_t)out_nodes.size();
        node.tri_base = (uint32_t)out_leaves.size();
        out_nodes.resize(out_nodes.size() + n_internal);
        uint32_t next_child = node.child_base, tri_off = 0;
        uint8_t* qlo[3] = {node.qlox, node.qloy, node.qloz};
        uint8_t* qhi[3] = {node.qhix, node.qhiy, node.qhiz};
        for (int s = 0; s < 8; s++) {
            int c = slot_child[s];
            if (c < 0) { // empty: inverted box, never hit
                node.meta[s] = 0;
                for (int a = 0; a < 3; a++) { qlo[a][s] = 255; qhi[a][s] = 0; }
                continue;
            }
            const BNode& cn = B.nodes[c];
            for (int a = 0; a < 3; a++) {
                float e = exp2i(ex[a]);
                float clo = cn.box.lo[a] - pad, chi = cn.box.hi[a] + pad;
                int ql = (int)std::floor(((double)clo - (double)lo[a]) / (double)e);
                int qh = (int)std::ceil(((double)chi - (double)lo[a]) / (double)e);
                ql = std::max(0, std::min(255, ql)); qh = std::max(0, std::min(255, qh));
                while (ql > 0 && lo[a] + (float)ql * e > clo) ql--;   // decoded plane must not cut into the child
                while (qh < 255 && lo[a] + (float)qh * e < chi) qh++;
                qlo[a][s] = (uint8_t)ql; qhi[a][s] = (uint8_t)qh;
            }
            if (cn.count == 0) {
                node.imask |= (uint8_t)(1u << s);
                node.meta[s] = (uint8_t)((1u << 5) | (24u + (uint32_t)s));
                q.push_back({c, next_child++, w.depth + 1});
                sah += (double)cn.box.area() / root_area;
            } else {
                const uint32_t n_rec = emit_leaf_records(tris, &B.order[cn.first], cn.count, out_tris, out_leaves); // 1 .. MQ_BVH_LEAF records
                uint32_t unary = n_rec == 1 ? 1u : (n_rec == 2 ? 3u : 7u);
                node.meta[s] = (uint8_t)((unary << 5) | tri_off);
                tri_off += n_rec;
                sah += (double)cn.box.area() / root_area * cn.count;
            }
        }
        out_nodes[w.out_index] = node;
    }
    }; // collapse
    if (!part_of.empty()) {
        std::vector<std::pair<int, uint32_t>> st; st.push_back({root, 1u});
        std::vector<uint32_t> part_depth;
        while (!st.empty()) { // the top of the tree: which subtrees are small enough to go to a task, and at which depth they hang
            const auto [b, d] = st.back(); st.pop_back();
            if (B.nodes[b].ntris <= COLLAPSE_PART) { part_of[b] = (int)part_root.size(); part_root.push_back(b); part_depth.push_back(d); continue; }
            int ch[8]; const int nc = gather(b, ch);
            for (int i = 0; i < nc; i++) if (B.nodes[ch[i]].count == 0) st.push_back({ch[i], d + 1});
        }
        parts.resize(part_root.size());
        TaskGroup cg;
        const bool pool = B.spawn_min != 0;
        for (size_t k = 0; k < part_root.size(); k++) {
            auto job = [&, k] { parts[k].tris.reserve(B.nodes[part_root[k]].ntris); collapse(part_root[k], part_depth[k], parts[k], false); };
            if (pool) TaskPool::get().spawn(cg, job); else job();
        }
        if (pool) TaskPool::get().wait(cg);
    }
    Collapsed all;
    all.tris.reserve(n);
    collapse(root, 1, all, !part_of.empty());
    out_nodes.swap(all.nodes); out_leaves.swap(all.leaves); out_tris.swap(all.tris);
    const double sah = all.sah; const uint32_t max_depth = all.max_depth;
    if (getenv("MQ_DEBUG_BUILD_TIMES")) fprintf(stderr, "bvh build: %u tris on %d threads, binary tree %.1f ms, collapse %.1f ms\n", n, TaskPool::get().threads(), std::chrono::duration<double, std::milli>(T1 - T0).count(), std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - T1).count());
    if (sah_cost) *sah_cost = (float)sah;
    if (depth_out) *depth_out = max_depth; // levels of 8-wide nodes
    if (out_tris.size() != n) { err = "internal: triangle count mismatch after collapse"; return false; }
    return true;
}

