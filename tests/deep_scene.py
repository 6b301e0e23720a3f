"""A scene whose 8-wide tree is deep AND whose central camera rays keep siblings pending on every level, so that the
per-lane traversal stack (12 entries in LDS, the rest in a global spill area indexed by block and thread) overflows into
the spill area.  No synthetic stand-in scene does: their trees are balanced and 6-7 levels deep.

Geometry: `groups` square frames (four thin slivers each, a hole in the middle) around the +x axis at x = ratio^k with a
half-size that grows with x, i.e. all frames look alike from the origin.  The binned-SAH builder peels the far frames
off one or two at a time, the collapse to 8-wide nodes then makes every node {a few frames, "everything nearer"}: a ray
along the axis enters the box of every frame and of every "nearer" subtree, but hits no sliver until the end wall behind the
nearest frame.  `emulate_stack_depth` replays the kernel's stack discipline on the host to prove the depth."""
import numpy as np


def chain_scene(groups=96, ratio=2.0, ext0=None):
    import mqhip
    vtx, idx = [], []

    def quad(p, du, dv):  # front face towards -x (normal = cross(v2 - v0, v1 - v0), raytrace.glsl:221-223)
        b = len(vtx)
        p, du, dv = (np.asarray(a, np.float64) for a in (p, du, dv))
        vtx.extend([p, p + du, p + du + dv, p + dv])
        idx.extend([[b, b + 1, b + 2], [b, b + 2, b + 3]])

    for k in range(groups):
        x = ratio ** k
        s, w = 0.30 * x, 0.02 * x  # half size of the frame, width of its slivers
        # four slivers: bottom, top, left, right (planes x = const, seen from the origin)
        quad([x, -s, -s], [0, 2 * s, 0], [0, 0, w])
        quad([x, -s, s - w], [0, 2 * s, 0], [0, 0, w])
        quad([x, -s, -s + w], [0, w, 0], [0, 0, 2 * s - 2 * w])
        quad([x, s - w, -s + w], [0, w, 0], [0, 0, 2 * s - 2 * w])
    vtx = np.array(vtx, np.float32)
    idx = np.array(idx, np.uint32)
    # winding: make every triangle face the origin side (-x)
    v0, v1, v2 = vtx[idx[:, 0]], vtx[idx[:, 1]], vtx[idx[:, 2]]
    n = np.cross(v2 - v0, v1 - v0)
    flip = n[:, 0] > 0
    idx[flip] = idx[flip][:, [0, 2, 1]]
    if ext0 is None:
        ext0 = np.zeros(1, mqhip.EXT_DTYPE)
        ext0["texnum_alpha"] = 1 | (15 << 12)  # texture 1, opaque
        ext0["n1_brush"] = 0xffffffff
    return vtx, idx, np.repeat(ext0, len(idx))


def emulate_stack_depth(nodes, org, d, root=0, start_sp=0):
    """Replays trav_node_pop / trav_next of mq_kernels.hip for one ray that never finds a closer hit (the worst case):
    returns the largest number of stack entries in use."""
    org, d = np.asarray(org, np.float64), np.asarray(d, np.float64)
    inv = 1.0 / np.where(np.abs(d) > 1e-20, d, 1e-20)
    octinv = (0 if d[0] < 0 else 1) | (0 if d[1] < 0 else 2) | (0 if d[2] < 0 else 4)
    stack, deepest = [None] * start_sp, start_sp
    G = (root, 0x80000000)  # the root "group": one pending child, bit 31
    visits = 0
    while True:
        if G[1] > 0x00ffffff:
            bit = G[1].bit_length() - 1
            gy = G[1] & ~(1 << bit)
            if gy > 0x00ffffff:
                stack.append((G[0], gy)); deepest = max(deepest, len(stack))
            slot = (bit - 24) ^ octinv
            rel = bin(gy & 0xff & ((1 << slot) - 1)).count("1")
            nd = nodes[G[0] + rel]
            visits += 1
            hm = 0
            for s in range(8):
                meta = int(nd["meta"][s])
                if meta == 0:
                    continue
                lo = np.array([nd["qlo"][a][s] for a in range(3)], np.float64)
                hi = np.array([nd["qhi"][a][s] for a in range(3)], np.float64)
                e = np.array([2.0 ** (int(nd["e"][a]) - 127) for a in range(3)])
                blo, bhi = nd["p"].astype(np.float64) + lo * e, nd["p"].astype(np.float64) + hi * e
                t0, t1 = (blo - org) * inv, (bhi - org) * inv
                tn, tf = np.minimum(t0, t1).max(), np.maximum(t0, t1).min()
                if max(tn, 0.0) <= min(tf, 1e4 * 1.000001 + 1e-6):
                    if (meta & 0x18) == 0x18:
                        hm |= (meta >> 5) << (24 + ((meta & 7) ^ octinv))
            G = (int(nd["child_base"]), (hm & 0xff000000) | int(nd["imask"]))
            continue
        if not stack[start_sp:]:
            return deepest, visits
        G = stack.pop()
