"""CPU tests of the host side of libmqhip: the C ABI surface, the reference's property table, the
scene sources (synthetic generator, BSP29/BSP2 loader) and the BVH builder.  No GPU compute: a
host-only context must refuse every device entry point with MQ_ENODEVICE (there is no CPU path)."""
import json
import os
import re
import struct

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def mq(mqlib):
    import mqhip
    return mqhip


def test_library_exports_every_declared_symbol(mq, mqlib):
    hdr = open(os.path.join(ROOT, "include", "mq.h")).read()
    declared = set(re.findall(r"^\s*(?:int|void|const char\*)\s+(mq_[a-z0-9_]+)\s*\(", hdr, re.M))
    assert len(declared) >= 40
    import ctypes
    for name in declared:
        assert hasattr(mqlib, name), "libmqhip.so does not export %s" % name
    assert declared == set(mqlib._mq_symbols), declared ^ set(mqlib._mq_symbols)
    assert mqlib.mq_abi_version() == 3
    assert ctypes.sizeof(mq.Uniform) == 124  # res/shader/scene_info.glsl.h:18-32
    assert mq.EXT_DTYPE.itemsize == 28       # src/game/quake_helpers.hpp:10-34


def test_host_only_context_refuses_device_work(mq):
    ctx = mq.Context(-1)
    ctx.synth_scene("synth_tiny", 1)
    ctx.commit()  # BVH build is host work
    assert ctx.scene_stats()["n_tris"] > 500
    for call in (lambda: ctx.connect(64, 48), lambda: ctx.process(ctx.synth_camera(0)), lambda: ctx.sync(),
                 lambda: ctx.trace_rays(np.zeros((1, 3), np.float32), np.ones((1, 3), np.float32)),
                 lambda: ctx.math_eval(0, np.zeros((1, 1), np.float32), 1), lambda: ctx.last_frame_ms(), lambda: ctx.counters()):
        with pytest.raises(mq.MqError) as e:
            call()
        assert e.value.code == mq.MQ_ENODEVICE
    with pytest.raises(mq.MqError):
        mq.Context(99)  # no such device


# property keys and shipped values transcribed from res/default_config.json:599-638 (render_markovchain) and :527-535 (gbuffer)
REFERENCE_JSON = {"graph": {"nodes": {
    "gbuffer": {"disable": False, "properties": {"enable albedo mipmap": True, "enable emission mipmap": True, "hide sun": True}, "type": "GBuffer"},
    "render_markovchain": {"disable": False, "properties": {
        "BSDF Prob": 0.10000000149011612, "LC buf size": 4000037, "LC grid min width": 0.009999999776482582, "LC grid power": 2.0,
        "LC grid steps per unit": 6.0, "LC grid tan(alpha/2)": 0.004999999888241291, "LC grid type": "quadratic", "ML Prior": 0.30000001192092896,
        "Phase Prob": 0.10000000149011612, "adaptive grid buf size": 32777259, "adaptive grid min width": 0.009999999776482582,
        "adaptive grid power": 1.7320507764816284, "adaptive grid prob": 0.699999988079071, "adaptive grid steps per unit": 1.0,
        "adaptive grid tan(alpha/2)": 0.0020000000949949026, "adaptive grid type": "exponential", "debug output": "light cache",
        "dist guide p": 0.8999999761581421, "dist mc grid width": 25, "dist mc samples": 3, "dist mc states per vertex": 10,
        "max path length": 3, "mc fast recovery": True, "mc samples": 5, "mc static width": 25.299999237060547, "particle size": 7.0,
        "randomize seed": True, "reference mode": False, "spp": 2, "static grid buf size": 800009, "surf: use LC": False,
        "volume forward project": True, "volume spp": 2, "volume: use LC": True}, "type": "Renderer (MCPG)"}}}}


def test_property_table_uses_reference_keys_and_defaults(mq):
    ctx = mq.Context(-1)
    names = ctx.property_names()
    ref_keys = set(REFERENCE_JSON["graph"]["nodes"]["render_markovchain"]["properties"]) | set(REFERENCE_JSON["graph"]["nodes"]["gbuffer"]["properties"])
    assert ref_keys <= set(names), ref_keys - set(names)
    # header defaults: src/render_mcpg/render_mcpg.hpp:108-166
    hdr = {"spp": 1, "volume spp": 0, "max path length": 3, "BSDF Prob": 0.15, "Phase Prob": 0.3, "ML Prior": 0.2, "dist guide p": 0.0,
           "mc samples": 5, "adaptive grid prob": 0.7, "adaptive grid tan(alpha/2)": 0.003, "adaptive grid power": 4.0,
           "adaptive grid steps per unit": 6.0, "static grid buf size": 800009, "mc static width": 25.3, "LC buf size": 4000000,
           "LC grid type": 0, "randomize seed": 1, "mc fast recovery": 1, "adaptive grid buf size": 32777259, "hide sun": 1}
    for k, v in hdr.items():
        assert ctx.get_property(k) == pytest.approx(v, rel=1e-6), k
    # json_defaults() == loading the reference's shipped JSON
    a = mq.Context(-1); a.json_defaults()
    b = mq.Context(-1)
    txt = json.dumps(REFERENCE_JSON)
    assert b.load_properties_json(txt, "render_markovchain") == 1  # NEEDS_RECONNECT: LC grid type / sizes changed
    b.load_properties_json(txt, "gbuffer")
    for k in names:
        assert a.get_property(k) == b.get_property(k), k
    assert b.get_property("LC grid type") == 1 and b.get_property("spp") == 2 and b.get_property("volume: use LC") == 1
    # option by string, unknown key, out-of-range values
    assert ctx.set_property("LC grid type", "quadratic") == 1      # reconnect class (render_mcpg.cpp:567-575)
    assert ctx.set_property("BSDF Prob", 0.25) == 0                # pipeline-refresh class
    assert ctx.set_property("BSDF Prob", 0.25) == 0
    with pytest.raises(mq.MqError):
        ctx.set_property("no such key", 1)
    with pytest.raises(mq.MqError):
        ctx.set_property("LC grid type", "cubic")
    with pytest.raises(mq.MqError):
        ctx.set_property("mc samples", 31)


def test_describe_matches_reference_buffer_sizes(mq):
    ctx = mq.Context(-1)
    d = ctx.describe(1920, 1080)
    px = 1920 * 1080
    assert d.bytes[mq.OUT_IRRADIANCE] == px * 16       # RGBA32F, render_mcpg.cpp:42-43
    assert d.bytes[mq.OUT_HITS] == px * 40             # CompressedHit, gbuffer.cpp:39
    assert d.bytes[mq.OUT_GB_ALBEDO] == px * 8 and d.bytes[mq.OUT_GB_MV] == px * 4
    assert d.state_bytes_lightcache == 4000000 * 16
    assert d.state_bytes_markovchain == (32777259 + 800009) * (64 + 8)  # render_mcpg.cpp:59 slots, 64 B states + count and chain head words


def test_synthetic_scenes_are_deterministic_and_well_formed(mq):
    a, b = mq.Context(-1), mq.Context(-1)
    a.synth_scene("synth_start", 5); b.synth_scene("synth_start", 5)
    ga, gb = a.get_geometry(0), b.get_geometry(0)
    assert np.array_equal(ga["vtx"], gb["vtx"]) and np.array_equal(ga["idx"], gb["idx"]) and np.array_equal(ga["ext"], gb["ext"])
    c = mq.Context(-1); c.synth_scene("synth_start", 6)
    assert not np.array_equal(c.get_geometry(0)["ext"], ga["ext"])
    assert 25000 < len(ga["idx"]) < 45000  # stand-in for id1 start.bsp, SURVEY 8d
    assert ga["flags"] & mq.MQ_GEO_OPAQUE
    # every triangle is non-degenerate and its reference normal cross(v2-v0, v1-v0) is axis aligned or a pillar side
    tri = ga["vtx"][ga["idx"]].astype(np.float64)
    nrm = np.cross(tri[:, 2] - tri[:, 0], tri[:, 1] - tri[:, 0])
    assert (np.linalg.norm(nrm, axis=1) > 1e-3).all()
    # emissive (fullbright) tiles exist and are a few percent; sky brushes only in outdoor scenes
    fb = ga["ext"]["texnum_fb_flags"]
    assert 0.005 < ((fb & 0xFFF) > 0).mean() < 0.1
    assert ((fb >> 12) == 5).sum() == 0
    d = mq.Context(-1); d.synth_scene("synth_tiny", 1)
    assert ((d.get_geometry(0)["ext"]["texnum_fb_flags"] >> 12) == 5).sum() > 0
    # alpha-tested slot uses texture alpha (nibble 0) and is not flagged opaque; dynamic slot has prev_vtx != vtx
    g1 = a.get_geometry(1)
    if g1 is not None:
        assert (g1["ext"]["texnum_alpha"] >> 12 == 0).all() and not (g1["flags"] & mq.MQ_GEO_OPAQUE)
    g2 = a.get_geometry(2)
    assert g2 is not None and not np.array_equal(g2["vtx"], g2["prev_vtx"])
    u0, u1 = a.synth_camera(0), a.synth_camera(1)
    assert list(u1.prev_cam_x)[:3] == list(u0.cam_x)[:3]
    assert abs(np.linalg.norm(list(u0.cam_w)[:3]) - 1) < 1e-5 and abs(np.dot(list(u0.cam_w)[:3], list(u0.cam_u)[:3])) < 1e-5
    with pytest.raises(mq.MqError):
        a.synth_scene("no_such_scene", 1)


def _decode_children(node):
    e = (node["e"].astype(np.uint32) << 23).view(np.float32)
    lo = node["p"][:, None] + node["qlo"].astype(np.float32) * e[:, None]
    hi = node["p"][:, None] + node["qhi"].astype(np.float32) * e[:, None]
    return lo, hi


def _walk_bvh(nodes, tris, root=0, leaves=None):
    """Checks the structural invariants below `root`; returns the sets of nodes and triangles reached.  A leaf slot addresses
    1..3 LEAF RECORDS (one or two triangles that share an edge, as four vertices): the record's triangles must be the
    triangle array's tri0 / tri0 + 1, vertex for vertex in their own order."""
    seen_nodes, seen_tris = set(), set()
    stack = [(root, None, None)]
    while stack:
        ni, plo, phi = stack.pop()
        assert ni not in seen_nodes
        seen_nodes.add(ni)
        node = nodes[ni]
        lo, hi = _decode_children(node)
        child = int(node["child_base"])
        for s in range(8):
            m = int(node["meta"][s])
            if m == 0:
                assert node["qlo"][0][s] > node["qhi"][0][s]  # empty slots can never be hit
                continue
            clo, chi = lo[:, s], hi[:, s]
            if plo is not None:  # child box inside the parent's decoded box, up to one step of this node's own (finer) grid
                tol = (node["e"].astype(np.uint32) << 23).view(np.float32) + 1e-3
                assert (clo >= plo - tol).all() and (chi <= phi + tol).all()
            if (m & 0x18) == 0x18:  # internal: low 5 bits = 24 + slot, imask bit set, children contiguous in slot order
                assert (m & 31) == 24 + s and (m >> 5) == 1 and (int(node["imask"]) >> s) & 1
                stack.append((child, clo, chi)); child += 1
            else:
                cnt = {1: 1, 3: 2, 7: 3}[m >> 5]
                first = int(node["tri_base"]) + (m & 31)
                for rec in range(first, first + cnt):
                    L = leaves[rec]
                    sel = int(L["sel"])
                    members = [(int(L["tri0"]), L["v"][:3], int(L["key0"]), bool(sel & 0x10000))]
                    if sel & 0x100:
                        members.append((int(L["tri0"]) + 1, L["v"][[sel & 3, (sel >> 2) & 3, (sel >> 4) & 3]], int(L["key1"]), bool(sel & 0x20000)))
                    else:
                        assert int(L["key1"]) == 0xffffffff
                    for t, v, key, anyhit in members:
                        assert t not in seen_tris
                        seen_tris.add(t)
                        assert v.tobytes() == tris[t]["v"].tobytes() and key == int(tris[t]["key"]) and anyhit == bool(int(tris[t]["flags"]) & 1)
                        assert (v.min(0) >= clo).all() and (v.max(0) <= chi).all()  # conservative quantised boxes
    return seen_nodes, seen_tris


def test_cwbvh_invariants(mq):
    ctx = mq.Context(-1)
    ctx.synth_scene("synth_tiny", 2)
    ctx.commit()
    nodes, tris = ctx.get_bvh()
    leaves = ctx.get_leaves()
    total = sum(len(ctx.get_geometry(s)["idx"]) for s in range(3) if ctx.get_geometry(s) is not None)
    assert len(tris) == total
    assert len(np.unique(tris["key"])) == total  # every triangle exactly once
    assert total / 2 <= len(leaves) <= total and ((leaves["sel"] & 0x100) != 0).sum() == total - len(leaves)
    assert ((leaves["sel"] & 0x13f) == 0x138).mean() > 0.5  # mostly quads: the second triangle is (v0, v2, v3)
    seen_nodes, seen_tris = _walk_bvh(nodes, tris, 0, leaves)
    n_s = ctx.scene_layout()[0]
    if n_s < len(nodes):  # the per-frame tree has its own root behind the static tree
        more = _walk_bvh(nodes, tris, n_s, leaves)
        assert not (seen_nodes & more[0]) and not (seen_tris & more[1])
        seen_nodes |= more[0]; seen_tris |= more[1]
    assert len(seen_nodes) == len(nodes) and len(seen_tris) == len(tris)
    st = ctx.scene_stats()
    assert st["bvh_bytes"] == len(nodes) * 80 + len(leaves) * 64


def test_static_and_per_frame_trees(mq):
    """Static slots and per-frame slots are two trees (quake_node.cpp:847-983): the per-frame tree is stored behind the
    static one and replacing a per-frame slot leaves the static part of the arrays untouched."""
    ctx = mq.Context(-1)
    ctx.synth_scene("synth_tiny", 2)
    ctx.commit()
    nodes0, tris0 = ctx.get_bvh()
    n_static = len(tris0)
    assert ctx.scene_layout() == (len(nodes0), n_static)
    ext0 = ctx.get_geometry(0)["ext"][:1]
    for step, n_tri in enumerate((5, 40, 3)):
        rng = np.random.default_rng(step)
        vtx = (rng.random((3 * n_tri, 3), dtype=np.float32) * 200 - 100).astype(np.float32)
        idx = np.arange(3 * n_tri, dtype=np.uint32).reshape(-1, 3)
        ctx.set_geometry(4, vtx, vtx + 1.0, idx, np.repeat(ext0, n_tri), mq.MQ_GEO_OPAQUE)
        ctx.commit()
        nodes, tris = ctx.get_bvh()
        assert len(tris) == n_static + n_tri and ctx.scene_layout() == (len(nodes0), n_static)
        assert nodes[:len(nodes0)].tobytes() == nodes0.tobytes() and tris[:n_static].tobytes() == tris0.tobytes()
        leaves = ctx.get_leaves()
        s_nodes, s_tris = _walk_bvh(nodes, tris, 0, leaves)
        d_nodes, d_tris = _walk_bvh(nodes, tris, len(nodes0), leaves)  # the per-frame root
        assert s_nodes == set(range(len(nodes0))) and d_nodes == set(range(len(nodes0), len(nodes)))
        assert s_tris == set(range(n_static)) and d_tris == set(range(n_static, n_static + n_tri))
        assert ((tris["key"][n_static:] >> 28) == 4).all() and (tris["flags"][n_static:] & 2).all()  # distinct previous positions
    ctx.set_geometry(4, np.zeros((0, 3), np.float32), None, np.zeros((0, 3), np.uint32), np.zeros(0, mq.EXT_DTYPE), 0)
    ctx.commit()
    nodes, tris = ctx.get_bvh()
    assert nodes.tobytes() == nodes0.tobytes() and tris.tobytes() == tris0.tobytes()


def _write_bsp(path, bsp2):
    """A one-room box map: 6 quad faces, two textures (a wall with fullbright texels, a sky), worldspawn sun keys."""
    verts = np.array([[x, y, z] for z in (0, 128) for y in (0, 256) for x in (0, 256)], np.float32)
    quads = [(0, 1, 3, 2), (4, 6, 7, 5), (0, 4, 5, 1), (2, 3, 7, 6), (0, 2, 6, 4), (1, 5, 7, 3)]
    edges = [(0, 0)]
    surfedges = []
    faces = []
    for qi, q in enumerate(quads):
        first = len(surfedges)
        for k in range(4):
            edges.append((q[k], q[(k + 1) % 4])); surfedges.append(len(edges) - 1)
        faces.append((first, 4, 1 if qi == 1 else 0))
    def miptex(name, w, h, px):
        off = 40
        return struct.pack("<16sII4I", name, w, h, off, 0, 0, 0) + bytes(px)
    wall = [100] * (16 * 16); wall[5] = 250; wall[6] = 251  # two fullbright texels (palette index >= 224)
    sky = [7] * (32 * 16)
    mips = [miptex(b"wall1", 16, 16, wall), miptex(b"sky1", 32, 16, sky)]
    ofs = 4 + 4 * len(mips)
    lump_tex = struct.pack("<i", len(mips))
    acc = b""
    for m in mips:
        lump_tex += struct.pack("<i", ofs + len(acc)); acc += m
    lump_tex += acc
    texinfo = b"".join(struct.pack("<8fii", 1, 0, 0, 0, 0, 1, 0, 0, i, 0) for i in range(2))
    if bsp2:
        lump_faces = b"".join(struct.pack("<iiiii4Bi", 0, 0, f[0], f[1], f[2], 0, 0, 0, 0, -1) for f in faces)
        lump_edges = b"".join(struct.pack("<II", *e) for e in edges)
    else:
        lump_faces = b"".join(struct.pack("<hhihh4Bi", 0, 0, f[0], f[1], f[2], 0, 0, 0, 0, -1) for f in faces)
        lump_edges = b"".join(struct.pack("<HH", *e) for e in edges)
    ents = b'{\n"classname" "worldspawn"\n"_sunlight" "8000"\n"_sunlight_color" "1 0.5 0.25"\n"_sun_mangle" "90 -45 0"\n}\n{\n"classname" "info_player_start"\n"origin" "128 128 24"\n"angle" "90"\n}\n\x00'
    model = struct.pack("<9f4iiii", 0, 0, 0, 256, 256, 128, 0, 0, 0, 0, 0, 0, 0, 0, 0, len(faces))
    lumps = [ents, b"", lump_tex, verts.tobytes(), b"", b"", texinfo, lump_faces, b"", b"", b"", b"", lump_edges,
             struct.pack("<%di" % len(surfedges), *surfedges), model]
    hdr_len = 4 + 15 * 8
    body = b""; table = b""
    for l in lumps:
        table += struct.pack("<ii", hdr_len + len(body), len(l)); body += l + b"\x00" * ((-len(l)) % 4)
    with open(path, "wb") as f:
        f.write((b"BSP2" if bsp2 else struct.pack("<i", 29)) + table + body)


@pytest.mark.parametrize("bsp2", [False, True])
def test_bsp_loader(mq, tmp_path, bsp2):
    path = str(tmp_path / ("box2.bsp" if bsp2 else "box29.bsp"))
    _write_bsp(path, bsp2)
    ctx = mq.Context(-1)
    ctx.load_bsp(path)
    g = ctx.get_geometry(0)
    assert len(g["idx"]) == 12 and len(g["vtx"]) == 24          # 6 quads -> fan triangles (quake_helpers.cpp:419-423)
    assert np.array_equal(g["idx"][:2], [[0, 1, 2], [0, 2, 3]])
    flags = g["ext"]["texnum_fb_flags"] >> 12
    assert (flags == 5).sum() == 2                                # the sky face carries MAT_FLAGS_SKY
    assert (g["ext"]["n1_brush"] == 0xFFFFFFFF).all()
    wall = g["ext"][flags == 0]
    assert (wall["texnum_alpha"] >> 12 == 15).all() and ((wall["texnum_fb_flags"] & 0xFFF) > 0).all()  # opaque + fullbright mask
    fb_px, _ = ctx.get_texture(int(wall["texnum_fb_flags"][0] & 0xFFF))
    assert (fb_px[..., :3].reshape(-1, 3).sum(1) > 0).sum() == 2  # only the two fullbright texels survive in the mask
    c = ctx.get_constants()
    assert c["sun_color"] == pytest.approx([2.0, 1.0, 0.5])       # 8000/4000 * colour (quake_node.cpp:266-283)
    assert abs(np.linalg.norm(c["sun_direction"]) - 1) < 1e-6
    u = ctx.synth_camera(0)
    assert abs(u.cam_x[2] - 46.0) < 1e-3                          # origin z + 22 eye height
    assert (u.sky_lf_ft & 0xFFFF) == 0xFFFF                       # classic sky marker (raytrace.glsl:35)
    ctx.commit()
    assert ctx.scene_stats()["n_tris"] == 12
    with pytest.raises(mq.MqError):
        ctx.load_bsp(str(tmp_path / "missing.bsp"))
    bad = tmp_path / "bad.bsp"
    bad.write_bytes(struct.pack("<i", 30) + b"\x00" * 200)
    with pytest.raises(mq.MqError):
        ctx.load_bsp(str(bad))


def test_scene_upload_validation(mq):
    ctx = mq.Context(-1)
    vtx = np.zeros((3, 3), np.float32)
    ext = np.zeros(1, mq.EXT_DTYPE)
    with pytest.raises(mq.MqError):
        ctx.set_geometry(0, vtx, None, np.array([[0, 1, 3]], np.uint32), ext, 0)  # index out of range
    with pytest.raises(mq.MqError):
        ctx.set_geometry(16, vtx, None, np.array([[0, 1, 2]], np.uint32), ext, 0)  # MAX_GEOMETRIES (config.h:6)
    with pytest.raises(mq.MqError):
        ctx.set_texture(4096, np.zeros((2, 2, 4), np.uint8), 0)                     # MAX_GLTEXTURES (config.h:5)
    ctx.commit()  # an empty scene commits (everything misses -> sky)
    assert ctx.scene_stats()["n_tris"] == 0


def test_cpp_node_adapter(built, tmp_path):
    """include/mq_node.hpp (the merian-style five-method adapter) compiles against mq.h alone and behaves."""
    import subprocess
    exe = str(tmp_path / "node_adapter_test")
    libdir = os.path.join(ROOT, "merian-quake_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "node_adapter_test.cpp"),
                    "-L" + libdir, "-lmqhip", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    r = subprocess.run([exe], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=120)
    assert r.returncode == 0 and "node adapter ok" in r.stdout, r.stdout


def test_deep_chain_scene_needs_more_than_the_lds_stack(mq):
    """tests/deep_scene.py (used by the GPU suite to exercise the traversal-stack spill area): its central ray keeps 15 entries
    on the per-lane stack -- the 12 LDS entries are not enough -- as a host replay of the kernel's stack discipline on the
    BVH the builder really produced shows."""
    import deep_scene
    ctx = mq.Context(-1)
    vtx, idx, ext = deep_scene.chain_scene(80, 1.6)
    ctx.set_geometry(0, vtx, None, idx, ext, mq.MQ_GEO_OPAQUE | mq.MQ_GEO_STATIC)
    ctx.commit()
    nodes, tris = ctx.get_bvh()
    assert len(tris) == len(idx)
    depth, visits = deep_scene.emulate_stack_depth(nodes, [0.25, 0, 0], [1, 0, 0])
    assert depth >= 14 and visits > 40, (depth, visits)
    depth_back, _ = deep_scene.emulate_stack_depth(nodes, [1.6 ** 80, 0, 0], [-1, 0, 0])
    assert depth_back <= 2  # looking back from the far end the first box test culls everything beyond T_MAX
    ctx.close()


def test_builder_output_does_not_depend_on_the_worker_pool():
    """The host's worker pool (mq_bvh.cpp: SAH subtrees, the passes over large nodes and the collapse of <= 1024-triangle
    subtrees as tasks, spliced into the depth-first layout) must produce the arrays a single thread produces: nodes,
    triangles and leaf records of a 640 k-triangle static scene plus a 20 k-triangle per-frame tree, hashed in a fresh
    process per setting (the pool's size is fixed when it starts)."""
    import subprocess
    import sys
    code = (
        "import sys, hashlib, numpy as np\n"
        "sys.path.insert(0, %r)\n"
        "import mqhip\n"
        "c = mqhip.Context(-1); c.json_defaults(); c.synth_scene('synth_sepulcher', 2); c.commit()\n"
        "v = mqhip.View(); v.forward[0] = 1.0; v.right[1] = -1.0; v.up[2] = 1.0\n"
        "rng = np.random.default_rng(1); n = 5000\n"
        "p = np.zeros(n, mqhip.PARTICLE_DTYPE); p['org'] = rng.uniform(-500, 500, (n, 3)); p['prev_org'] = p['org'] - 1; p['seed'] = rng.integers(1, 2 ** 32, n); p['color_rgba'] = 0xffffff\n"
        "c.dyn_begin(); c.dyn_add_particles(p, v, 1, 2, 0.5, 0.4); c.dyn_end(2); c.commit()\n"
        "nodes, tris = c.get_bvh()\n"
        "g = c.get_geometry(2)\n"
        "print(len(nodes), len(tris), hashlib.md5(nodes.tobytes() + tris.tobytes() + c.get_leaves().tobytes() + g['vtx'].tobytes() + g['ext'].tobytes()).hexdigest())\n"
    ) % os.path.join(ROOT, "merian-quake_amd")
    outs = {}
    for name, env in (("serial", {"MQ_BVH_FORK_DEPTH": "0"}), ("one thread", {"MQ_BVH_THREADS": "1"}), ("three threads", {"MQ_BVH_THREADS": "3"}), ("eight threads", {"MQ_BVH_THREADS": "8"})):
        r = subprocess.run([sys.executable, "-c", code], env={**os.environ, **env}, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, (name, r.stderr[-2000:])
        outs[name] = r.stdout.strip().splitlines()[-1]
    assert len(set(outs.values())) == 1, outs
    assert int(outs["serial"].split()[1]) == 639310 + 4 * 5000  # the static triangles + four per particle
