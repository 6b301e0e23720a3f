"""Scene ingestion beyond the world model: brush models of a BSP as entities, external normal /
gloss maps, alias models (MDL), sprites (SPR) and particles as per-frame geometry with previous positions -- the product's
producers (mq_producers.cpp, mq_bsp.cpp) against the numpy oracle (oracle/producers.py), and a BSP-loaded scene with all
of them rendered on the GPU against the renderer's oracle."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import producers as P  # the numpy oracle
import quake_files as Q

TOL = 1e-4  # positions: float32 arithmetic in another order (stated in oracle/producers.py)


def view_of(mq, origin, angles):
    f, r, u = P.angle_vectors(angles)
    v = mq.View()
    for k in range(3):
        v.origin[k], v.forward[k], v.right[k], v.up[k] = origin[k], f[k], r[k], u[k]
    return v, dict(origin=origin, forward=f, right=r, up=u)


def assert_geo_equal(got, ref):
    gv, gp, gi, ge = got["vtx"], got["prev_vtx"], got["idx"], got["ext"]
    rv, rp, ri, re_ = ref
    assert gv.shape == rv.shape and gi.shape == ri.shape and len(ge) == len(re_)
    assert np.abs(gv - rv).max() < TOL and np.abs(gp - rp).max() < TOL
    assert np.array_equal(gi, ri)
    for f in ("texnum_alpha", "texnum_fb_flags", "st"):
        assert np.array_equal(ge[f], re_[f]), f
    return ge, re_


@pytest.fixture(scope="module")
def mq(mqlib):
    import mqhip
    return mqhip


@pytest.mark.parametrize("bsp2", [False, True])
def test_bsp_brush_models_liquids_fences_and_external_maps(mq, tmp_path, bsp2):
    maps = tmp_path / "id1" / "maps"; tex = tmp_path / "id1" / "textures"
    maps.mkdir(parents=True); tex.mkdir()
    path = str(maps / "room.bsp")
    Q.write_bsp(path, bsp2, rich=True)
    rng = np.random.default_rng(3)
    norm = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8); gloss = rng.integers(0, 256, (4, 8, 4), dtype=np.uint8)
    Q.write_tga(str(tex / "wall1_norm.tga"), norm); Q.write_tga(str(tex / "wall1_gloss.tga"), gloss, rle=True)
    ctx = mq.Context(-1)
    ctx.load_bsp(path)
    opaque, alpha = ctx.get_geometry(0), ctx.get_geometry(1)
    assert len(opaque["idx"]) == 16 and len(alpha["idx"]) == 4  # room 12 + water 2 + lava 2 | fence, both sides (quake_node.cpp:847-894)
    fl = opaque["ext"]["texnum_fb_flags"] >> 12
    assert (fl == 4).sum() == 2 and (fl == 1).sum() == 2 and (fl == 5).sum() == 2  # water, lava, sky (quake_helpers.cpp:447-457)
    assert (alpha["ext"]["texnum_alpha"] >> 12 == 0).all() and (opaque["ext"]["texnum_alpha"] >> 12 == 15).all()  # fence: use the texture's alpha
    fence_px, _ = ctx.get_texture(int(alpha["ext"]["texnum_alpha"][0] & 0xfff))
    assert (fence_px[..., 3] == 0).any() and (fence_px[..., 3] == 255).any()
    walls = opaque["ext"][fl == 0]
    tn_norm, tn_gloss = int(walls["n0_gloss_norm"][0] >> 16), int(walls["n0_gloss_norm"][0] & 0xffff)
    assert tn_norm > 0 and tn_gloss > 0 and (walls["n0_gloss_norm"] == walls["n0_gloss_norm"][0]).all()
    npx, nflags = ctx.get_texture(tn_norm); gpx, gflags = ctx.get_texture(tn_gloss)
    assert np.array_equal(npx, norm) and np.array_equal(gpx, gloss)
    assert not (nflags & mq.MQ_TEX_SRGB) and not (gflags & mq.MQ_TEX_SRGB)  # linear, quake_node.hpp:93-95
    assert (opaque["ext"][fl == 4]["n0_gloss_norm"] == 0).all()
    # the door: brush model 1, placed per frame under an entity transform (add_geo_brush, quake_helpers.cpp:362-390)
    assert ctx.bsp_model_count() == 2
    ctx.dyn_begin()
    ctx.dyn_add_brush_model(1, [10, 0, 5], [0, 30, 0], [8, 0, 5], [0, 25, 0])
    ctx.dyn_end(2)
    door = ctx.get_geometry(2)
    assert len(door["idx"]) == 8 and door["flags"] == 0 and (door["ext"]["n1_brush"] == 0xffffffff).all()
    ctx.dyn_begin(); ctx.dyn_add_brush_model(1, [0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0]); ctx.dyn_end(3)
    rest = ctx.get_geometry(3)
    g = P.Geo()
    P.add_brush_model(g, (rest["vtx"], rest["idx"], rest["ext"]), [10, 0, 5], [0, 30, 0], [8, 0, 5], [0, 25, 0])
    assert_geo_equal(door, g.arrays())
    assert np.abs(door["vtx"] - door["prev_vtx"]).max() > 1.0  # it moved
    with pytest.raises(mq.MqError):
        ctx.dyn_add_brush_model(1, [0, 0, 0], [0, 0, 0], [0, 0, 0], [0, 0, 0])  # outside begin / end


def test_particles_match_oracle(mq):
    rng = np.random.default_rng(11)
    n = 60
    parts = np.zeros(n, mq.PARTICLE_DTYPE)
    parts["org"] = rng.uniform(-200, 200, (n, 3)); parts["prev_org"] = parts["org"] - rng.uniform(-2, 2, (n, 3)); parts["vel"] = rng.uniform(-80, 80, (n, 3))
    cols = [0x0000003c, 0x00ffffff, 0x0040a0ff, 0x00808080, 0x000000c8, 0x0010d0f0]
    parts["color_rgba"] = rng.choice(cols, n); parts["type"] = rng.choice([0, 1, 3, 5], n); parts["seed"] = rng.integers(1, 2 ** 32, n)
    ctx = mq.Context(-1)
    view, vd = view_of(mq, [0, 0, 20], [10, 40, 0])
    ctx.dyn_begin(); ctx.dyn_add_particles(parts, view, 33, 44, 2.5, 2.4); ctx.dyn_end(2)
    got = ctx.get_geometry(2)
    g = P.Geo()
    P.add_particles(g, parts, vd["origin"], vd["forward"], 33, 44, 2.5, 2.4)
    ge, re_ = assert_geo_equal(got, g.arrays())
    assert len(got["idx"]) == 4 * n  # one tetrahedron each
    solid = (ge["texnum_fb_flags"] >> 12) == 8
    assert solid.any() and (~solid).any()
    assert np.array_equal(ge["n0_gloss_norm"][solid], re_["n0_gloss_norm"][solid]) and np.array_equal(ge["n1_brush"][solid], re_["n1_brush"][solid])  # colour / emitted colour
    assert (ge["texnum_alpha"][~solid] & 0xfff).tolist().count(33) > 0 and ((ge["texnum_fb_flags"][~solid] & 0xfff) == 44).any()  # blood patches, explosion patches that emit
    assert np.abs(got["vtx"] - got["prev_vtx"]).max() > 0.5


@pytest.mark.parametrize("sprite_type", [0, 1, 2, 3, 4])
def test_sprites_match_oracle(mq, tmp_path, sprite_type):
    rng = np.random.default_rng(5)
    path = str(tmp_path / "s.spr")
    Q.write_spr(path, rng, sprite_type)
    ctx = mq.Context(-1)
    model, nxt = ctx.load_spr(path, 100)
    assert nxt == 102
    px, _ = ctx.get_texture(100)
    assert px.shape == (24, 16, 4) and (px[..., 3] == 0).any()  # index 255 is transparent
    view, vd = view_of(mq, [5, -3, 30], [-8, 75, 3])
    inst = mq.SpriteInstance()
    spr = dict(type=sprite_type, frames=[dict(up=12, down=-12, left=-8, right=8, smax=1, tmax=1, texnum=100), dict(up=4, down=-4, left=-4, right=4, smax=1, tmax=1, texnum=101)])
    for frame in (0, 1):
        d = dict(origin=[100, 50, 40], prev_origin=[98, 50, 41], angles=[10, 20, 30], scale=1.5, frame=frame)
        for k in range(3):
            inst.origin[k], inst.prev_origin[k], inst.angles[k] = d["origin"][k], d["prev_origin"][k], d["angles"][k]
        inst.scale, inst.frame = d["scale"], frame
        ctx.dyn_begin(); ctx.dyn_add_sprite(model, inst, view); ctx.dyn_end(2)
        g = P.Geo(); P.add_sprite(g, spr, d, vd)
        ge, re_ = assert_geo_equal(ctx.get_geometry(2), g.arrays())
        assert len(ge) == 4 and (ge["texnum_fb_flags"] >> 12 == 7).all() and (ge["texnum_alpha"] >> 12 == 0).all()  # two quads; MAT_FLAGS_SPRITE; texture alpha
        assert np.array_equal(ge["n0_gloss_norm"], re_["n0_gloss_norm"])


def test_alias_models_match_oracle(mq, tmp_path):
    rng = np.random.default_rng(9)
    path = str(tmp_path / "m.mdl")
    nposes = Q.write_mdl(path, rng)
    ctx = mq.Context(-1)
    model, nxt = ctx.load_mdl(path, 200)
    m = P.parse_mdl(open(path, "rb").read())
    assert len(m["poses"]) == nposes == 4 and nxt == 203  # skin 0 (+ its fullbright mask), skin 1 (a group: its first picture)
    skin_px, flags = ctx.get_texture(200)
    assert skin_px.shape == (16, 32, 4) and np.array_equal(skin_px[..., 0], m["skins"][0])  # grey-ramp palette: index = value
    fb_px, _ = ctx.get_texture(201)
    assert (fb_px[..., 0] > 0).sum() == (m["skins"][0] >= 224).sum()
    inst = mq.AliasInstance()
    d = dict(origin=[30, -20, 10], angles=[15, 200, 5], prev_origin=[28, -20, 10], prev_angles=[14, 195, 5], pose1=1, pose2=2, blend=0.3, prev_blend=0.1, skin=0, fovscale=0.0)
    for k in range(3):
        inst.origin[k], inst.angles[k], inst.prev_origin[k], inst.prev_angles[k] = d["origin"][k], d["angles"][k], d["prev_origin"][k], d["prev_angles"][k]
    inst.pose1, inst.pose2, inst.blend, inst.prev_blend, inst.skin, inst.fovscale = 1, 2, 0.3, 0.1, 0, 0.0
    ctx.dyn_begin(); ctx.dyn_add_alias(model, inst); ctx.dyn_end(2)
    got = ctx.get_geometry(2)
    g = P.Geo(); P.add_alias(g, m, d, [200, 202], [201, 0])
    ge, re_ = assert_geo_equal(got, g.arrays())
    assert len(got["idx"]) == 20 and (ge["n1_brush"] != 0xffffffff).all()
    assert (ge["texnum_fb_flags"] == 201).all() and (ge["texnum_alpha"] == (200 | (15 << 12))).all()
    inst.skin = 1; inst.fovscale = 1.3  # the view model: fov-independent gun (quake_helpers.cpp:244-246), second skin
    d.update(skin=1, fovscale=1.3)
    ctx.dyn_begin(); ctx.dyn_add_alias(model, inst); ctx.dyn_end(2)
    g = P.Geo(); P.add_alias(g, m, d, [200, 202], [201, 0])
    assert_geo_equal(ctx.get_geometry(2), g.arrays())
    inst.pose1 = 99
    ctx.dyn_begin(); ctx.dyn_add_alias(model, inst); ctx.dyn_end(2)
    assert ctx.get_geometry(2) is None  # a frame outside the model adds nothing (:241-243)


@pytest.mark.gpu
def test_bsp_scene_with_per_frame_entities_renders_like_the_oracle(mqlib, tmp_path):
    """A BSP-loaded map (liquids, sky, an alpha-tested fence, fullbright texels, external normal / gloss maps) with a
    moving door (brush model), an alias model, a sprite and particles re-emitted every frame: radiance, first-hit
    records, motion vectors and the g-buffer bit-identical to the oracle over four frames, guided frame included in
    reference mode only (deterministic)."""
    import mqhip
    import orc
    maps = tmp_path / "id1" / "maps"; tex = tmp_path / "id1" / "textures"
    maps.mkdir(parents=True); tex.mkdir()
    path = str(maps / "room.bsp")
    Q.write_bsp(path, False, rich=True)
    rng = np.random.default_rng(3)
    nm = np.full((8, 8, 4), 255, np.uint8); nm[..., 0] = rng.integers(100, 156, (8, 8)); nm[..., 1] = rng.integers(100, 156, (8, 8))
    Q.write_tga(str(tex / "wall1_norm.tga"), nm); Q.write_tga(str(tex / "wall1_gloss.tga"), rng.integers(60, 200, (8, 8, 4), dtype=np.uint8))
    Q.write_mdl(str(tmp_path / "m.mdl"), rng); Q.write_spr(str(tmp_path / "s.spr"), rng, 2)
    ctx = mqhip.Context(0)
    ctx.header_defaults()
    ctx.load_bsp(path)
    alias, nxt = ctx.load_mdl(str(tmp_path / "m.mdl"), 300)
    sprite, nxt = ctx.load_spr(str(tmp_path / "s.spr"), nxt)
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "reference mode": 1, "spp": 2, "max path length": 3, "adaptive grid buf size": 1 << 16, "static grid buf size": 1 << 12, "LC buf size": 1 << 14}.items():
        ctx.set_property(k, v)
    W, H = 160, 120
    o = None
    parts = np.zeros(12, mqhip.PARTICLE_DTYPE)
    parts["org"] = rng.uniform(60, 200, (12, 3)) * [1, 1, 0.4] + [0, 0, 20]; parts["vel"] = rng.uniform(-30, 30, (12, 3)); parts["seed"] = rng.integers(1, 2 ** 32, 12)
    parts["color_rgba"] = rng.choice([0x0000003c, 0x00ffffff, 0x0040a0ff], 12); parts["type"] = rng.choice([0, 3, 5], 12)
    lit = 0.0; moved = False
    for f in range(4):
        u = ctx.synth_camera(0)
        u.frame = f; u.cl_time = f / 60.0
        view, _ = view_of(mqhip, [u.cam_x[0], u.cam_x[1], u.cam_x[2]], [0, 90, 0])
        parts["prev_org"] = parts["org"]; parts["org"] = parts["org"] + parts["vel"] / 60.0
        ai = mqhip.AliasInstance(); si = mqhip.SpriteInstance()
        for k in range(3):
            ai.origin[k] = (128, 190 - 2 * f, 30)[k]; ai.prev_origin[k] = (128, 192 - 2 * f, 30)[k]; ai.angles[k] = (0, 10 * f, 0)[k]; ai.prev_angles[k] = (0, 10 * f - 10, 0)[k]
            si.origin[k] = (90, 180, 50 + f)[k]; si.prev_origin[k] = (90, 180, 49 + f)[k]
        ai.pose1, ai.pose2, ai.blend, ai.prev_blend = f % 3, (f + 1) % 3, 0.25 * f, 0.25 * max(f - 1, 0); si.scale = 1.0; si.frame = f % 2
        ctx.dyn_begin()
        ctx.dyn_add_brush_model(1, [0, 4.0 * f, 0], [0, 0, 0], [0, 4.0 * max(f - 1, 0), 0], [0, 0, 0])
        ctx.dyn_add_alias(alias, ai); ctx.dyn_add_sprite(sprite, si, view)
        ctx.dyn_add_particles(parts, view, 1, 2, u.cl_time, u.cl_time - 1 / 60.0)
        ctx.dyn_end(2)
        ctx.commit()
        if o is None:
            ctx.connect(W, H)
            o = orc.Oracle(orc.params_from_ctx(ctx, ctx.get_constants()))
            o.connect(W, H)
        orc.mirror_scene(ctx, o)
        o.commit(1)
        ctx.process(u); o.process(u, threads=8)
        img, ref = ctx.irradiance(), o.irradiance()
        bad = (img.view(np.uint32) != ref.view(np.uint32)).any(-1)
        assert not bad.any(), "frame %d: %d pixels differ, first %r" % (f, bad.sum(), np.argwhere(bad)[0])
        for g, r in ((mqhip.OUT_HITS, orc.OUT_HITS), (mqhip.OUT_GB_MV, orc.OUT_GB_MV), (mqhip.OUT_GBUFFER, orc.OUT_GBUFFER), (mqhip.OUT_GB_ALBEDO, orc.OUT_GB_ALBEDO), (mqhip.OUT_GB_IRRADIANCE, orc.OUT_GB_IRRADIANCE)):
            assert np.array_equal(ctx.read_output(g), o.output(r)), "frame %d output %d" % (f, g)
        lit += ref[..., :3].sum()
        moved = moved or bool(f and (ctx.read_output(mqhip.OUT_GB_MV) != 0).any())
    assert lit > 0 and moved
    ctx.close()


def test_per_frame_uniform_matches_oracle(mq, mqlib):
    """mq_uniform_update (QuakeNode::process, quake_node.cpp:768-824) over a sequence of frames: camera and previous
    camera, time difference (1 when the clock stands), fog coefficients from Quake's fog or from the override, sky texture
    numbers by sky mode, player flags; mq_constants_fov."""
    rng = np.random.default_rng(9)
    lib = mqlib if hasattr(mqlib, "mq_uniform_update") else mq.load_library()
    u = mq.Uniform()
    prev = dict(cam_x=[0, 0, 0, 0], cam_w=[0, 0, 0, 0], cam_u=[0, 0, 0, 0], cl_time=0.0)
    t = 0.0
    for frame in range(12):
        st = dict(vieworg=[float(x) for x in rng.uniform(-500, 500, 3)], viewangles=[float(x) for x in rng.uniform(-180, 180, 3)],
                  cl_time=t, frame=frame, render=int(frame != 3), has_player=int(frame % 4 != 1), weapon=int(rng.integers(1, 3)), waterlevel=int(rng.integers(0, 4)),
                  sky_mode=frame % 3, sky=[int(x) for x in rng.integers(1, 4000, 6)], notexture=7, mu_overwrite=int(frame % 5 == 4), mu_t=0.003,
                  mu_s_div_mu_t=[0.9, 0.8, 0.7], fog_density=float(rng.uniform(0, 0.3)), fog_color=[float(x) for x in rng.uniform(0, 1, 3)])
        s = mq.FrameState()
        for k, v in st.items():
            if isinstance(v, list):
                for i, x in enumerate(v):
                    getattr(s, k)[i] = x
            else:
                setattr(s, k, v)
        mq.uniform_update(lib, u, s)
        ref = P.uniform_update(prev, st)
        for name in ("cam_x", "cam_w", "cam_u", "prev_cam_x", "prev_cam_w", "prev_cam_u"):
            got = np.array(list(getattr(u, name)), np.float32)
            want = ref[name] if name != "cam_u" else np.array(list(ref[name][:3]) + [got[3]], np.float32)  # cam_u.w is not written (stays whatever it was)
            assert np.allclose(got, want, rtol=2e-6, atol=1e-7), (frame, name, got, want)
        assert abs(u.cl_time - ref["cl_time"]) <= 1e-6 * max(1.0, abs(ref["cl_time"]))
        assert u.frame == ref["frame"] and u.player == ref["player"]
        sky = [u.sky_rt_bk & 0xffff, u.sky_rt_bk >> 16, u.sky_lf_ft & 0xffff, u.sky_lf_ft >> 16, u.sky_up_dn & 0xffff, u.sky_up_dn >> 16]
        assert sky == ref["sky"], (frame, sky, ref["sky"])
        if frame > 0:
            assert u.cam_w[3] > 0
        prev = dict(cam_x=list(u.cam_x), cam_w=list(u.cam_w), cam_u=list(u.cam_u), cl_time=u.cl_time)
        t += 0.0 if frame == 5 else float(rng.uniform(0.005, 0.05))  # one frame with a standing clock: time difference 1
    k = mq.Constants()
    assert lib.mq_constants_fov(__import__("ctypes").byref(k), 90.0) == 0
    assert abs(k.fov_tan_alpha_half - 1.0) < 1e-6 and k.fov == 90.0


def test_alias_batch_equals_the_sequence_of_single_calls(mq, tmp_path):
    """mq_dyn_add_alias_batch (n entities on the worker pool: the reference's parallel_for over the visible entities,
    quake_node.cpp:904-938) leaves exactly what n calls of mq_dyn_add_alias leave -- vertices, previous vertices, indices,
    extra data, in the same order -- also behind other per-frame geometry and with an entity whose pose is out of range."""
    import quake_files as Q
    rng = np.random.default_rng(21)
    Q.write_mdl(str(tmp_path / "a.mdl"), rng, numverts=40, numtris=70)
    Q.write_mdl(str(tmp_path / "b.mdl"), rng, numverts=90, numtris=160, numframes=4)
    got = []
    for batched in (False, True):
        ctx = mq.Context(-1)
        ctx.header_defaults()
        ctx.synth_scene("synth_tiny", 2)
        ma, nxt = ctx.load_mdl(str(tmp_path / "a.mdl"), 300)
        mb, nxt = ctx.load_mdl(str(tmp_path / "b.mdl"), nxt)
        r2 = np.random.default_rng(5)
        models, insts = [], []
        for e in range(37):
            ai = mq.AliasInstance()
            for k in range(3):
                ai.origin[k] = float(r2.uniform(0, 300)); ai.prev_origin[k] = ai.origin[k] - float(r2.uniform(0, 2))
                ai.angles[k] = float(r2.uniform(-180, 180)); ai.prev_angles[k] = ai.angles[k] - 3.0
            ai.pose1, ai.pose2 = int(r2.integers(0, 3)), int(r2.integers(0, 3))
            ai.blend, ai.prev_blend, ai.skin, ai.fovscale = float(r2.random()), float(r2.random()), int(r2.integers(0, 2)), 1.0
            if e == 11:
                ai.pose2 = 99  # out of range: the entity adds nothing
            models.append(ma if e % 3 else mb); insts.append(ai)
        ctx.dyn_begin()
        view = mq.View(); view.forward[0] = 1.0; view.right[1] = -1.0; view.up[2] = 1.0
        parts = np.zeros(5, mq.PARTICLE_DTYPE); parts["org"] = r2.uniform(0, 100, (5, 3)); parts["seed"] = 7; parts["color_rgba"] = 0xffffff
        ctx.dyn_add_particles(parts, view, 1, 2, 0.5, 0.4)  # something in the slot before the entities
        if batched:
            ctx.dyn_add_alias_batch(models, insts)
        else:
            for m, ai in zip(models, insts):
                ctx.dyn_add_alias(m, ai)
        ctx.dyn_end(2)
        got.append(ctx.get_geometry(2))
    a, b = got
    assert len(a["idx"]) == 20 + 36 * 0 + sum((70 if e % 3 else 160) for e in range(37) if e != 11)
    for k in ("vtx", "prev_vtx", "idx", "ext"):
        assert a[k].tobytes() == b[k].tobytes(), k
