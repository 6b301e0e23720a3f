"""CPU ORACLE (numpy) for the per-frame geometry producers -- TEST INFRASTRUCTURE, never imported by the product.

Restates, on plain inputs, what the reference builds from quakespasm's live structures:
  add_particles   src/game/quake_helpers.cpp:50-216     add_geo_alias   :218-359
  add_geo_brush   :362-469 (entity transform part)      add_geo_sprite  :471-626
and the two on-disk formats the product reads itself (id Software MDL "IDPO" v6, SPR "IDSP" v1).
PARITY UNPINNED: the reference has no fixtures for these and its quakespasm fork is an empty submodule; the
definitions for what it takes from absent code (merian::XORShift32, glm::rotate, the normal codec) are those of
DESIGN.md section 3.  Positions are compared with the product's within 1e-4 units (float32 arithmetic in another
order), integer and half-precision fields exactly.
"""
import struct

import numpy as np

EXT_DTYPE = np.dtype([("texnum_alpha", "<u2"), ("texnum_fb_flags", "<u2"), ("n0_gloss_norm", "<u4"), ("n1_brush", "<u4"), ("n2", "<u4"), ("st", "<u2", (6,))])
MAT_FLAGS_SPRITE, MAT_FLAGS_SOLID = 7, 8
PT_FIRE, PT_EXPLODE2 = 3, 5
F = np.float32


def half(x):
    return np.array(x, np.float32).astype(np.float16).view(np.uint16)


def encode_normal(n):
    n = np.asarray(n, np.float32)
    l1 = np.abs(n).sum(dtype=np.float32)
    px, py = n[0] / l1, n[1] / l1
    if n[2] < 0:
        px, py = (F(1) - abs(py)) * (F(1) if px >= 0 else F(-1)), (F(1) - abs(px)) * (F(1) if py >= 0 else F(-1))
    q = lambda v: int(np.floor(np.clip(F(v), -1, 1) * F(32767) + F(0.5)))
    return (q(px) & 0xffff) | ((q(py) & 0xffff) << 16)


def normalize(v):
    v = np.asarray(v, np.float32)
    l = np.sqrt((v * v).sum(dtype=np.float32))
    return v / l if l > 0 else v


class XorShift:
    def __init__(self, seed):
        self.s = seed & 0xffffffff or 1

    def next(self):
        s = self.s
        s ^= (s << 13) & 0xffffffff; s ^= s >> 17; s ^= (s << 5) & 0xffffffff
        self.s = s
        return (s >> 8) / 16777216.0


def angle_vectors(a):
    d2r = F(np.pi) / F(180)
    sy, cy, sp, cp, sr, cr = (np.float32(f(F(a[i]) * d2r)) for i, f in ((1, np.sin), (1, np.cos), (0, np.sin), (0, np.cos), (2, np.sin), (2, np.cos)))
    fwd = np.array([cp * cy, cp * sy, -sp], np.float32)
    right = np.array([-sr * sp * cy + cr * sy, -sr * sp * sy - cr * cy, -sr * cp], np.float32)
    up = np.array([cr * sp * cy + sr * sy, cr * sp * sy - sr * cy, cr * cp], np.float32)
    return fwd, right, up


def entity_matrix(origin, angles):
    f, r, u = angle_vectors(angles)
    return np.stack([f, -r, u], 1).astype(np.float32), np.asarray(origin, np.float32)


def rodrigues(v, axis, ang):
    c, s = F(np.cos(F(ang))), F(np.sin(F(ang)))
    return v * c + np.cross(axis, v).astype(np.float32) * s + axis * (np.dot(axis, v).astype(np.float32) * (F(1) - c))


def ext_rec(texnum_alpha, fb_flags, n0, n1, n2, st):
    e = np.zeros(1, EXT_DTYPE)
    e["texnum_alpha"], e["texnum_fb_flags"], e["n0_gloss_norm"], e["n1_brush"], e["n2"] = texnum_alpha, fb_flags, n0, n1, n2
    e["st"] = half(st)
    return e


def texnum_alpha(texnum, has_alpha):
    return min(texnum, 4095) | ((0 if has_alpha else 15) << 12)


class Geo:
    def __init__(self):
        self.vtx, self.prev, self.idx, self.ext = [], [], [], []

    def arrays(self):
        return (np.array(self.vtx, np.float32).reshape(-1, 3), np.array(self.prev, np.float32).reshape(-1, 3), np.array(self.idx, np.uint32).reshape(-1, 3),
                np.concatenate(self.ext) if self.ext else np.zeros(0, EXT_DTYPE))


def add_particles(g, parts, view_origin, view_forward, texnum_blood, texnum_explosion, cl_time, prev_cl_time):
    voff = np.array([[0, 1, 0], [-0.5, -0.5, 0.87], [-0.5, -0.5, -0.87], [1, -0.5, 0]], np.float32)
    tet = [0, 1, 2, 0, 2, 3, 0, 3, 1, 1, 3, 2]
    for p in parts:
        org, prev_org, vel = (np.asarray(p[k], np.float32) for k in ("org", "prev_org", "vel"))
        scale = np.dot(org - np.asarray(view_origin, np.float32), np.asarray(view_forward, np.float32)).astype(np.float32)
        scale = F(1) + F(0.08) if scale < 20 else F(1) + scale * F(0.004)
        scale = scale * F(0.5)
        c = int(p["color_rgba"]); cb = [c & 0xff, (c >> 8) & 0xff, (c >> 16) & 0xff, c >> 24]
        xr = XorShift(int(p["seed"]))
        texnum = texnum_fb = 0
        if cb[1] == 0 and cb[2] == 0 and cb[0] > 10:
            texnum = texnum_blood
        elif p["type"] == PT_EXPLODE2 or (p["type"] == PT_FIRE and not (cb[0] == cb[1] == cb[2])) or 0.299 * cb[0] + 0.587 * cb[1] + 0.114 * cb[2] > 200:
            texnum = texnum_fb = texnum_explosion; scale = scale * F(2)
        speed = np.sqrt((vel * vel).sum(dtype=np.float32))
        vert = prev_vert = None
        for _ in range(3):
            po = F(2 * (xr.next() - 0.5) + 2 * (xr.next() - 0.5))
            rand_angle = xr.next()
            axis = normalize([xr.next(), xr.next(), xr.next()])
            ang, pang = F((rand_angle + cl_time * 0.001 * float(speed)) * 2 * np.pi), F((rand_angle + prev_cl_time * 0.001 * float(speed)) * 2 * np.pi)
            vert, prev_vert = [], []
            for k in range(4):
                vo = F(0.5 * ((xr.next() - 0.5) + (xr.next() - 0.5)))
                rs = F(xr.next())
                local = (voff[k] * scale) * (F(1) + rs) + vo
                vert.append((org + po) + rodrigues(local, axis, ang)); prev_vert.append((prev_org + po) + rodrigues(local, axis, pang))
        base = len(g.vtx) // 3
        for k in range(4):
            g.vtx += list(vert[k]); g.prev += list(prev_vert[k])
        for k in range(4):
            i0, i1, i2 = tet[3 * k:3 * k + 3]
            g.idx += [base + i0, base + i1, base + i2]
            if texnum:
                enc = encode_normal(normalize(np.cross(vert[i2] - vert[i0], vert[i1] - vert[i0])))
                g.ext.append(ext_rec(texnum, texnum_fb, enc, enc, enc, [0, 1, 0, 0, 1, 0]))
            else:
                for _i in range(3):
                    cb[0] = int(min(255.0, max(0.0, cb[0] * (1 + xr.next() * 0.1 - 0.05))))
                cc = cb[0] | (cb[1] << 8) | (cb[2] << 16) | (cb[3] << 24)
                c_fb = cc if 0.299 * cb[0] + 0.587 * cb[1] + 0.114 * cb[2] > 150 else 0
                g.ext.append(ext_rec(0, MAT_FLAGS_SOLID << 12, cc, c_fb, 0, [0, 1, 0, 0, 1, 0]))


def add_sprite(g, spr, inst, view):
    fr = spr["frames"][min(max(inst["frame"], 0), len(spr["frames"]) - 1)]
    vpn, vright, vup, r_origin = (np.asarray(view[k], np.float32) for k in ("forward", "right", "up", "origin"))
    origin, prev_origin = np.asarray(inst["origin"], np.float32), np.asarray(inst["prev_origin"], np.float32)
    t = spr["type"]
    if t == 0:
        s_up = np.array([0, 0, 1], np.float32); s_right = normalize(np.cross(vpn, s_up))
    elif t == 1:
        f = origin - r_origin; f[2] = 0; f = normalize(f); s_right = np.array([f[1], -f[0], 0], np.float32); s_up = np.array([0, 0, 1], np.float32)
    elif t == 2:
        s_up, s_right = vup, vright
    elif t == 3:
        _, r, u = angle_vectors(inst["angles"]); s_up, s_right = u, r
    elif t == 4:
        a = F(inst["angles"][2]) * (F(np.pi) / F(180)); sr, cr = F(np.sin(a)), F(np.cos(a))
        s_right = vright * cr + vup * sr; s_up = vright * -sr + vup * cr
    else:
        return
    s_up, s_right = normalize(s_up), normalize(s_right)
    scale = F(inst["scale"]) if inst["scale"] > 0 else F(1)
    for k in range(2):
        sg = F(1 if k == 0 else -1)
        v = [(s_up * F(fr["down"]) + s_right * (sg * F(fr["left"]))) * scale, (s_up * F(fr["up"]) + s_right * (sg * F(fr["left"]))) * scale,
             (s_up * F(fr["up"]) + s_right * (sg * F(fr["right"]))) * scale, (s_up * F(fr["down"]) + s_right * (sg * F(fr["right"]))) * scale]
        base = len(g.vtx) // 3
        for p in v:
            g.vtx += list(p + origin); g.prev += list(p + prev_origin)
        g.idx += [base, base + 1, base + 2, base, base + 2, base + 3]
        enc = encode_normal(normalize(np.cross(v[2] - v[0], v[1] - v[0])))
        tn = texnum_alpha(fr["texnum"], True)
        g.ext.append(ext_rec(tn, MAT_FLAGS_SPRITE << 12, enc, enc, enc, [0, fr["tmax"], 0, 0, fr["smax"], 0]))
        g.ext.append(ext_rec(tn, MAT_FLAGS_SPRITE << 12, enc, enc, enc, [0, fr["tmax"], fr["smax"], 0, fr["smax"], fr["tmax"]]))


def parse_mdl(data):
    """-> dict with scale, scale_origin, skin size, poses (numposes, numverts, 4) uint8, VBO vertices (vertindex, st), indexes"""
    h = struct.unpack_from("<ii3f3ff3f8if", data, 0)
    assert h[0] == 0x4f504449 and h[1] == 6
    scale, origin = h[2:5], h[5:8]
    numskins, sw, sh, numverts, numtris, numframes = h[12:18]
    at = struct.calcsize("<ii3f3ff3f8if")
    skins = []
    for _ in range(numskins):
        (group,) = struct.unpack_from("<i", data, at); at += 4
        n = 1
        if group:
            (n,) = struct.unpack_from("<i", data, at); at += 4 + 4 * n
        skins.append(np.frombuffer(data, np.uint8, sw * sh, at).reshape(sh, sw)); at += sw * sh * n
    stv = np.frombuffer(data, "<i4", 3 * numverts, at).reshape(-1, 3); at += 12 * numverts
    tris = np.frombuffer(data, "<i4", 4 * numtris, at).reshape(-1, 4); at += 16 * numtris
    poses = []
    for _ in range(numframes):
        (t,) = struct.unpack_from("<i", data, at); at += 4
        n = 1
        if t:
            (n,) = struct.unpack_from("<i", data, at); at += 4 + 8 + 4 * n
        for _p in range(n):
            at += 8 + 16
            poses.append(np.frombuffer(data, np.uint8, 4 * numverts, at).reshape(-1, 4)); at += 4 * numverts
    vertindex, st, indexes, seen = [], [], [], {}
    for tr in tris:
        for k in range(3):
            vi = int(tr[1 + k]); s = float(stv[vi, 1]); t = float(stv[vi, 2])
            if not tr[0] and stv[vi, 0]:
                s += sw // 2
            key = (vi, s, t)
            if key not in seen:
                seen[key] = len(vertindex); vertindex.append(vi); st.append((s, t))
            indexes.append(seen[key])
    return dict(scale=np.array(scale, np.float32), scale_origin=np.array(origin, np.float32), skinwidth=sw, skinheight=sh, skins=skins, poses=np.array(poses),
                vertindex=np.array(vertindex), st=np.array(st, np.float32), indexes=np.array(indexes))


def add_alias(g, m, inst, skin_texnum, skin_fb_texnum):
    fov = np.array([1, inst["fovscale"] if inst["fovscale"] > 0 else 1, inst["fovscale"] if inst["fovscale"] > 0 else 1], np.float32)
    R, T = entity_matrix(inst["origin"], [-inst["angles"][0], inst["angles"][1], inst["angles"][2]])
    PR, PT = entity_matrix(inst["prev_origin"], [-inst["prev_angles"][0], inst["prev_angles"][1], inst["prev_angles"][2]])
    so, sc = m["scale_origin"] * fov, m["scale"] * fov
    base = len(g.vtx) // 3
    a = m["poses"][inst["pose1"]][m["vertindex"], :3].astype(np.float32); b = m["poses"][inst["pose2"]][m["vertindex"], :3].astype(np.float32)
    cur = (a * (F(1) - F(inst["blend"])) + b * F(inst["blend"])) * sc + so
    old = (a * (F(1) - F(inst["prev_blend"])) + b * F(inst["prev_blend"])) * sc + so
    world = (cur[:, :1] * R[:, 0] + cur[:, 1:2] * R[:, 1]) + cur[:, 2:3] * R[:, 2] + T
    pworld = (old[:, :1] * PR[:, 0] + old[:, 1:2] * PR[:, 1]) + old[:, 2:3] * PR[:, 2] + PT
    g.vtx += list(world.astype(np.float32).reshape(-1)); g.prev += list(pworld.astype(np.float32).reshape(-1))
    g.idx += list(base + m["indexes"])
    sk = min(max(inst["skin"], 0), len(skin_texnum) - 1)
    iw, ih = F(1) / F(m["skinwidth"]), F(1) / F(m["skinheight"])
    for t in range(len(m["indexes"]) // 3):
        i0, i1, i2 = m["indexes"][3 * t:3 * t + 3]
        p0, p1, p2 = world[i0].astype(np.float32), world[i1].astype(np.float32), world[i2].astype(np.float32)
        enc = encode_normal(normalize(np.cross(p2 - p0, p1 - p0)))
        st = [(m["st"][i][0] + F(0.5)) * iw if c == 0 else (m["st"][i][1] + F(0.5)) * ih for i in (i0, i1, i2) for c in (0, 1)]
        g.ext.append(ext_rec(texnum_alpha(skin_texnum[sk], False), skin_fb_texnum[sk], enc, enc, enc, st))


def add_brush_model(g, model_geo, origin, angles, prev_origin, prev_angles):
    vtx, idx, ext = model_geo
    R, T = entity_matrix(origin, [-angles[0], angles[1], angles[2]])
    PR, PT = entity_matrix(prev_origin, [-prev_angles[0], prev_angles[1], prev_angles[2]])
    base = len(g.vtx) // 3
    w = (vtx[:, :1] * R[:, 0] + vtx[:, 1:2] * R[:, 1]) + vtx[:, 2:3] * R[:, 2] + T
    pw = (vtx[:, :1] * PR[:, 0] + vtx[:, 1:2] * PR[:, 1]) + vtx[:, 2:3] * PR[:, 2] + PT
    g.vtx += list(w.astype(np.float32).reshape(-1)); g.prev += list(pw.astype(np.float32).reshape(-1))
    g.idx += list(base + idx.reshape(-1)); g.ext.append(ext)


def uniform_update(prev, st):
    """QuakeNode::process, quake_node.cpp:768-824: this frame's UniformData from the previous frame's (`prev`: dict with
    cam_x[4], cam_w[4], cam_u[4], cl_time) and the frame state `st` (dict, the fields of mq_frame_state).  float32 like the C++."""
    u = {}
    flags = 0
    if st["render"] and st["has_player"]:
        flags = (1 if st["weapon"] == 1 else 0) | (2 if st["waterlevel"] >= 3 else 0)
    u["player"] = flags
    u["frame"] = st["frame"]
    u["prev_cam_x"] = np.array(prev["cam_x"], np.float32); u["prev_cam_w"] = np.array(prev["cam_w"], np.float32); u["prev_cam_u"] = np.array(prev["cam_u"], np.float32)
    f, _, up = angle_vectors(st["viewangles"])
    u["cam_w"] = np.array([f[0], f[1], f[2], 0], np.float32); u["cam_u"] = np.array([up[0], up[1], up[2], prev["cam_u"][3]], np.float32)
    u["cam_x"] = np.array([st["vieworg"][0], st["vieworg"][1], st["vieworg"][2], 1], np.float32)
    sky = [st["notexture"]] * 6
    if st["render"] and st["sky_mode"] == 1:
        sky = list(st["sky"])
    elif st["render"] and st["sky_mode"] == 2:
        sky[0], sky[1], sky[2] = st["sky"][0], st["sky"][1], 0xffff
    u["sky"] = sky
    if st["mu_overwrite"]:
        mu_t = F(st["mu_t"])
        u["cam_x"][3] = mu_t
        ms = [F(F(st["mu_s_div_mu_t"][k]) * mu_t) for k in range(3)]
    else:
        mu_t = F(F(np.power(F(st["fog_density"]), F(2))) * F(0.1))
        u["cam_x"][3] = mu_t
        ms = [F(F(np.power(F(st["fog_color"][k]), F(1) / F(1.2))) * mu_t) for k in range(3)]
    u["prev_cam_x"][3], u["prev_cam_w"][3], u["prev_cam_u"][3] = ms
    dt = F(np.float64(st["cl_time"]) - np.float64(F(prev["cl_time"])))
    u["cam_w"][3] = dt if dt > 0 else F(1)
    u["cl_time"] = F(st["cl_time"])
    return u
