"""Writers for small synthetic files in id Software's formats (BSP29 / BSP2, MDL, SPR) and TGA: test INPUTS.
No Quake data ships with the repository or the reference; these exercise the same loaders real maps would go through."""
import struct

import numpy as np


def _miptex(name, w, h, px):
    return struct.pack("<16sII4I", name, w, h, 40, 0, 0, 0) + bytes(px)


def write_bsp(path, bsp2=False, rich=False):
    """A box room (6 quads: wall texture with two fullbright texels, sky ceiling) with worldspawn sun keys.
    rich: + a water pool quad, a lava quad, a fence ('{' alpha-tested) quad in the room, and a door slab as brush model 1."""
    verts = [[x, y, z] for z in (0, 128) for y in (0, 256) for x in (0, 256)]
    quads = [(0, 1, 3, 2, 0), (4, 6, 7, 5, 1), (0, 4, 5, 1, 0), (2, 3, 7, 6, 0), (0, 2, 6, 4, 0), (1, 5, 7, 3, 0)]  # (v0..v3, texinfo)
    names = [b"wall1", b"sky1"]
    wall = [100] * (16 * 16); wall[5] = 250; wall[6] = 251
    pix = [(16, 16, wall), (32, 16, [7] * (32 * 16))]
    n_world = len(quads)
    if rich:
        def quad(p, eu, ev, ti):
            b = len(verts)
            verts.extend([p, [p[0] + eu[0], p[1] + eu[1], p[2] + eu[2]], [p[0] + eu[0] + ev[0], p[1] + eu[1] + ev[1], p[2] + eu[2] + ev[2]], [p[0] + ev[0], p[1] + ev[1], p[2] + ev[2]]])
            quads.append((b, b + 1, b + 2, b + 3, ti))
        names += [b"*water1", b"*lava1", b"{fence1", b"door1"]
        fence = [(255 if ((i // 64) + (i % 16) // 4) % 2 else 90) for i in range(16 * 16)]
        pix += [(16, 16, [40 + (i % 7) for i in range(256)]), (16, 16, [230 + (i % 5) for i in range(256)]), (16, 16, fence), (16, 16, [120 + (i % 9) for i in range(256)])]
        quad([40, 40, 8], [0, 80, 0], [80, 0, 0], 2)      # water surface, facing up
        quad([140, 40, 6], [0, 60, 0], [60, 0, 0], 3)     # lava, facing up
        quad([128, 200, 0], [0, 0, 100], [100, 0, 0], 4)  # fence, facing -y
        quad([128, 200, 0], [100, 0, 0], [0, 0, 100], 4)  # and its back side
        n_world = len(quads)
        for (p, eu, ev) in (([60, 120, 0], [0, 0, 96], [64, 0, 0]), ([124, 120, 0], [0, 0, 96], [-64, 0, 0]), ([60, 128, 0], [64, 0, 0], [0, 0, 96]), ([124, 112, 0], [-64, 0, 0], [0, 0, 96])):
            quad(p, eu, ev, 5)  # a door slab (4 of its faces): brush model 1
    edges, surfedges, faces = [(0, 0)], [], []
    for q in quads:
        first = len(surfedges)
        for k in range(4):
            edges.append((q[k], q[(k + 1) % 4])); surfedges.append(len(edges) - 1)
        faces.append((first, 4, q[4]))
    mips = [_miptex(n, w, h, px) for n, (w, h, px) in zip(names, pix)]
    ofs = 4 + 4 * len(mips); lump_tex = struct.pack("<i", len(mips)); acc = b""
    for m in mips:
        lump_tex += struct.pack("<i", ofs + len(acc)); acc += m
    lump_tex += acc
    texinfo = b"".join(struct.pack("<8fii", 1, 0, 0, 0, 0, 1, 0.5, 0, i, 0) if i < 2 or True else b"" for i in range(len(names)))
    # texture axes per face orientation would be per-texinfo in a real map; one set (s = x, t = y + z/2) is enough for a loader test
    if bsp2:
        lump_faces = b"".join(struct.pack("<iiiii4Bi", 0, 0, f[0], f[1], f[2], 0, 0, 0, 0, -1) for f in faces)
        lump_edges = b"".join(struct.pack("<II", *e) for e in edges)
    else:
        lump_faces = b"".join(struct.pack("<hhihh4Bi", 0, 0, f[0], f[1], f[2], 0, 0, 0, 0, -1) for f in faces)
        lump_edges = b"".join(struct.pack("<HH", *e) for e in edges)
    ents = b'{\n"classname" "worldspawn"\n"_sunlight" "8000"\n"_sunlight_color" "1 0.5 0.25"\n"_sun_mangle" "90 -45 0"\n}\n{\n"classname" "info_player_start"\n"origin" "128 128 24"\n"angle" "90"\n}\n\x00'
    model = struct.pack("<9f4iiii", 0, 0, 0, 256, 256, 128, 0, 0, 0, 0, 0, 0, 0, 0, 0, n_world)
    if rich:
        model += struct.pack("<9f4iiii", 60, 112, 0, 124, 128, 96, 0, 0, 0, 0, 0, 0, 0, 0, n_world, len(quads) - n_world)
    lumps = [ents, b"", lump_tex, np.array(verts, np.float32).tobytes(), b"", b"", texinfo, lump_faces, b"", b"", b"", b"", lump_edges,
             struct.pack("<%di" % len(surfedges), *surfedges), model]
    hdr_len = 4 + 15 * 8; body = b""; table = b""
    for l in lumps:
        table += struct.pack("<ii", hdr_len + len(body), len(l)); body += l + b"\x00" * ((-len(l)) % 4)
    with open(path, "wb") as f:
        f.write((b"BSP2" if bsp2 else struct.pack("<i", 29)) + table + body)


def write_tga(path, rgba, rle=False):
    h, w = rgba.shape[:2]
    hdr = struct.pack("<BBBHHBHHHHBB", 0, 0, 10 if rle else 2, 0, 0, 0, 0, 0, w, h, 32, 0x28)  # top-left origin, 8 alpha bits
    px = rgba[..., [2, 1, 0, 3]].astype(np.uint8).reshape(-1, 4)
    if not rle:
        body = px.tobytes()
    else:
        body = b""; i = 0
        while i < len(px):
            n = 1
            while i + n < len(px) and n < 128 and (px[i + n] == px[i]).all():
                n += 1
            if n > 1:
                body += bytes([0x80 | (n - 1)]) + px[i].tobytes(); i += n
            else:
                body += bytes([0]) + px[i].tobytes(); i += 1
    open(path, "wb").write(hdr + body)


def write_mdl(path, rng, numverts=14, numtris=20, numframes=3, skinw=32, skinh=16, group_frame=True):
    """A random closed-ish mesh with a seam, two skins (one a skin group), three frames (one a frame group of two poses)."""
    hdr = struct.pack("<ii3f3ff3f8if", 0x4f504449, 6, 0.25, 0.5, 0.125, -16.0, -8.0, 4.0, 30.0, 0.0, 0.0, 22.0, 2, skinw, skinh, numverts, numtris, numframes, 0, 0, 1.0)
    body = b""
    skin0 = rng.integers(0, 200, skinw * skinh, dtype=np.uint8); skin0[:7] = 240  # some fullbright texels
    body += struct.pack("<i", 0) + skin0.tobytes()
    body += struct.pack("<ii2f", 1, 2, 0.1, 0.2) + rng.integers(0, 200, 2 * skinw * skinh, dtype=np.uint8).tobytes()
    onseam = rng.integers(0, 2, numverts) * 0x20
    stv = np.stack([onseam, rng.integers(0, skinw // 2, numverts), rng.integers(0, skinh, numverts)], 1).astype("<i4")
    body += stv.tobytes()
    tris = np.concatenate([rng.integers(0, 2, (numtris, 1)), np.stack([rng.permutation(numverts)[:3] for _ in range(numtris)])], 1).astype("<i4")
    body += tris.tobytes()
    nposes = 0
    for f in range(numframes):
        n = 2 if (group_frame and f == 1) else 1
        if n == 1:
            body += struct.pack("<i", 0)
        else:
            body += struct.pack("<ii4B4B2f", 1, n, 0, 0, 0, 0, 255, 255, 255, 0, 0.1, 0.2)
        for _ in range(n):
            body += struct.pack("<4B4B16s", 0, 0, 0, 0, 255, 255, 255, 0, b"frame") + rng.integers(0, 256, (numverts, 4), dtype=np.uint8).tobytes(); nposes += 1
    open(path, "wb").write(hdr + body)
    return nposes


def write_spr(path, rng, sprite_type=2, frames=((16, 24), (8, 8))):
    hdr = struct.pack("<iiifiiifi", 0x50534449, 1, sprite_type, 20.0, 32, 32, len(frames), 0.0, 0)
    body = b""
    for w, h in frames:
        px = rng.integers(0, 256, w * h, dtype=np.uint8); px[::5] = 255  # transparent texels
        body += struct.pack("<i", 0) + struct.pack("<iiii", -w // 2, h // 2, w, h) + px.tobytes()
    open(path, "wb").write(hdr + body)
