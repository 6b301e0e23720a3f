"""The file loaders read what a user points them at (BSP29 / BSP2 maps, MDL alias models, SPR sprites, TGA normal / gloss
maps, palette lumps): a damaged file has to end in an error code, never in a read outside the file.  Valid files from the
test writers are truncated at every kind of position and hit with random byte and header-field mutations; every load
either succeeds or raises MqError.  Part of the CPU suite, so `tools/run_sanitized.sh` runs it under ASAN + UBSan."""
import os
import struct

import numpy as np
import pytest

import quake_files as Q


def _mutations(data, rng, n_random):
    n = len(data)
    cuts = sorted(set([0, 1, 3, 4, 7, 8, 16, 64, n // 3, n // 2, n - 64, n - 5, n - 1] + [int(x) for x in rng.integers(0, n, 24)]))
    for c in cuts:
        if 0 <= c < n:
            yield data[:c]
    for _ in range(n_random):
        b = bytearray(data)
        for _k in range(int(rng.integers(1, 6))):
            at = int(rng.integers(0, min(n, 4096)))  # headers and directories are where the structure is
            b[at] = int(rng.integers(0, 256))
        yield bytes(b)
    for _ in range(n_random):  # whole little-endian words set to extreme values
        b = bytearray(data)
        at = 4 * int(rng.integers(0, min(n, 2048) // 4))
        b[at:at + 4] = struct.pack("<i", int(rng.choice([-1, -2 ** 31, 2 ** 31 - 1, 0x7fffff00, 65536, 1 << 24, 0])))
        yield bytes(b)


def _try(mq, fn):
    try:
        fn()
    except mq.MqError:
        pass


@pytest.fixture(scope="module")
def mq(mqlib):
    import mqhip
    return mqhip


def test_damaged_alias_models_and_sprites_are_refused_not_crashed_on(mq, tmp_path):
    rng = np.random.default_rng(7)
    good_mdl, good_spr = str(tmp_path / "m.mdl"), str(tmp_path / "s.spr")
    Q.write_mdl(good_mdl, rng); Q.write_spr(good_spr, rng)
    ctx = mq.Context(-1)
    assert ctx.load_mdl(good_mdl, 100)[0] >= 0 and ctx.load_spr(good_spr, 200)[0] >= 0
    for name, good, load in (("mdl", good_mdl, ctx.load_mdl), ("spr", good_spr, ctx.load_spr)):
        data = open(good, "rb").read()
        bad = str(tmp_path / ("bad." + name))
        for i, m in enumerate(_mutations(data, rng, 150)):
            open(bad, "wb").write(m)
            _try(mq, lambda: load(bad, 300))
    ctx.close()


@pytest.mark.parametrize("bsp2", [False, True])
def test_damaged_maps_and_texture_files_are_refused_not_crashed_on(mq, tmp_path, bsp2):
    rng = np.random.default_rng(11 + int(bsp2))
    maps = tmp_path / "id1" / "maps"; tex = tmp_path / "id1" / "textures"
    maps.mkdir(parents=True); tex.mkdir()
    good = str(maps / "good.bsp")
    Q.write_bsp(good, bsp2, rich=True)
    px = rng.integers(0, 256, (8, 8, 4), dtype=np.uint8)
    Q.write_tga(str(tex / "wall1_norm.tga"), px); Q.write_tga(str(tex / "wall1_gloss.tga"), px, rle=True)
    ctx = mq.Context(-1)
    ctx.load_bsp(good)
    data = open(good, "rb").read()
    bad = str(maps / "bad.bsp")
    for m in _mutations(data, rng, 200):
        open(bad, "wb").write(m)
        _try(mq, lambda: ctx.load_bsp(bad))
    # damaged external maps beside an intact map: the map still loads (a broken texture file is skipped or refused)
    for which in ("wall1_norm.tga", "wall1_gloss.tga"):
        tdata = open(str(tex / which), "rb").read()
        for m in _mutations(tdata, rng, 60):
            open(str(tex / which), "wb").write(m)
            _try(mq, lambda: ctx.load_bsp(good))
        open(str(tex / which), "wb").write(tdata)
    # a damaged palette
    pal = str(tmp_path / "palette.lmp")
    for size in (0, 1, 767, 768, 769):
        open(pal, "wb").write(bytes(rng.integers(0, 256, size, dtype=np.uint8)))
        _try(mq, lambda: ctx.load_bsp(good, pal))
    ctx.close()


def test_damaged_graph_files_are_refused_not_crashed_on(mq):
    """mq_load_properties_json has its own small JSON reader (graph files are user-edited)."""
    import json
    from test_host_logic import REFERENCE_JSON
    rng = np.random.default_rng(5)
    text = json.dumps(REFERENCE_JSON, indent=1).encode()
    ctx = mq.Context(-1)
    assert ctx.load_properties_json(text.decode(), "render_markovchain") >= 0
    specials = [b'"', b"\\", b"{", b"}", b"[", b"]", b":", b",", b"\x00", b"\xff", b"e", b"-", b"1e999", b"\\u12", b"nul", b"tru"]
    for _ in range(600):
        b = bytearray(text)
        for _k in range(int(rng.integers(1, 5))):
            at = int(rng.integers(0, len(b)))
            if rng.random() < 0.5:
                b[at:at + 1] = specials[int(rng.integers(0, len(specials)))]
            elif rng.random() < 0.5:
                del b[at:at + int(rng.integers(1, 40))]
            else:
                b = b[:at]
        txt = bytes(b).split(b"\x00")[0].decode("latin-1")  # the C ABI takes a NUL-terminated string
        for node in ("render_markovchain", "gbuffer", "accum", "no such node"):
            _try(mq, lambda: ctx.load_properties_json(txt, node))
    ctx.close()
