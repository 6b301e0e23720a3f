"""Deterministic parity of every LEARNING WRITE of the path:

  * what a frame PROPOSES to write -- queued Markov-chain update records (mc.glsl:159-184), light-cache stores
    (light_cache.glsl:54-84), fast-recovery invalidations (mcpg.comp:175-178, volume.comp:226-229), distance-chain
    stores (volume.comp:201-215) -- logged by kernels and oracle from the same given state with the stores switched
    off, compared as sorted multisets, bit for bit;
  * the update pass itself (mq_link_kernel + mq_apply_kernel vs compute_updates.comp:56-124 restated in
    oracle/mq_oracle.c:apply_slot): the same per-slot ordered update lists on both sides, the whole Markov-chain
    table compared afterwards;
  * a free-running learning frame on the device, replayed by the oracle's rules from the device's own log.
"""
import os

import numpy as np
import pytest

import orc
from test_gpu_parity import SMALL, VOL, _copy_learned_state, make_pair

pytestmark = pytest.mark.gpu

KIND_UPDATE, KIND_LC, KIND_RECOVER, KIND_DIST = 1, 2, 3, 4
# The update pass of every slot also writes the STATIC grid cell of its position (compute_updates.comp:96-105); at the
# reference's cell width (25.3 units) a small scene has a handful of such cells and all slots of a frame interfere
# through them.  The update-pass tests use a fine static grid so that most slots are independent of each other.
FINE = {"adaptive grid buf size": 1 << 20, "static grid buf size": 1 << 18, "mc static width": 0.7, "LC buf size": 1 << 16}


@pytest.fixture(scope="module")
def gpu_ctx(mqlib):
    import mqhip
    ctx = mqhip.Context(0)
    yield ctx
    ctx.close()


def sort_rows(a):
    a = np.ascontiguousarray(a, np.uint32).reshape(-1, 16)
    return a[np.lexsort(a.T[::-1])]


def assert_logs_equal(got, ref, what):
    assert len(got) == len(ref), "%s: %d records on the device, %d in the oracle" % (what, len(got), len(ref))
    g, r = sort_rows(got), sort_rows(ref)
    bad = (g != r).any(1)
    assert not bad.any(), "%s: %d of %d records differ, first device %r oracle %r" % (what, bad.sum(), len(g), g[bad][0], r[bad][0])


def by_kind(log):
    return {k: log[log[:, 15] == k] for k in (KIND_UPDATE, KIND_LC, KIND_RECOVER, KIND_DIST)}


def gpu_table_as_oracle(gmc):
    """device Markov-chain table (64-byte states) -> the oracle's 52-byte layout"""
    o = np.zeros(len(gmc), orc.Oracle.MC_DTYPE)
    o["w_tgt"] = gmc["w_tgt"]; o["sum_w"] = gmc["sum_w"]; o["w_cos"] = gmc["w_cos"]; o["T"] = gmc["T"]; o["id"] = gmc["id"]
    o["N"] = (gmc["n_hash"] & 0xffff).astype(np.uint16); o["hash"] = (gmc["n_hash"] >> 16).astype(np.uint16); o["mv"] = gmc["mv"]
    return o


MC_FIELDS = ("id", "w_tgt", "sum_w", "w_cos", "mv", "T", "N", "hash")


def tables_equal_mask(a, b):
    eq = np.ones(len(a), bool)
    for f in MC_FIELDS:
        x, y = a[f], b[f]
        if x.dtype.kind == "f":
            x, y = x.view(np.uint32), y.view(np.uint32)
        e = x == y
        eq &= e if e.ndim == 1 else e.all(-1)
    return eq


def learned_pair(ctx, scene, seed, props, W, H, frames, first=0):
    """oracle learns `frames` sequential frames; the device renders one frame (so its tables exist) and gets the state"""
    o = make_pair(ctx, scene, seed, props, W, H)
    for f in range(frames):
        o.process(ctx.synth_camera(first + f), threads=1)
    for f in range(frames):  # the device's delay-1 inputs (previous volume depth) need the same history
        ctx.process(ctx.synth_camera(first + f))
    return o


@pytest.mark.parametrize("quirk_n16", [1, 0])
def test_surface_learning_writes_match_oracle(gpu_ctx, quirk_n16):
    """Every learning write the guided surface estimator proposes in one frame, from a given state: update records
    (position, weight, target, motion vector, normal, state id, slot), light-cache stores (cell, checksum, new
    irradiance, N, re-key), fast-recovery invalidations.  Both settings of the 16-bit N*N quirk (mc.glsl:26 with
    the uint16_t N of grid.h:19; on = reference behaviour), with states at and around the values where it bites."""
    ctx = gpu_ctx
    W, H = 128, 80
    props = {"reference mode": 0, "spp": 2, "max path length": 3, "quirk: 16-bit N*N": quirk_n16, **SMALL}
    o = learned_pair(ctx, "synth_start", 11, props, W, H, 5)
    omc = o.state(0)
    live = np.flatnonzero(omc["sum_w"] > 0)
    assert len(live) > 1000
    for k, n in enumerate((255, 256, 257, 511, 512, 768, 1023, 1024)):  # learned states at the wrap points of N*N
        omc["N"][live[k::16]] = n
    _copy_learned_state(ctx, o)
    ctx.set_property("debug: freeze learning", 1); ctx.set_property("debug: log learning writes", 1)
    try:
        o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
        assert o.params.quirk_n16_wrap == quirk_n16 and o.params.log_learning == 1
        for f in (5, 6):
            u = ctx.synth_camera(f)
            o.learn_log_reset(W * H * 16)
            ctx.process(u); o.process(u, threads=8)
            img, ref = ctx.irradiance(), o.irradiance()
            assert np.array_equal(img.view(np.uint32), ref.view(np.uint32)), "frame %d: radiance differs" % f
            glog, olog = by_kind(ctx.learn_log()), by_kind(o.learn_log())
            for kind, name in ((KIND_UPDATE, "update records"), (KIND_LC, "light-cache stores"), (KIND_RECOVER, "fast-recovery invalidations")):
                assert_logs_equal(glog[kind], olog[kind], "frame %d %s" % (f, name))
            assert len(olog[KIND_UPDATE]) > 300 and len(olog[KIND_LC]) > 5000 and len(olog[KIND_RECOVER]) > 10, {k: len(v) for k, v in olog.items()}
            assert (olog[KIND_LC][:, 1] == 1).sum() > 50  # re-keyed cells seeded from the coarser level
        # the quirk changes the lobes: the two settings must not render the same frame
        test_surface_learning_writes_match_oracle.frames = getattr(test_surface_learning_writes_match_oracle, "frames", {})
        test_surface_learning_writes_match_oracle.frames[quirk_n16] = ref.copy()
        fr = test_surface_learning_writes_match_oracle.frames
        if len(fr) == 2:
            assert not np.array_equal(fr[0], fr[1]), "the N*N quirk switch has no effect"
    finally:
        ctx.set_property("debug: freeze learning", 0); ctx.set_property("debug: log learning writes", 0)
        ctx.set_property("quirk: 16-bit N*N", 1)


LEARN_VARIANTS = [{"mc fast recovery": 0}, {"surf: use LC": 0}, {"adaptive grid type": "quadratic", "LC grid type": "quadratic"},
                  {"quirk: LC max(wo_p,10)": 0}, {"max path length": 5, "spp": 1}, {"adaptive grid prob": 0.0}, {"mc samples": 12}]


@pytest.mark.parametrize("variant", LEARN_VARIANTS, ids=[",".join("%s=%s" % kv for kv in v.items()) for v in LEARN_VARIANTS])
def test_surface_learning_writes_parameter_variants(gpu_ctx, variant):
    """The same multiset comparison of the proposed learning writes with the estimator's parameters away from their defaults."""
    ctx = gpu_ctx
    W, H = 128, 80
    props = {"reference mode": 0, "spp": 2, "max path length": 3, **SMALL, **variant}
    o = learned_pair(ctx, "synth_start", 11, props, W, H, 5)
    _copy_learned_state(ctx, o)
    ctx.set_property("debug: freeze learning", 1); ctx.set_property("debug: log learning writes", 1)
    try:
        o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
        u = ctx.synth_camera(5)
        o.learn_log_reset(W * H * 32)
        ctx.process(u); o.process(u, threads=8)
        assert np.array_equal(ctx.irradiance().view(np.uint32), o.irradiance().view(np.uint32))
        glog, olog = by_kind(ctx.learn_log()), by_kind(o.learn_log())
        for kind, name in ((KIND_UPDATE, "update records"), (KIND_LC, "light-cache stores"), (KIND_RECOVER, "fast-recovery invalidations")):
            assert_logs_equal(glog[kind], olog[kind], name)
        assert len(olog[KIND_UPDATE]) > 100 and len(olog[KIND_LC]) > 1000
        if variant.get("mc fast recovery", 1) == 0:
            assert len(olog[KIND_RECOVER]) == 0
    finally:
        ctx.set_property("debug: freeze learning", 0); ctx.set_property("debug: log learning writes", 0)
        ctx.header_defaults()


def test_volume_learning_writes_match_oracle(gpu_ctx):
    """The same for a frame with the single-scatter volume estimator: its Markov-chain updates (jittered pseudo-normal,
    volume.comp:218-224), invalidations (:226-229) and distance-chain stores (:201-215) join the surface pass's."""
    ctx = gpu_ctx
    W, H = 112, 72
    props = {"reference mode": 0, "spp": 1, "max path length": 3, **VOL, "volume forward project": 0}
    o = learned_pair(ctx, "synth_start_fog", 7, props, W, H, 5)
    ctx.set_property("debug: freeze learning", 1)
    o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
    try:
        u = ctx.synth_camera(5)
        ctx.process(u); o.process(u, threads=8)  # drains what each side's last volume pass queued
        _copy_learned_state(ctx, o, with_distance=True)
        ctx.set_property("debug: log learning writes", 1)
        o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
        u = ctx.synth_camera(6)
        o.learn_log_reset(W * H * 16)
        ctx.process(u); o.process(u, threads=8)
        assert np.array_equal(ctx.volume().view(np.uint32), o.volume().view(np.uint32))
        glog, olog = by_kind(ctx.learn_log()), by_kind(o.learn_log())
        for kind, name in ((KIND_UPDATE, "update records"), (KIND_LC, "light-cache stores"), (KIND_RECOVER, "invalidations"), (KIND_DIST, "distance-chain stores")):
            assert_logs_equal(glog[kind], olog[kind], name)
        assert len(olog[KIND_DIST]) > 100 and len(olog[KIND_UPDATE]) > 300, {k: len(v) for k, v in olog.items()}
    finally:
        ctx.set_property("debug: freeze learning", 0); ctx.set_property("debug: log learning writes", 0)


def with_ranks(records, cap=10):
    """arrival ranks in array order per slot; arrivals beyond the cap are dropped (mc.glsl:169-184)"""
    r = np.array(records, np.uint32).reshape(-1, 16)
    order = np.argsort(r[:, 14], kind="stable")
    r = r[order]
    slot = r[:, 14]
    start = np.r_[0, np.flatnonzero(slot[1:] != slot[:-1]) + 1]
    rank = np.arange(len(r)) - np.repeat(start, np.diff(np.r_[start, len(r)]))
    r[:, 13] = (r[:, 13] & 0xffff) | (rank.astype(np.uint32) << 16)
    r[:, 15] = 0
    return r[rank < cap]


def interfering_slots(touches):
    """slots that touch a table entry which a different slot touches too"""
    cells, inv = np.unique(touches[:, 1], return_inverse=True)
    lo = np.full(len(cells), np.iinfo(np.int64).max); hi = np.full(len(cells), -1, np.int64)
    np.minimum.at(lo, inv, touches[:, 0].astype(np.int64)); np.maximum.at(hi, inv, touches[:, 0].astype(np.int64))
    return np.unique(touches[(lo != hi)[inv], 0])


def drop_interfering(o, records, u, state0):
    """oracle dry run with a touch list: slots whose applications read or write a table entry that another slot's
    application also touches are removed (which of two racing slots wins is not defined on the device)"""
    t = o.apply_updates(records, u, want_touches=True)
    o.state(0)[:] = state0
    bad_slots = interfering_slots(t)
    keep = ~np.isin(records[:, 14], bad_slots)
    return records[keep], len(bad_slots)


def test_update_pass_matches_oracle(gpu_ctx):
    """mq_link_kernel + mq_apply_kernel against compute_updates.comp:56-124 (oracle apply_slot): both sides start from
    the same Markov-chain table and get the same queue contents -- the update records a real guided frame proposes,
    grouped by the slot they were sent to with their arrival ranks, plus groups of 1..10 records sent to live states
    whose ids they partly carry (the `same id: continue the chain` branch, :74-82, at every list length up to the
    cap).  Afterwards EVERY state of the table must be bit-identical: the exponentially weighted estimate, the
    reservoir pick, the stochastic writes into the static and the adaptive grid, T, N, hashes."""
    import mqhip
    ctx = gpu_ctx
    W, H = 128, 80
    props = {"reference mode": 0, "spp": 2, "max path length": 3, **FINE}
    o = learned_pair(ctx, "synth_start", 11, props, W, H, 5)
    _copy_learned_state(ctx, o)
    ctx.set_property("debug: freeze learning", 1); ctx.set_property("debug: log learning writes", 1)
    try:
        o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
        frames = []
        for f in range(5, 15):  # ~600 proposed updates per frame of this size
            u = ctx.synth_camera(f)
            o.learn_log_reset(W * H * 16)
            o.process(u, threads=8)
            frames.append(by_kind(sort_rows(o.learn_log()))[KIND_UPDATE])
        proposed = np.concatenate(frames)
        assert len(proposed) > 3000
    finally:
        ctx.set_property("debug: freeze learning", 0); ctx.set_property("debug: log learning writes", 0)
    o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
    rng = np.random.default_rng(5)
    rng.shuffle(proposed)  # arrival order is arbitrary: any order is a valid input
    half = len(proposed) // 2
    real = with_ranks(proposed[:half])
    # synthetic groups on live states that no real record addresses
    state0 = o.state(0).copy()
    live = np.setdiff1d(np.flatnonzero(state0["sum_w"] > 0), real[:, 14])
    rng.shuffle(live)
    rest, groups, at = proposed[half:].copy(), [], 0
    for k, slot in enumerate(live):
        n = 1 + k % 10
        if at + n > len(rest):
            break
        g = rest[at:at + n]; at += n
        g[:, 14] = slot
        own = rng.random(n) < 0.6
        g[own, 7] = state0["id"][slot]  # these continue the slot's chain, the others start new ones
        groups.append(g)
    synth = with_ranks(np.concatenate(groups))
    records = np.concatenate([real, synth])
    sizes = np.bincount(np.unique(records[:, 14], return_counts=True)[1], minlength=11)
    assert (sizes[1:] >= 10).all(), sizes  # every list length 1..10 occurs
    u2 = ctx.synth_camera(6)  # the pass draws its random numbers from (slot, frame, seed): one uniform for dry run and run
    records, n_bad = drop_interfering(o, records, u2, state0)
    assert n_bad < 0.25 * len(np.unique(records[:, 14])), n_bad
    assert np.array_equal(o.state(0), state0)
    sizes = np.bincount(np.unique(records[:, 14], return_counts=True)[1], minlength=11)
    assert (sizes[1:] >= 10).all(), sizes  # every list length 1..10 still occurs
    own = records[:, 7] == state0["id"][records[:, 14]]
    assert own.sum() > 1000 and (~own).sum() > 1000  # chains continued (compute_updates.comp:74-82) and chains started
    assert len(interfering_slots(o.apply_updates(records, u2, want_touches=True))) == 0
    o.state(0)[:] = state0
    o.apply_updates(records, u2)
    ctx.apply_updates(records, u2)
    got = gpu_table_as_oracle(ctx.state_read(0, len(state0)))
    ref = o.state(0)
    eq = tables_equal_mask(got, ref)
    assert eq.all(), "%d states differ after the update pass, first slot %d: device %r oracle %r" % ((~eq).sum(), np.argmax(~eq), got[~eq][0], ref[~eq][0])
    changed = ~tables_equal_mask(ref, state0)
    assert changed.sum() > 0.5 * len(np.unique(records[:, 14])), changed.sum()


@pytest.mark.parametrize("tables", ["fine", "reference-like"])
def test_free_running_learning_frame_replays_in_oracle(gpu_ctx, tables):
    """A guided frame on the device with learning ON, verified after the fact.  The frame is racy by design (SURVEY
    App. D.1): which updates a path proposes depends on what other paths have already invalidated or cached.  But the
    device logs what it queued, with arrival ranks, and the update pass can be run in the reference's own dispatch
    order ("debug: sequential update pass": ascending slot order, one slot after the other).  Then: from the device's
    own table before the frame, the oracle applies the logged invalidations (mcpg.comp:175-178) and the logged updates
    in rank order (compute_updates.comp:56-124) -- and EVERY state of the device's table after the frame must equal
    the oracle's.  Also: exactly the first ten arrivals of a slot are kept (mc.glsl:169-184), counters agree, and
    every light-cache cell ends the frame holding exactly one of the (irradiance, N) payloads the log says were stored
    there (light_cache.glsl:77-80): no torn cells.  "reference-like": the reference's coarse static grid, popular states exceed the cap."""
    import mqhip
    ctx = gpu_ctx
    W, H = 128, 80
    props = {"reference mode": 0, "spp": 4, "max path length": 3, **(FINE if tables == "fine" else SMALL)}
    o = learned_pair(ctx, "synth_start", 11, props, W, H, 10)
    _copy_learned_state(ctx, o)
    n_mc = len(o.state(0))
    before = gpu_table_as_oracle(ctx.state_read(0, n_mc))
    ctx.set_property("debug: log learning writes", 1); ctx.set_property("debug: sequential update pass", 1)
    ctx.enable_counters(True)
    try:
        u = ctx.synth_camera(10)
        ctx.process(u)
        log = by_kind(ctx.learn_log())
        cnt = ctx.counters()
    finally:
        ctx.set_property("debug: log learning writes", 0); ctx.set_property("debug: sequential update pass", 0)
        ctx.enable_counters(False)
    after = gpu_table_as_oracle(ctx.state_read(0, n_mc))
    upd = log[KIND_UPDATE]
    ranks = upd[:, 13] >> 16
    per_slot = np.bincount(upd[:, 14], minlength=n_mc)
    assert cnt["queue_overflow"] == 0 and len(upd) > 1000
    if tables != "fine":
        assert (per_slot > 10).sum() >= 3, "the frame never hit the cap"
    # ranks of a slot are exactly 0 .. n-1; the first ten are applied, the rest dropped
    order = np.lexsort((ranks, upd[:, 14]))
    s_sorted, r_sorted = upd[order, 14], ranks[order]
    start = np.r_[0, np.flatnonzero(s_sorted[1:] != s_sorted[:-1]) + 1]
    expect = np.arange(len(upd)) - np.repeat(start, np.diff(np.r_[start, len(upd)]))
    assert np.array_equal(r_sorted, expect)
    assert cnt["mc_updates_accepted"] == int(np.minimum(per_slot, 10).sum())
    assert cnt["mc_updates_dropped"] == int(np.maximum(per_slot.astype(np.int64) - 10, 0).sum())
    # replay: invalidations first (they happen in the bounce kernels, before the update pass), then the update pass
    o.set_params(orc.params_from_ctx(ctx, ctx.get_constants()))
    start_state = before.copy()
    start_state["sum_w"][log[KIND_RECOVER][:, 14]] = 0.0
    kept = upd[ranks < 10].copy()
    kept[:, 15] = 0
    o.state(0)[:] = start_state
    o.apply_updates(kept, u)
    ref = o.state(0)
    eq = tables_equal_mask(after, ref)
    assert eq.all(), "%d states differ, first slot %d: device %r oracle %r" % ((~eq).sum(), np.argmax(~eq), after[~eq][0], ref[~eq][0])
    assert (~tables_equal_mask(ref, before)).sum() > 1000
    # Light cache: a cell ends the frame holding what ONE logged store put there.  The device publishes the 8-byte
    # (irradiance, N) payload of a cell with one single-copy-atomic store and no lock (of N racing writers the last store
    # survives; the reference's try-lock, light_cache.glsl:59-64, keeps the first and cancels the rest): no cell may hold
    # the halves of two writers' payloads (round 2's 16-byte stores tore at dword granularity).
    lc_after = ctx.state_read(1, int(ctx.get_property("LC buf size")))
    lcl = log[KIND_LC]
    f_lo = lc_after["irr"][:, 0].astype(np.uint64) | (lc_after["irr"][:, 1].astype(np.uint64) << 16)
    f_hi = lc_after["irr"][:, 2].astype(np.uint64) | (lc_after["N"].astype(np.uint64) << 16)
    cell = lcl[:, 14].astype(np.uint64)
    lo_ok = np.isin((cell << 32) | f_lo[lcl[:, 14]], (cell << 32) | lcl[:, 2])
    hi_ok = np.isin((cell << 32) | f_hi[lcl[:, 14]], (cell << 32) | lcl[:, 3])
    assert lo_ok.all() and hi_ok.all(), "%d light-cache cells hold a dword nobody logged" % len(np.unique(lcl[~(lo_ok & hi_ok), 14]))
    whole = set(zip(lcl[:, 14].tolist(), (lcl[:, 2].astype(np.uint64) | (lcl[:, 3].astype(np.uint64) << 32)).tolist()))
    cells_logged = np.unique(lcl[:, 14])
    mixed = [c for c in cells_logged.tolist() if (c, int(f_lo[c] | (f_hi[c] << np.uint64(32)))) not in whole]
    assert len(mixed) == 0, (len(mixed), len(cells_logged))
    assert len(cells_logged) > 3000
    chk_ok = np.isin((cell << 32) | lc_after["hash"][lcl[:, 14]].astype(np.uint64), (cell << 32) | lcl[:, 0])
    assert chk_ok.all()  # and its key is one of the keys stored (or confirmed) there this frame


def test_state_dumps_carry_the_reference_statistics(gpu_ctx, tmp_path):
    """tools/dump_state.py: mc_dump.json / lc_dump.json / update_buffer_dump.json in the reference's formats
    (render_mcpg.cpp:322-416) with the statistics its notebooks query (scripts/duckdb queries.md:2-53): updates per slot
    and frame (last_update_count), light-cache lock successes and cancellations (the cache then runs the reference's
    try-lock).  The aggregates must agree with what the frame's own counters and log say."""
    import json
    import sys
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))
    import dump_state
    ctx = gpu_ctx
    W, H = 160, 96
    make_pair(ctx, "synth_start", 11, {"reference mode": 0, "spp": 2, "max path length": 3, "debug: LC lock statistics": 1, **SMALL}, W, H)
    try:
        for f in range(6):
            ctx.process(ctx.synth_camera(f))
        ctx.set_property("debug: log learning writes", 1)
        ctx.enable_counters(True)
        ctx.process(ctx.synth_camera(6))
        cnt = ctx.counters()
        log = by_kind(ctx.learn_log())
        summary = dump_state.dump(ctx, str(tmp_path), None)
    finally:
        ctx.enable_counters(False)
        ctx.set_property("debug: log learning writes", 0); ctx.set_property("debug: LC lock statistics", 0)
    mc = json.load(open(tmp_path / "mc_dump.json")); lc = json.load(open(tmp_path / "lc_dump.json")); ub = json.load(open(tmp_path / "update_buffer_dump.json"))
    assert len(mc) == SMALL["adaptive grid buf size"] and len(lc) == SMALL["LC buf size"]
    assert set(mc[0]) == {"id", "N", "hash", "w_cos", "sum_w", "w_tgt", "last_update_count", "tgt_change", "w_change", "cos_change"}
    assert set(lc[0]) == {"hash", "irr", "N", "update_succeeded", "update_canceled"}
    # the duckdb queries: light-cache lock outcome ...
    ok, cancel = sum(c["update_succeeded"] for c in lc), sum(c["update_canceled"] for c in lc)
    assert ok > 10000 and cancel > 0 and summary["lc_update_succeeded"] == ok and summary["lc_update_canceled"] == cancel
    # (frame 0 cancels every update -- the zeroed lock word equals the frame number, light_cache.glsl:59-64 -- so cancel > 0 even without contention)
    # ... and updates per slot: every slot the last frame's log addresses reports min(arrivals, 10); a slot keeps the count of the frame that last touched it
    upd = log[KIND_UPDATE]
    per_slot = np.bincount(upd[:, 14], minlength=SMALL["adaptive grid buf size"] + SMALL["static grid buf size"])
    touched = np.flatnonzero(per_slot[: SMALL["adaptive grid buf size"]])
    assert len(touched) > 200
    assert all(mc[int(sl)]["last_update_count"] == min(int(per_slot[sl]), 10) for sl in touched)
    assert sum(1 for m in mc if m["last_update_count"] > 0) >= len(touched)
    assert len(ub) == (per_slot > 0).sum() and sum(u["last_update_count"] for u in ub) == cnt["mc_updates_accepted"] == int(np.minimum(per_slot, 10).sum())
    assert all(len(u["ids"]) == 10 and u["update_count"] == 0 for u in ub)
