"""Section report of a -DMQ_PROF build (in-kernel lap profiling, see PLAP in csrc/mq_kernels.hip).

    make -C merian-quake_amd EXTRA=-DMQ_PROF B=build/prof OUT=lib/libmqhip_prof.so
    MQHIP_LIB=merian-quake_amd/lib/libmqhip_prof.so python tools/prof_sections.py [--width W --height H]

Prints the share of wave-clocks each code section of the frame kernels takes (clocks between laps
of a wave include the time its SIMD spent on other waves, so read the numbers as shares).  The laps are
intrusive (each is a scalar clock read the wave waits for): the traversal kernel of this build runs about
2.5x slower than the product's, the shading kernels 1.3-1.6x -- shares within a kernel, not times."""
import argparse, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "merian-quake_amd"))
import mqhip

SECTIONS = {
    0: "primary: loop overhead", 1: "primary: traversal", 2: "primary: shade first hit", 3: "primary: g-buffer stores",
    4: "advance: entry", 5: "advance: K state lookups", 6: "advance: sample direction", 7: "advance: pdf mixture",
    8: "advance: bsdf value", 9: "after advance (reconverge)", 10: "append + emit ray/path",
    35: "bounce: throughput update", 36: "bounce: termination test", 32: "finish: reach the branch", 33: "finish: contribution", 34: "finish: stores",
    16: "bounce: loop overhead", 17: "bounce: load path + hit", 18: "bounce: shade hit", 19: "bounce: light cache get",
    20: "bounce: light cache update", 21: "bounce: enqueue update",
    24: "trace: ballot", 25: "trace: refill", 26: "trace: ray fetch", 27: "trace: node phase", 28: "trace: triangle phase", 29: "trace: pop/writeback",
}
GROUPS = {"primary": [0, 1, 2, 3], "shared by primary+bounce": [35, 36, 4, 5, 6, 7, 8, 32, 33, 34, 9, 10], "bounce": [16, 17, 18, 19, 20, 21], "trace": [24, 25, 26, 27, 28, 29]}

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--width", type=int, default=1920); ap.add_argument("--height", type=int, default=1080)
    ap.add_argument("--scene", default="synth_sepulcher"); ap.add_argument("--frames", type=int, default=20)
    ap.add_argument("--world", type=int, default=1, help="profile rank 0 of a tile partition of this many ranks")
    a = ap.parse_args()
    ctx = mqhip.Context(0)
    ctx.json_defaults()
    for k, v in {"randomize seed": 0, "seed": 0x5EED, "spp": 1, "max path length": 3, "reference mode": 0, "volume spp": 0}.items():
        ctx.set_property(k, v)
    ctx.synth_scene(a.scene, 2); ctx.commit(); ctx.set_partition(0, a.world); ctx.connect(a.width, a.height)
    for f in range(10):
        ctx.process(ctx.synth_camera(f))
    ctx.section_clocks(reset=True)
    for f in range(10, 10 + a.frames):
        ctx.process(ctx.synth_camera(f))
    clk = ctx.section_clocks()
    # lane utilisation counters of the trace loop (not clocks)
    it, nb, n_exec, n_lanes, t_exec, t_lanes = clk[11], clk[30], clk[12], clk[13], clk[14], clk[15]
    dr_it, dr_lanes = clk[22], clk[23]
    if clk[31]:
        print(f"trace kernel: {clk[31]} waves, mean lifetime {clk[37] / clk[31]:.0f} clocks; section clocks per wave {sum(clk[24:30]) / clk[31]:.0f}")
    for i in (11, 12, 13, 14, 15, 22, 23, 30, 31, 37):
        clk[i] = 0
    if it:
        print(f"trace loop: {it} iterations, busy lanes {nb / it:.1f}/64; node phase ran in {100 * n_exec / it:.0f} % with {n_lanes / max(n_exec, 1):.1f} lanes; "
              f"triangle phase ran in {100 * t_exec / it:.0f} % with {t_lanes / max(t_exec, 1):.1f} lanes")
        print(f"   drain (queue exhausted, pool empty): {100 * dr_it / it:.0f} % of the iterations with {dr_lanes / max(dr_it, 1):.1f} busy lanes; "
              f"before the drain {(nb - dr_lanes) / max(it - dr_it, 1):.1f} busy lanes")
    hist = clk[40:104]; clk = clk[:40]
    if sum(hist):
        n = float(sum(hist)); acc = 0; marks = {}
        for b, h in enumerate(hist):
            acc += h
            for q in (0.5, 0.9, 0.99, 0.999):
                if q not in marks and acc >= q * n: marks[q] = 8 * (b + 1)
        mean = sum((8 * b + 4) * h for b, h in enumerate(hist)) / n
        print(f"   loop iterations per ray: mean {mean:.1f}, median <= {marks[0.5]}, 90 % <= {marks[0.9]}, 99 % <= {marks[0.99]}, 99.9 % <= {marks[0.999]}, last bin (>= 504): {hist[63]}")
    other = [i for i in range(40) if clk[i] and not any(i in ids for ids in GROUPS.values())]
    if other:
        print("sections outside the groups:", {i: clk[i] for i in other})
    tot = float(sum(clk)) or 1.0
    for g, ids in GROUPS.items():
        gs = sum(clk[i] for i in ids)
        print(f"{g}: {100 * gs / tot:.1f} % of all wave-clocks ({gs / 1e9:.2f} G clocks)")
        for i in ids:
            print(f"   {SECTIONS[i]:34s} {100 * clk[i] / tot:6.2f} %   ({100 * clk[i] / max(gs, 1):5.1f} % of group)")

if __name__ == "__main__":
    main()
