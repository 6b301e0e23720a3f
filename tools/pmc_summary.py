"""Mean per-launch value of every counter in a rocprofv3 counter_collection.csv, per kernel.
    python tools/pmc_summary.py gpurun_out/<pass> [...]   ->  JSON on stdout"""
import collections, csv, glob, json, sys

def summarise(d):
    out = collections.defaultdict(dict)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        agg = collections.defaultdict(lambda: [0.0, 0])
        for r in csv.DictReader(open(f)):
            k = (r["Kernel_Name"].split("(")[0].replace("void ", ""), r["Counter_Name"])
            agg[k][0] += float(r["Counter_Value"]); agg[k][1] += 1
        for (kn, cn), (s, n) in agg.items():
            out[kn][cn] = {"mean": s / n, "launches": n}
    return out

if __name__ == "__main__":
    res = {}
    for d in sys.argv[1:]:
        for kn, v in summarise(d).items():
            res.setdefault(kn, {}).update(v)
    json.dump(res, sys.stdout, indent=1)
