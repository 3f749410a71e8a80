"""Fits profiles/traffic.json entries from the FETCH_SIZE / WRITE_SIZE passes of tools/profile_all.sh.
usage: fit_traffic.py <prof dir> <tag>     (reads <prof dir>/hbm_<scene>_<W>x<H>_s<spp>_f<frames>_{fetch,write}/**/_counter_collection.csv)

One launch of f frames moves fixed + f * per_frame bytes; least squares over the measured f.  Units and the
gfx950 correction as MI355X_MICROARCH.md's HBM section prescribes: FETCH_SIZE / WRITE_SIZE are KiB;
FETCH_SIZE x 2 (this kernel's reads are the 16-byte-per-lane coalesced scene staging); WRITE_SIZE as reported
(12-byte-per-lane stores: uncalibrated width, stated as such)."""
import collections, csv, glob, json, os, re, sys

d, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pts = collections.defaultdict(lambda: collections.defaultdict(dict))      # scene -> frames -> {fetch, write}
for sub in sorted(glob.glob(os.path.join(d, "hbm_*"))):
    m = re.match(r"hbm_(\w+?)_(\d+x\d+_s\d+)_f(\d+)_(fetch|write)$", os.path.basename(sub))
    if not m:
        continue
    scene, frames, kind = m.group(1) + "_" + m.group(2) + "_l8", int(m.group(3)), m.group(4)
    total, last = collections.defaultdict(float), -1
    # (a scratch directory that has seen several runs holds one file per run: the newest is this run's)
    files = sorted(glob.glob(os.path.join(sub, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    for f in files[-1:]:
        rows = [r for r in csv.DictReader(open(f)) if "rt_render_kernel" in r["Kernel_Name"]]
        if not rows:
            continue
        last = max(int(r["Dispatch_Id"]) for r in rows)           # the 1-spp warm-up launch comes first
        for r in rows:
            if int(r["Dispatch_Id"]) == last:
                total[r["Counter_Name"]] += float(r["Counter_Value"])
    if kind == "fetch" and "FETCH_SIZE" in total:
        pts[scene][frames]["fetch"] = total["FETCH_SIZE"] * 1024 * 2
    if kind == "write" and "WRITE_SIZE" in total:
        pts[scene][frames]["write"] = total["WRITE_SIZE"] * 1024
path = os.path.join(ROOT, "profiles", "traffic.json")
tj = json.load(open(path))
for scene, byf in pts.items():
    xs, ys = [], []
    for f, v in sorted(byf.items()):
        if "fetch" in v and "write" in v:
            xs.append(f); ys.append(v["fetch"] + v["write"])
            print("%s f=%d: fetch %.2f MB + write %.2f MB = %.2f MB" % (scene, f, v["fetch"] / 1e6, v["write"] / 1e6, ys[-1] / 1e6))
    if not xs:
        continue
    if len(xs) == 1:
        fixed, per = 0.0, ys[0] / xs[0]
    else:
        n = len(xs); mx = sum(xs) / n; my = sum(ys) / n
        per = sum((x - mx) * (y - my) for x, y in zip(xs, ys)) / sum((x - mx) ** 2 for x in xs)
        fixed = my - per * mx
    key = scene
    tj[key] = {"fixed_bytes": fixed, "per_frame_bytes": per, "points": {str(x): y for x, y in zip(xs, ys)},
               "fetch_write": {str(f): v for f, v in sorted(byf.items())}, "profile": "profiles/%s/" % tag}
    print(key, "fixed %.3f MB, per frame %.3f MB" % (fixed / 1e6, per / 1e6))
json.dump(tj, open(path, "w"), indent=1)
