"""Fits the per-pixel cost weights (RT_COST_STEP / _GEN / _HIT, rt_device_scene.h) to measured rank times:
least squares of  ms_r = a * steps_r + b * gens_r + c * hits_r + d  over every rank of every run in a
scaling_probe dump, with the per-tile component maps of tools/cost_maps.py.  CPU only.

    fit_cost_weights.py cost_maps.npz probe_dump.json [more dumps...]
"""
import json, sys
import numpy as np
maps = np.load(sys.argv[1])
S, G, Hh = (maps[k].astype(np.float64) for k in ("step", "gen", "hit"))
rows, ms, tags = [], [], []
for path in sys.argv[2:]:
    d = json.load(open(path))
    for n, byp in d["runs"].items():
        if int(n) < 2:
            continue
        for part, run in byp.items():
            for r, (t, ids) in enumerate(zip(run["ms"], run["lists"])):
                ids = np.asarray(ids, np.int64)
                rows.append([S[ids].sum(), G[ids].sum(), Hh[ids].sum(), 1.0])
                ms.append(t)
                tags.append("N=%s %s r%d" % (n, part, r))
A, y = np.asarray(rows), np.asarray(ms)
scale = A.max(axis=0)
x, res, rank, sv = np.linalg.lstsq(A / scale, y, rcond=None)
x = x / scale
pred = A @ x
print("ms = %.4g * steps + %.4g * gens + %.4g * hits + %.3g" % tuple(x))
print("relative weights (step = 4): step 4, gen %.2f, hit %.2f" % (4 * x[1] / x[0], 4 * x[2] / x[0]))
err = (pred - y) / y
print("fit error: rms %.2f%%, worst %+.2f%% (%s)" % (100 * np.sqrt((err ** 2).mean()), 100 * err[np.abs(err).argmax()], tags[int(np.abs(err).argmax())]))
for w in ((4, 0, 0), (4, 3, 8), (4, round(4 * x[1] / x[0]), round(4 * x[2] / x[0]))):
    c = A[:, :3] @ np.asarray(w, float)
    k, off = np.polyfit(c, y, 1)
    e = (k * c + off - y) / y
    print("weights %s: one-parameter fit rms %.2f%%, worst %+.2f%%" % (w, 100 * np.sqrt((e ** 2).mean()), 100 * e[np.abs(e).argmax()]))
