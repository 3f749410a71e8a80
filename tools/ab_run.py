"""Same-box A/B of library builds (development tool): python tools/ab_run.py "monkey:256:8,cube:256:8" base l1 d2 [--rounds 2] [--check]
Each case is scene:spp:frames (one multi-frame launch, tools/profile_run.py); libraries are ray-tracer_amd/libraytracer_amd_<name>.so.
--check first renders monkey 96x64x8spp x 3 frames with every library and compares the frames bit for bit with the first one's."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = [a for a in sys.argv[1:] if not a.startswith("--")]
rounds = int(sys.argv[sys.argv.index("--rounds") + 1]) if "--rounds" in sys.argv else 2
if "--rounds" in sys.argv:
    args.remove(sys.argv[sys.argv.index("--rounds") + 1])
cases, libs = args[0].split(","), args[1:]
def lib(l):
    return os.path.join(ROOT, "ray-tracer_amd", "libraytracer_amd_%s.so" % l)
if "--check" in sys.argv:
    ref = None
    for l in libs:
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "frame_hash.py")], env=dict(os.environ, RT_AMD_LIB=lib(l)), capture_output=True, text=True, timeout=600)
        h = out.stdout.strip().splitlines()[-1] if out.stdout.strip() else "FAILED " + out.stderr[-300:]
        ref = ref or h
        print("check %-12s %s %s" % (l, h, "" if h == ref else "  <-- DIFFERS"), flush=True)
res = {}
for r in range(rounds):
    for c in cases:
        sc, spp, fr = c.split(":")
        for l in libs:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_run.py"), sc, spp, "1920", "1080", fr], env=dict(os.environ, RT_AMD_LIB=lib(l)),
                                 capture_output=True, text=True, timeout=600).stdout.strip().splitlines()
            ms = float(out[-1].split("launch:")[1].split("ms")[0]) if out and "launch:" in out[-1] else float("nan")
            res.setdefault((c, l), []).append(ms)
for c in cases:
    base = statistics.median(res[(c, libs[0])])
    for l in libs:
        v = res[(c, l)]
        print("%-22s %-12s median %9.2f ms  (%s)  %+.2f %%" % (c, l, statistics.median(v), " ".join("%.1f" % x for x in v), 100.0 * (statistics.median(v) / base - 1.0)), flush=True)
