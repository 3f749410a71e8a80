"""Same-box A/B of library builds (development tool): interleaves rounds over the given
libraries, each in its own child process, and prints the median 1/8-frame (critical path) and
full-frame times.   python tools/ab_libs.py 256 3 libA.so libB.so ..."""
import os, subprocess, sys, statistics
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
spp, rounds, libs = sys.argv[1], int(sys.argv[2]), sys.argv[3:]
res = {l: ([], []) for l in libs}
for r in range(rounds):
    for l in libs:
        env = dict(os.environ, RT_AMD_LIB=os.path.join(ROOT, "ray-tracer_amd", l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "critical_probe.py"), spp, "8:24:24"], env=env, capture_output=True, text=True, timeout=300).stdout
        line = [x for x in out.splitlines() if "of frame" in x][-1].split()
        res[l][0].append(float(line[line.index("frame") + 1])); res[l][1].append(float(line[line.index("frame", line.index("frame") + 1) + 1]))
for l in libs:
    print("%-32s 1/8 frame median %.1f ms (%s)   full frame median %.1f ms (%s)" % (l, statistics.median(res[l][0]), " ".join("%.0f" % v for v in res[l][0]),
          statistics.median(res[l][1]), " ".join("%.0f" % v for v in res[l][1])), flush=True)
