"""Same-box sweep of the kernel's scheduling knobs in THROUGHPUT mode (development tool): every setting is a
child process running one multi-frame launch (tools/profile_run.py), rounds interleaved over the settings.
   python tools/sweep_knobs.py <scene> <spp> <frames> <rounds> "RT_AMD_HIT_BREAK=32 RT_AMD_READY_BREAK=32" "..." ...
An empty string is the default setting; RT_AMD_LIB=... selects a development build."""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
scene, spp, frames, rounds, specs = sys.argv[1], sys.argv[2], sys.argv[3], int(sys.argv[4]), sys.argv[5:]
res = {s: [] for s in specs}
for r in range(rounds):
    for s in specs:
        env = dict(os.environ)
        for kv in s.split():
            k, v = kv.split("=", 1)
            env[k] = v
        try:
            out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "profile_run.py"), scene, spp, "1920", "1080", frames], env=env, capture_output=True, text=True, timeout=int(os.environ.get("RT_SWEEP_TIMEOUT", "600")))
        except subprocess.TimeoutExpired:
            print("TIMEOUT", s, flush=True)
            continue
        print("  .. %s: %s" % (s or "(default)", (out.stdout.strip().splitlines() or ["?"])[-1][-40:]), flush=True)
        m = re.search(r"launch: ([0-9.]+) ms", out.stdout)
        if not m:
            print("FAILED", s, out.stdout[-300:], out.stderr[-600:], flush=True)
            continue
        res[s].append(float(m.group(1)))
for s in specs:
    if res[s]:
        print("%-70s median %8.2f ms  (%s)" % (s or "(default)", statistics.median(res[s]), " ".join("%.1f" % v for v in res[s])), flush=True)
