"""Condenses a tools/profile_all.sh output directory into the text summary kept under profiles/."""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
print("# rocprofv3 summary of", d)
for f in glob.glob(os.path.join(d, "stats", "*", "*_kernel_stats.csv")):
    print("\n## kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --steps 8 --warmup 8 --no-cpu-baseline --no-frame-by-frame-leg)")
    for i, row in enumerate(csv.reader(open(f))):
        if i < 6:
            print(",".join(row))
bj = os.path.join(d, "bench_under_rocprof.json")
if os.path.exists(bj):
    try:
        j = json.loads(open(bj).read().strip().splitlines()[-1])
        print("\n## bench.py line under rocprofv3: value %.1f %s, kernel_ms_avg (HIP events) %.3f" % (j["value"], j["unit"], j["roofline"]["kernel_ms_avg"]))
    except Exception as e:
        print("bench json unreadable:", e)
t = collections.defaultdict(float)
n = collections.defaultdict(int)
for f in sorted(glob.glob(os.path.join(d, "pmc*", "*", "*_counter_collection.csv"))):
    rows = [row for row in csv.DictReader(open(f)) if "rt_render_kernel" in row["Kernel_Name"]]
    if not rows:
        continue
    # tools/profile_run.py launches a 1-spp warm-up first (the first launch of a view also measures tile
    # costs); the profiled launch is the last dispatch of the kernel
    last = max(int(row["Dispatch_Id"]) for row in rows)
    for row in rows:
        if int(row["Dispatch_Id"]) == last:
            t[row["Counter_Name"]] += float(row["Counter_Value"])
            n[row["Counter_Name"]] += 1
print("\n## PMC counters, one launch of rt_render_kernel (monkey 1920x1080, 1024 spp, 8 bounces, 8 progressive frames in the launch), summed over the device")
for k in sorted(t):
    print("%-26s %.6g" % (k, t[k]))
if t.get("SQ_ACTIVE_INST_VALU"):
    print("\nlane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.3f" % (t["SQ_THREAD_CYCLES_VALU"] / (64 * t["SQ_ACTIVE_INST_VALU"])))
if t.get("FETCH_SIZE") is not None and t.get("WRITE_SIZE") is not None:
    # MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the
    # bytes of a wide coalesced streaming read (double it); WRITE_SIZE is exact for 16 B/lane stores.
    # This kernel's reads are the per-workgroup scene staging (16 B/lane, coalesced) -> doubled;
    # its writes are 12-byte-per-lane scattered stores (uncalibrated width, reported as is).
    fetch = t["FETCH_SIZE"] * 1024 * 2
    write = t["WRITE_SIZE"] * 1024
    print("HBM traffic per launch: fetch %.1f MB (2 x FETCH_SIZE KiB, gfx950 correction) + write %.1f MB = %.1f MB" % (fetch / 1e6, write / 1e6, (fetch + write) / 1e6))
    print(json.dumps({"hbm_bytes_per_launch": fetch + write, "fetch_bytes": fetch, "write_bytes": write}))
