"""Condenses a tools/profile_all.sh output directory into the text summary kept under profiles/.
usage: summarize_profile.py <dir> [scenes...]"""
import collections, csv, glob, json, os, sys
d = sys.argv[1]
scenes = sys.argv[2:] or ["monkey", "three_sphere", "cube"]
print("# rocprofv3 summary of", d, "(collected with tools/profile_all.sh)")
for s in scenes:
    what = {"reference_scene0": "1000x800, 100 spp, 5 bounces (the reference's default workload)", "sphere50k": "1920x1080, 16 spp, 8 bounces (50,880 triangles: BVH in LDS, triangles from L2)",
            "soup6k": "1920x1080, 64 spp, 8 bounces (6,000 triangles: BVH in LDS, triangles from L2)"}.get(s, "1920x1080, 1024 spp, 8 bounces")
    print("\n" + "=" * 100 + "\n# scene %s, %s" % (s, what))
    for f in glob.glob(os.path.join(d, "stats_" + s, "**", "*_kernel_stats.csv"), recursive=True):
        print("\n## kernel stats (rocprofv3 --kernel-trace --stats -- python3 bench.py --config N --steps 20 --warmup 5 --no-cpu-baseline --no-frame-by-frame-leg)")
        for i, row in enumerate(csv.reader(open(f))):
            if i < 5:
                print(",".join(row))
    for f in glob.glob(os.path.join(d, "stats_" + s, "**", "*_kernel_trace.csv"), recursive=True):
        rows = [r for r in csv.DictReader(open(f)) if "rt_render_kernel" in r.get("Kernel_Name", "")]
        for r in rows:
            print("dispatch %s: %.3f ms  VGPR %s accum %s SGPR %s LDS %s scratch %s  grid %s wg %s" % (
                r.get("Dispatch_Id"), (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6, r.get("VGPR_Count"), r.get("Accum_VGPR_Count"),
                r.get("SGPR_Count"), r.get("LDS_Block_Size"), r.get("Scratch_Size"), r.get("Grid_Size_X"), r.get("Workgroup_Size_X")))
    bj = os.path.join(d, "bench_under_rocprof_%s.json" % s)
    if os.path.exists(bj):
        try:
            j = json.loads([l for l in open(bj).read().splitlines() if l.startswith("{")][-1])
            print("\n## bench.py line under rocprofv3: value %.1f %s, kernel_ms_avg (HIP events) %.3f over %s frames per launch" % (
                j["value"], j["unit"], j["roofline"]["kernel_ms_avg"], j["roofline"]["frames_per_launch"]))
        except Exception as e:
            print("bench json unreadable:", e)
    t = collections.defaultdict(float)
    for f in sorted(glob.glob(os.path.join(d, "pmc*_" + s, "**", "*_counter_collection.csv"), recursive=True)):
        rows = [row for row in csv.DictReader(open(f)) if "rt_render_kernel" in row["Kernel_Name"]]
        if not rows:
            continue
        # tools/profile_run.py launches a 1-spp warm-up first (the first launch of a view also measures tile
        # costs); the profiled launch is the last dispatch of the kernel
        last = max(int(row["Dispatch_Id"]) for row in rows)
        for row in rows:
            if int(row["Dispatch_Id"]) == last:
                t[row["Counter_Name"]] += float(row["Counter_Value"])
    if t:
        print("\n## PMC counters, one launch of rt_render_kernel (8 progressive frames in the launch), summed over the device")
        for k in sorted(t):
            print("%-26s %.6g" % (k, t[k]))
        for f in glob.glob(os.path.join(d, "pmc1_%s.log" % s)):
            print("run:", open(f).read().strip().splitlines()[-1])
    if t.get("SQ_ACTIVE_INST_VALU"):
        print("lane utilisation = SQ_THREAD_CYCLES_VALU / (64 * SQ_ACTIVE_INST_VALU) = %.3f" % (t["SQ_THREAD_CYCLES_VALU"] / (64 * t["SQ_ACTIVE_INST_VALU"])))
    if t.get("SQ_WAVE_CYCLES") and t.get("SQ_WAIT_ANY"):
        print("of wave cycles: issuing %.3f, s_waitcnt %.3f, issue-stalled %.3f" % (t["SQ_ACTIVE_INST_ANY"] / t["SQ_WAVE_CYCLES"], t["SQ_WAIT_ANY"] / t["SQ_WAVE_CYCLES"], t["SQ_WAIT_INST_ANY"] / t["SQ_WAVE_CYCLES"]))
    if t.get("SQ_LDS_IDX_ACTIVE"):
        print("LDS bank-conflict cycles / LDS-active cycles = %.3f" % (t["SQ_LDS_BANK_CONFLICT"] / t["SQ_LDS_IDX_ACTIVE"]))
    if t.get("SQ_INSTS_VALU") and t.get("SQ_WAVES"):
        print("wave-instructions per launch: VALU %.4g  SALU %.4g  LDS %.4g" % (t["SQ_INSTS_VALU"], t.get("SQ_INSTS_SALU", 0), t.get("SQ_INSTS_LDS", 0)))
    # HBM traffic: MI355X_MICROARCH.md §HBM: FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reads half the
    # bytes of a wide coalesced streaming read (double it); WRITE_SIZE is exact for 16 B/lane stores.  This
    # kernel's reads are the per-workgroup scene staging (16 B/lane, coalesced) -> doubled; its writes are
    # 12-byte-per-lane stores (uncalibrated width, reported as is).
    for fr in (1, 8, 20):
        v = {}
        for kind, cname in (("fetch", "FETCH_SIZE"), ("write", "WRITE_SIZE")):
            for f in glob.glob(os.path.join(d, "hbm_%s_1920x1080_s1024_f%d_%s" % (s, fr, kind), "**", "*_counter_collection.csv"), recursive=True):
                rows = [row for row in csv.DictReader(open(f)) if "rt_render_kernel" in row["Kernel_Name"] and row["Counter_Name"] == cname]
                if rows:
                    last = max(int(row["Dispatch_Id"]) for row in rows)
                    v[kind] = sum(float(row["Counter_Value"]) for row in rows if int(row["Dispatch_Id"]) == last)
        if "fetch" in v and "write" in v:
            fetch, write = v["fetch"] * 1024 * 2, v["write"] * 1024
            print("HBM traffic of a launch of %2d frame(s): fetch %.1f MB (2 x FETCH_SIZE KiB, gfx950 correction) + write %.1f MB = %.1f MB  (algorithmic: %.1f MB)" % (
                fr, fetch / 1e6, write / 1e6, (fetch + write) / 1e6, 24e-6 * 1920 * 1080 * fr))
