"""Instruction counts of the render kernel's hot loops, from the compiler's assembly (development tool, CPU only):
   python tools/isa_loops.py [-DFLAG ...]      prints, for rt_render_kernel<1024,true,1>, the basic-block span of the descend loop
   (the one holding the stack's ds_write_b64) and of the leaf loop (the one holding v_div_fixup), by instruction class."""
import os, re, subprocess, sys, collections
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
flags = [a for a in sys.argv[1:] if a.startswith("-") and a != "-v"]
out = "/tmp/isa/rk_%s.s" % (re.sub(r"[^A-Za-z0-9]+", "_", "".join(flags)) or "base")
os.makedirs("/tmp/isa", exist_ok=True)
subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-slp-vectorize", "-std=c++17", "-I" + os.path.join(ROOT, "include"),
                       "-S", "--cuda-device-only", os.path.join(ROOT, "ray-tracer_amd", "csrc", "rt_kernel.hip"), "-o", out] + flags, stderr=subprocess.DEVNULL)
txt = open(out).read()
name = "_Z16rt_render_kernelILi1024ELb1ELi1EEv14rt_kernel_args"
body = txt[txt.index(name + ":"):]
body = body[:body.index("s_endpgm")]
lines = [l for l in body.splitlines() if l.strip() and not l.strip().startswith(";")]
def cls(op):
    if op.startswith("v_"): return "VALU"
    if op.startswith("ds_"): return "LDS"
    if op.startswith("s_cbranch") or op == "s_branch": return "BRANCH"
    if op.startswith("s_waitcnt") or op.startswith("s_nop"): return "WAIT/NOP"
    if op.startswith("s_"): return "SALU"
    if op.startswith("global_") or op.startswith("buffer_") or op.startswith("flat_"): return "VMEM"
    return "OTHER"
# loops: label L ... backward branch to L ; find innermost loop containing the marker
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
def loop_around(marker_idx):
    best = None
    for i, l in enumerate(lines):
        m = re.search(r"s_(?:cbranch_\w+|branch)\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] <= marker_idx <= i and labels[m.group(1)] < i:
            span = (labels[m.group(1)], i)
            if best is None or span[1] - span[0] < best[1] - best[0]: best = span
    return best
for what, marker in (("descend loop", "ds_write_b64"), ("leaf loop", "v_div_fixup_f32")):
    idxs = [i for i, l in enumerate(lines) if marker in l]
    # the LAST occurrence is in the traversal section for both
    span = loop_around(idxs[-1])
    # extend the descend loop's span backwards over latch blocks that branch forward into it
    c = collections.Counter(); ops = collections.Counter()
    for l in lines[span[0]:span[1] + 1]:
        if re.match(r"^\.LBB", l): continue
        op = l.split()[0]
        c[cls(op)] += 1; ops[op] += 1
    print("%s: lines %d..%d  total %d  %s" % (what, span[0], span[1], sum(c.values()), dict(c)))
    if "-v" in sys.argv: print("   ", dict(ops))
m = re.search(name + r".*?\.vgpr_count:\s+(\d+)", txt[txt.index(".amdgpu_metadata"):], re.S)
for key in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count"):
    mm = re.search(r"\.name:\s+" + name + r"\b.*?", txt, re.S)
meta = txt[txt.index(".amdgpu_metadata"):]
blk = meta[meta.index(name):]
blk2 = meta[:meta.index(name)]
# metadata lists fields before/after .name alphabetically; grab the enclosing kernel record
start = blk2.rfind("- .agpr_count"); end = meta.index(name) + blk.index("- .agpr_count") if "- .agpr_count" in blk else len(meta)
rec = meta[start:end]
print({k: int(re.search(r"\." + k + r":\s+(\d+)", rec).group(1)) for k in ("vgpr_count", "sgpr_count", "vgpr_spill_count", "sgpr_spill_count")})
