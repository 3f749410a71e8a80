"""The C snippets of INTEGRATION.md are compiled against include/rt_amd.h (syntax + types, gcc -fsyntax-only): round 3 shipped
a positional `rt_tile_spec` initialiser that put the tile count into a pointer member after a field was added, and a call
with an argument missing - nothing compiled the documentation."""
import os
import re
import subprocess

from conftest import ROOT

PRELUDE = r"""
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
#include "rt_amd.h"
/* what the surrounding prose of INTEGRATION.md names but does not declare */
static int32_t time_ms, n_frames, frame_num_in, tiles_x, tiles_y, cap, n, n_i, W, H, i;
static int32_t seeds[32], owner[1 << 16];
static uint32_t ids[1 << 16], costs[1 << 16], peaks[1 << 16], cost_of_every_tile[1 << 16], list_i[1 << 16], cost_i[1 << 16], peak_i[1 << 16];
static float *d_frame, *d_tiles_i;
static void *stream, *stream_i;
"""


def c_blocks():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    return re.findall(r"```c\n(.*?)```", text, re.S)


def test_integration_md_c_snippets_compile(tmp_path):
    blocks = c_blocks()
    assert len(blocks) >= 2
    src = PRELUDE
    for k, b in enumerate(blocks):
        lines = [l for l in b.splitlines() if not l.startswith("#include")]
        body = "\n".join(lines)
        if k > 0:
            # later blocks continue the first one's program (ctx, b, cam, rs, frame, frame_num ...)
            first = "\n".join(l for l in blocks[0].splitlines() if not l.startswith("#include"))
            body = first + "\n" + body
        src += "\nvoid snippet_%d(void)\n{\n%s\n}\n" % (k, body)
    f = tmp_path / "integration_snippets.c"
    f.write_text(src)
    r = subprocess.run(["gcc", "-std=gnu11", "-fsyntax-only", "-Wall", "-Wno-unused-variable", "-Wno-unused-but-set-variable", "-Werror=incompatible-pointer-types",
                        "-Werror=int-conversion", "-Werror=implicit-function-declaration", "-I" + os.path.join(ROOT, "include"), str(f)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr


def test_header_mentions_no_removed_kernels():
    h = open(os.path.join(ROOT, "include", "rt_amd.h")).read()
    assert "pooled kernel" not in h          # removed in round 3 (13d30ad)


def test_documented_knob_defaults_are_the_headers():
    """INTEGRATION.md §7 lists the scheduling knobs with their defaults; they must be the constants the library is built with"""
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    hdr = open(os.path.join(root, "ray-tracer_amd", "csrc", "rt_device_scene.h")).read()
    doc = open(os.path.join(root, "INTEGRATION.md")).read()
    d = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define RT_DEF_([A-Z_0-9]+)\s+(\d+)", hdr)}
    rows = {m.group(1): m.group(2).strip() for m in re.finditer(r"\| `(RT_AMD_[A-Z_]+)`(?:, `RT_AMD_[A-Z_]+`)? \| ([^|]+) \|", doc)}
    assert rows["RT_AMD_WORK_THRESHOLD"] == str(d["WORK_THRESHOLD"])
    assert rows["RT_AMD_READY_BREAK"] == str(d["READY_BREAK"])
    assert rows["RT_AMD_HIT_BREAK"] == str(d["HIT_BREAK"])
    assert rows["RT_AMD_DESCEND_KEEP"] == str(d["DESCEND_KEEP"])
    assert rows["RT_AMD_SHADE_BATCH"] == str(d["SHADE_BATCH"])
    assert rows["RT_AMD_HIT_LOW"].startswith("%d, %d (1024-thread workgroups) / %d" % (d["HIT_LOW"], d["MIX_BREAK_1024"], d["MIX_BREAK"]))
