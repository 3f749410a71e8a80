#!/bin/bash
# Register / scratch / LDS figures of every kernel in rt_kernel.hip, read from the gfx950 code object's
# metadata (what the driver's judge reads with llvm-readelf --notes).  usage: tools/kernel_meta.sh [extra hipcc flags]
# Also leaves the ISA in /tmp/rt_kernel.s for reading.
set -e
HERE=$(cd "$(dirname "$0")/.." && pwd)
SRC=$HERE/ray-tracer_amd/csrc/rt_kernel.hip
FLAGS="--offload-arch=gfx950 -O3 -ffp-contract=off -fno-slp-vectorize -std=c++17 -I$HERE/include"
hipcc $FLAGS "$@" --cuda-device-only -c $SRC -o /tmp/rt_kernel.bundle 2>/dev/null
/opt/rocm/lib/llvm/bin/clang-offload-bundler --unbundle --type=o --input=/tmp/rt_kernel.bundle --targets=hipv4-amdgcn-amd-amdhsa--gfx950 --output=/tmp/rt_kernel.co
hipcc $FLAGS "$@" --cuda-device-only -S $SRC -o /tmp/rt_kernel.s 2>/dev/null
/opt/rocm/lib/llvm/bin/llvm-readelf --notes /tmp/rt_kernel.co | python3 -c '
import re, sys
txt = sys.stdin.read()
for blk in txt.split("- .agpr_count:")[1:]:
    g = lambda k: (re.search(r"\." + k + r":\s*(\S+)", blk) or [None, "?"])[1]
    print("%-64s vgpr %3s spill %2s  sgpr %3s spill %2s  scratch %4s B  static-lds %s" % (g("name")[:64], g("vgpr_count"), g("vgpr_spill_count"), g("sgpr_count"), g("sgpr_spill_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size")))
'
