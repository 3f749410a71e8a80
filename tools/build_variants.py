"""Development builds of the library side by side (never loaded by default): python tools/build_variants.py name=-DFLAG[,-DFLAG2] ... name@GITREV ..."""
import importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("ray-tracer_amd.build")


def build_at(rev, name):
    """the library as of a git revision (a temporary worktree), as ray-tracer_amd/libraytracer_amd_<name>.so: the A side of an A/B"""
    import subprocess, tempfile, shutil
    d = tempfile.mkdtemp(prefix="rt_wt_")
    subprocess.check_call(["git", "-C", ROOT, "worktree", "add", "--detach", d, rev], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    try:
        src = [os.path.join(d, "ray-tracer_amd", "csrc", f) for f in ("rt_kernel.hip", "rt_capi.cpp", "rt_host.cpp")]
        out = os.path.join(ROOT, "ray-tracer_amd", "libraytracer_amd_%s.so" % name)
        flags = [f if not f.startswith("-I") else "-I" + os.path.join(d, "include") for f in b.FLAGS]
        subprocess.check_call([b.hipcc()] + flags + src + ["-o", out])
        return out
    finally:
        subprocess.call(["git", "-C", ROOT, "worktree", "remove", "--force", d], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":
    # name=-DFLAG[,-DFLAG2]  builds the working tree with extra flags;  name@REV  builds a git revision
    specs = sys.argv[1:]
    def one(spec):
        if "@" in spec:
            name, rev = spec.split("@", 1)
            return build_at(rev, name)
        name, fl = spec.split("=", 1)
        return b.build_variant(name, [f for f in fl.split(",") if f])
    with ThreadPoolExecutor(max_workers=6) as ex:
        for out in ex.map(one, specs):
            print("built", out)
