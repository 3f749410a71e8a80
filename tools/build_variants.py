"""Development builds of the library side by side (never loaded by default): python tools/build_variants.py name=-DFLAG[,-DFLAG2] ..."""
import importlib, os, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
b = importlib.import_module("ray-tracer_amd.build")
specs = [a.split("=", 1) for a in sys.argv[1:]]
with ThreadPoolExecutor(max_workers=6) as ex:
    for out in ex.map(lambda s: b.build_variant(s[0], [f for f in s[1].split(",") if f]), specs):
        print("built", out)
