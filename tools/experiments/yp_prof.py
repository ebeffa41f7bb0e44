"""Phase times of yolo_post_kernel (scratch build with ADAS_CFLAGS=-DADAS_YP_PROF)."""
import ctypes as C, importlib, os, sys, json, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import load_pkg
load_pkg()
L = importlib.import_module("adas_amd._lib")
sys.argv = ["bench.py", "--no-cpu-baseline", "--steps", "20"] + sys.argv[1:]
import runpy
try:
    runpy.run_path(os.path.join(ROOT, "bench.py"), run_name="__main__")
except SystemExit:
    pass
out = (C.c_ulonglong * 8)()
lib = C.CDLL(os.path.join(ROOT, "vehicle-cv-adas_amd", "libadas_hip.so"))
lib.adas_debug_yolo_prof(out)
tot = sum(out[:4])
for n, v in zip(("compact", "boxes", "nms", "gather"), out[:4]):
    print("%-8s %5.1f %%  (%.0f ticks of 10 ns)" % (n, 100.0 * v / max(1, tot), v))
