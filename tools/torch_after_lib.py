"""Scratch: does torch still find the GPU after libstellar_rw.so has used it in the same process?"""
import os, sys, ctypes
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
stage = sys.argv[1]
eng = pkg.Engine(0)
if stage >= "b":
    eng.generate_rmat(16, 16 << 16, seed=42)
    eng.walk(walk_length=10, seed=1)
if stage >= "c":
    eng.walk(walk_length=10, seed=1, p=0.25, q=4.0)
if stage >= "d":
    eng.walk(walk_length=10, seed=1, p=0.25, q=4.0, sampler="alias")
hip = ctypes.CDLL("libamdhip64.so")
n = ctypes.c_int(-1)
rc = hip.hipGetDeviceCount(ctypes.byref(n))
print("stage", stage, "hipGetDeviceCount rc", rc, "n", n.value, "last error", hip.hipGetLastError(), flush=True)
import torch
try:
    torch.cuda.init()
    print("stage", stage, "torch ok", torch.cuda.device_count(), flush=True)
except Exception as e:
    print("stage", stage, "torch FAILED:", e, flush=True)
