import sys, os, time, ctypes as C
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import _pkg
pkg = _pkg.load()
lib = pkg.lib()
eng = pkg.Engine(0)
for mb in (64, 380, 380, 1024):
    p = C.c_void_p()
    t = time.time(); rc = lib.srw_host_alloc(C.c_size_t(mb << 20), C.byref(p)); dt = time.time() - t
    t = time.time(); lib.srw_host_free(p); df = time.time() - t
    print("pinned %4d MB: alloc %.1f ms  free %.1f ms (rc %d)" % (mb, dt * 1e3, df * 1e3, rc))
