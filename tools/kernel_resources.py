"""Scratch: VGPR / SGPR / scratch / LDS of every kernel in a built library (llvm-objdump --offloading + llvm-readelf --notes)."""
import glob, os, re, subprocess, sys, tempfile, shutil
lib = os.path.abspath(sys.argv[1]); pat = sys.argv[2] if len(sys.argv) > 2 else ""
tmp = tempfile.mkdtemp(); dst = os.path.join(tmp, os.path.basename(lib)); shutil.copy(lib, dst)
llvm = "/opt/rocm/lib/llvm/bin/"
subprocess.run([llvm + "llvm-objdump", "--offloading", dst], capture_output=True)
for f in sorted(glob.glob(dst + ".*gfx950")):
    txt = subprocess.run([llvm + "llvm-readelf", "--notes", f], capture_output=True, text=True).stdout
    for blk in txt.split("- .agpr_count")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        name = g("name")
        if pat in name:
            print(name[:70], "vgpr", g("vgpr_count"), "sgpr", g("sgpr_count"), "scratch", g("private_segment_fixed_size"), "lds", g("group_segment_fixed_size"))
shutil.rmtree(tmp)
