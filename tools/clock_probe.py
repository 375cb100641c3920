"""Scratch: headline kernel time per launch next to the clocks / power sampled from sysfs while it runs (one engine, one
allocation).  usage: clock_probe.py [launches] [scale]"""
import glob, os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import _pkg
pkg = _pkg.load()
n = int(sys.argv[1]) if len(sys.argv) > 1 else 40
sc = int(sys.argv[2]) if len(sys.argv) > 2 else 26
gap = int(sys.argv[3]) if len(sys.argv) > 3 else 1

def rd(p):
    try:
        return open(p).read()
    except Exception:
        return ""
devs = sorted(glob.glob("/sys/class/drm/card*/device"))
dev = next((d for d in devs if os.path.exists(d + "/pp_dpm_sclk")), devs[0] if devs else "")
hw = sorted(glob.glob(dev + "/hwmon/hwmon*"))
hw = hw[0] if hw else ""
print("sysfs device", dev, "hwmon", hw, "temps:", " ".join(rd(f).strip() for f in sorted(glob.glob(hw + "/temp*_label"))), flush=True)
def cur(txt):
    for l in txt.splitlines():
        if l.rstrip().endswith("*"):
            return l.split(":")[1].strip().rstrip("*").strip()
    return txt.strip().replace("\n", "|")[:40]
samples = []
stop = False
def poll():
    while not stop:
        samples.append((time.perf_counter(), cur(rd(dev + "/pp_dpm_sclk")), cur(rd(dev + "/pp_dpm_fclk")), cur(rd(dev + "/pp_dpm_mclk")),
                        cur(rd(dev + "/pp_dpm_socclk")), rd(hw + "/power1_average").strip() or rd(hw + "/power1_input").strip(),
                        "/".join(rd(f).strip()[:-3] for f in sorted(glob.glob(hw + "/temp*_input"))), rd(hw + "/freq1_input").strip()))
        time.sleep(0.01)
eng = pkg.Engine(0)
eng.generate_rmat(sc, 16 << sc, seed=42)
eng.walk(fetch=False, walk_length=80, num_walks=1, seed=42)
th = threading.Thread(target=poll); th.start()
rows = []
for it in range(n):
    t0 = time.perf_counter()
    st = eng.walk(fetch=False, walk_length=80, num_walks=1, first_walk=it + 1, seed=42)
    t1 = time.perf_counter()
    rows.append((t0, t1, st["kernel_ms"]))
    if gap and it % 10 == 9:
        time.sleep(0.5)          # an idle gap every ten launches
stop = True; th.join()
every = max(1, n // 60)
for t0, t1, ms in rows[::every]:
    ss = [s for s in samples if t0 <= s[0] <= t1]
    mid = ss[len(ss) // 2] if ss else ("",) * 8
    print("kernel_ms %.1f  sclk %s fclk %s mclk %s socclk %s power_uW %s temp %s freq1 %s (%d samples)" % (ms, mid[1], mid[2], mid[3], mid[4], mid[5], mid[6], mid[7], len(ss)), flush=True)
