#!/bin/bash
# Scratch: end-to-end CLI at BASELINE config 2 (RMAT-20 text edge list, numWalks 10, walkLength 80)
set -e
python - <<'PY'
import sys, time
sys.path.insert(0, "oracle")
import numpy as np, oracle_py
t = time.time()
s, d = oracle_py.rmat_edges(20, 16 << 20, seed=42)
a = np.empty((len(s), 2), dtype=np.int32); a[:, 0] = s; a[:, 1] = d
np.savetxt("/tmp/rmat20.txt", a, fmt="%d %d")
print("edge list written in %.1f s" % (time.time() - t))
PY
ls -la /tmp/rmat20.txt
rm -rf /tmp/out_c2
export SRW_TIMING=1
( time stellar-random-walk_amd/stellar-rw --cmd randomwalk --numWalks 10 --p 1 --q 1 --walkLength 80 --input /tmp/rmat20.txt --output /tmp/out_c2 ) 2>&1 | sort | uniq -c | sort -rn | head -20
ls -la /tmp/out_c2/path | head; head -c 300 /tmp/out_c2/path/part-00000; echo; wc -l /tmp/out_c2/path/part-00000
echo "--- same job with --deviceFormat false (host threads format) ---"
rm -rf /tmp/out_c2d
( time stellar-random-walk_amd/stellar-rw --cmd randomwalk --numWalks 10 --p 1 --q 1 --walkLength 80 --input /tmp/rmat20.txt --output /tmp/out_c2d --deviceFormat false ) 2>&1 | grep "timing\|real\|user\|sys"
cmp /tmp/out_c2/path/part-00000 /tmp/out_c2d/path/part-00000 && echo "host-formatted output identical"
