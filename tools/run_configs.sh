#!/bin/bash
# Measures every BASELINE.json config that fits one MI355X (scratch tool; output -> gpurun_out/configs.log)
B="python bench.py --cpu-baseline 0"
show() { tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); r=j['roofline']; print('   value %.3e steps/s  ms/step %.1f  kernel %s  alg %.0f GB/s (frac %.3f)  record %s' % (j['value'], j['ms_per_step'], r['kernel'], r['achieved'], r['frac'], r.get('record_bytes')))"; }
echo "C2  RMAT-20 ef16 undirected p=1 q=1 L=80 (Mode R)";                     $B --scale 20 --steps 10 --warmup 2 2>/dev/null | show
echo "C3  RMAT-24 ef16 undirected WEIGHTED p=.25 q=4 L=80, Mode A";            $B --scale 24 --weighted 1 --p 0.25 --q 4 --sampler alias --steps 5 --warmup 1 2>/dev/null | show
echo "C3  same, Mode R (reference-exact), 1 iteration";                        $B --scale 24 --weighted 1 --p 0.25 --q 4 --steps 1 --warmup 0 2>/dev/null | show
echo "C4  RMAT-27 ef16 undirected p=1 q=1 L=80 on ONE GPU (Mode R)";          $B --scale 27 --steps 3 --warmup 1 2>/dev/null | show
echo "C5* RMAT-26 ef27 DIRECTED p=4 q=.5 L=80 (Friendster stand-in), Mode A";  $B --scale 26 --edge-factor 27 --directed 1 --p 4 --q 0.5 --sampler alias --steps 3 --warmup 1 2>/dev/null | show
echo "C5* same, Mode R (reference-exact), 1 iteration";                        $B --scale 26 --edge-factor 27 --directed 1 --p 4 --q 0.5 --steps 1 --warmup 0 2>/dev/null | show
echo "headline RMAT-26 ef16 undirected p=1 q=1 (Mode R)";                      $B --steps 10 --warmup 2 2>/dev/null | show
