#!/bin/bash
# Scratch: config 3 (or "$1") under different table plans — table build seconds against milliseconds per iteration.
# usage: tools/eb_plan_sweep.sh [SPEC P Q [EF]]   (run from the repo root on the GPU box; output -> gpurun_out/eb_sweep.log)
SPEC=${1:-24w}; P=${2:-0.25}; Q=${3:-4}; EF=${4:-16}
mkdir -p gpurun_out
run() { echo "=== $*" ; env SRW_TIMING=1 "$@" python tools/one_walk.py $SPEC $P $Q reference 3 $EF 2>&1 | grep -v "table plan: [0-9]* chunks" ; }
{
run SRW_X=default
run SRW_EB_FINE_CAP=0
run SRW_EB_FINE_CAP=0 SRW_EB_CM_MAX=0
run SRW_EB_CHUNKS=64
run SRW_EB_CHUNKS=64 SRW_EB_FINE_CAP=0
run SRW_EB_CHUNKS=64 SRW_EB_FINE_CAP=0 SRW_EB_CM_MAX=0
run SRW_EB_CHUNKS=64 SRW_EB_FINE_CAP=0 SRW_EB_CM_MAX=0 SRW_EB_MIN_SH=8
run SRW_X=default_again
} > gpurun_out/eb_sweep_$SPEC.log 2>&1
