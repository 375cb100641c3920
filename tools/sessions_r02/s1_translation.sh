set -x
cd /tmp && export TMPDIR=/tmp
mkdir -p $GRAFT_REPO_ROOT/gpurun_out/s1
hipcc --offload-arch=gfx950 -O3 $GRAFT_REPO_ROOT/tools/microbench_translation.hip -o /tmp/mbt && /tmp/mbt > $GRAFT_REPO_ROOT/gpurun_out/s1/mbt.txt 2>&1
rocprofv3 -L > $GRAFT_REPO_ROOT/gpurun_out/s1/counters_all.txt 2>&1
grep -i -E "utcl|tlb|translat|xnack|TCP_.*MISS|TCC_.*(REQ|MISS|HIT|EA)" $GRAFT_REPO_ROOT/gpurun_out/s1/counters_all.txt | head -150 > $GRAFT_REPO_ROOT/gpurun_out/s1/counters_grep.txt
rocm-smi --showmeminfo vram > $GRAFT_REPO_ROOT/gpurun_out/s1/smi.txt 2>&1
cat /sys/module/amdgpu/parameters/vm_fragment_size /sys/module/amdgpu/parameters/vm_block_size /sys/module/amdgpu/parameters/noretry > $GRAFT_REPO_ROOT/gpurun_out/s1/amdgpu_params.txt 2>&1
tail -30 $GRAFT_REPO_ROOT/gpurun_out/s1/mbt.txt
