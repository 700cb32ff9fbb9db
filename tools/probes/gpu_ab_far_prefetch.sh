cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
for r in 1 2 3; do for m in 10000000 40000000 160000000; do for pf in 0 1; do
echo "round $r AMC_FAR_PREFETCH=$pf: $(AMC_FAR_PREFETCH=$pf python3 tools/gpu_workload.py ladder $m | tail -1)"; done; done; done | tee gpurun_out/r03_far_prefetch16_ab.txt

AMC_FAR_PREFETCH=1 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "sweep_bit_exact or three_grid or beyond_one_grid" 2>&1 | tail -2
