set -e
mkdir -p gpurun_out
timeout -k 10 500 python -m pytest tests/test_gpu_parity.py -m gpu -x -q > gpurun_out/narrow_pytest.log 2>&1 || { tail -30 gpurun_out/narrow_pytest.log; exit 1; }
tail -3 gpurun_out/narrow_pytest.log
for r in 1 2 3; do
  for w in 0 1; do
    for c in 3 5; do
      AMC_WIDE_COUNTERS=$w STEPS=4000 timeout -k 10 120 python tools/gpu_configs.py $c | sed "s/^/wide=$w /" | cut -c1-140 >> gpurun_out/narrow_ab.txt
    done
  done
done
cat gpurun_out/narrow_ab.txt
