mkdir -p gpurun_out/tw
for m in 0 1 3 5; do
  echo "== GPEMSR_WINO_TRAIN=$m" >> gpurun_out/tw/sweep.log
  GPEMSR_WINO_TRAIN=$m timeout -k 10 300 python -m pytest "tests/test_train_gpu.py::test_gradient_distance_to_fp64_against_the_references_own" -q -m gpu -s 2>&1 | grep -E "median|passed|failed|SR distance|loss vs" >> gpurun_out/tw/sweep.log
  GPEMSR_WINO_TRAIN=$m python bench.py --mode train --steps 5 --warmup 2 2>/dev/null | python -c "import sys,json; j=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(j['value'], j['ms_per_step'])" >> gpurun_out/tw/sweep.log
done
cat gpurun_out/tw/sweep.log
