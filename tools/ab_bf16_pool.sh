python -m pytest tests/test_gpu_round5.py tests/test_gpu_bf16.py -q 2>&1 | tail -5
for i in 1 2; do
PN2_BF16_POOL=0 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_stored_$i.json 2>gpurun_out/bf_stored_$i.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_pool_$i.json 2>gpurun_out/bf_pool_$i.err
done
