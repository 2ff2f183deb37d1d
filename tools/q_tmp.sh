python -m pytest tests/test_gpu_round5.py -q -k "bf16" 2>&1 | tail -3
for i in 1 2; do
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_both_$i.json 2>gpurun_out/bf_both_$i.err
done
