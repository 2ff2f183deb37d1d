# in-step A/B of the bf16 node's un-stored first / pooled last layers (fused_mlp.BF16_FIRST / BF16_POOL), alternating
for i in 1 2; do
PN2_BF16_POOL=0 PN2_BF16_FIRST=0 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_stored_$i.json 2>gpurun_out/bf_stored_$i.err
PN2_BF16_FIRST=0 python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_pool_$i.json 2>gpurun_out/bf_pool_$i.err
python bench.py --dtype bf16 --no-cpu-baseline > gpurun_out/bf_both_$i.json 2>gpurun_out/bf_both_$i.err
done
