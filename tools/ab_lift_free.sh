# in-step A/B of fused_mlp.LIFT_FREE (stored y0 of the lifted layers vs re-formed): python bench.py, alternating
for i in 1 2 3; do
PN2_LIFT_FREE=0 python bench.py > gpurun_out/ab_stored_$i.json 2>gpurun_out/ab_stored_$i.err
PN2_LIFT_FREE_MIN_ROWS=524288 python bench.py > gpurun_out/ab_free_$i.json 2>gpurun_out/ab_free_$i.err
done
