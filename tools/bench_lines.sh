#!/bin/bash
# The round's bench lines (runs ON THE GPU BOX): one JSON line per workload into gpurun_out/$1/lines.jsonl
OUT=gpurun_out/$1; mkdir -p $OUT; : > $OUT/lines.jsonl
run() { name=$1; shift; echo "== $name: bench.py $@"; python3 bench.py "$@" 2> $OUT/$name.err | tee -a $OUT/lines.jsonl | cut -c1-200; }
run headline
run c2_fwd_256 --vol 256 --img 256 --grads none --steps 20 --no-cpu-baseline --pmc off
run c3_tf_only --grads tf --no-cpu-baseline --pmc off
run c3_tf_only_bricks --grads tf --no-tape --no-cpu-baseline --pmc off
run c3_tf1 --grads tf --tf tf1 --no-cpu-baseline --pmc off
run tf1 --tf tf1 --no-cpu-baseline --pmc off
run c5_view_1024_f16 --vol 1024 --img 1024 --vol-dtype f16 --jitter --steps 3 --warmup 1 --no-cpu-baseline --pmc off
run ct_scene_tf1 --tf tf1 --scene ct --no-cpu-baseline --pmc off
run ct_scene_tf1_grads_vol --tf tf1 --scene ct --grads vol --no-cpu-baseline --pmc off
run views8_256 --vol 256 --img 256 --views 8 --steps 10 --no-cpu-baseline --pmc off
run inside_camera --cam inside --steps 5 --no-cpu-baseline --pmc off
run opt_demo --workload opt --steps 10 --warmup 3
run opt_demo_steady --workload opt --steps 10 --warmup 12
run opt_demo_ct --workload opt --scene ct --steps 10 --warmup 3
# multi-GPU control flow on the one card of this box (rehearsals, not scaling numbers): RCCL with ONE rank, gloo with four
echo "== rccl_one_rank_rehearsal"; DR_BENCH_FORCE_DIST=1 DR_ALLREDUCE_SINGLE_RANK=1 python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --pmc off 2> $OUT/rccl1.err | tee $OUT/rccl_one_rank.json | cut -c1-200
run gpus4_gloo_rehearsal --gpus 4 --steps 3 --warmup 1 --no-cpu-baseline --pmc off
run gpus4_rows_gloo_rehearsal --gpus 4 --split rows --steps 3 --warmup 1 --no-cpu-baseline --pmc off
