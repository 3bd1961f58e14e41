#!/bin/bash
# One-call evidence collection for a round (run through gpurun): kernel stats + PMC passes in the three arithmetic modes,
# the per-shape conv table, the default bench line with the full loop, every BASELINE config on one GPU, and the
# config-3 kernel stats. Outputs under gpurun_out/; summarise with tools/pmc_summary.py and copy into profiles/.
set -e
bash tools/profile_round.sh f16f8 > gpurun_out/pr_f16f8.log 2>&1
bash tools/profile_round.sh f16x3 > gpurun_out/pr_f16x3.log 2>&1
bash tools/profile_round.sh f32 > gpurun_out/pr_f32.log 2>&1
python tools/step_profile.py --batch 64 --steps 20 --precision f16f8 --csv gpurun_out/conv_shapes_f16f8.csv > gpurun_out/sp64.txt 2>&1
python tools/step_profile.py --batch 64 --steps 20 --precision f16x3 --csv gpurun_out/conv_shapes_f16x3.csv >> gpurun_out/sp64.txt 2>&1
python tools/step_profile.py --batch 64 --steps 10 --precision f32 --csv gpurun_out/conv_shapes_f32.csv >> gpurun_out/sp64.txt 2>&1
python bench.py > gpurun_out/bench_full.log 2>&1
: > gpurun_out/configs.txt
run_cfg() { echo "== bench.py $*" >> gpurun_out/configs.txt; python bench.py "$@" --no-cpu-baseline 2>/dev/null | tail -n 1 >> gpurun_out/configs.txt; }
run_cfg --batch 4 --res 16 --lres 8 --T 100 --steps 100
run_cfg --image-size 128
run_cfg --lres 8
run_cfg --batch 32 --lres 32 --T 100 --steps 50
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
rm -rf gpurun_out/prof_cfg3
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_cfg3 -- python3 bench.py --image-size 128 --steps 5 --warmup 1 --no-alt --no-cpu-baseline --no-full-loop > gpurun_out/cfg3.log 2>&1
# the "other" kernels of a step (copies / fills): with and without hipGraph replay
rm -rf gpurun_out/prof_nograph
SR3_NO_GRAPH=1 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_nograph -- python3 bench.py --steps 5 --warmup 1 --no-alt --no-cpu-baseline --no-full-loop > gpurun_out/nograph.log 2>&1
(python tools/step_profile.py --batch 1 --steps 30; python tools/step_profile.py --batch 4 --steps 30; python tools/step_profile.py --batch 4 --res 16 --lres 8 --T 100 --steps 30) > gpurun_out/small_batch.txt 2>&1
tail -n 1 gpurun_out/bench_full.log | cut -c1-300
