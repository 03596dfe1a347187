#!/bin/bash
# Extra bench lines of round 3 (each its own `python bench.py` process, un-profiled): the driver's own command, the
# t5x beam-1 selection rule, BASELINE configs[4] as its own line, the f32 engine as the headline, batch scaling.
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out/lines
run() { name=$1; shift; timeout 900 python bench.py "$@" > gpurun_out/lines/$name.log 2>&1; echo "exit $? : $name ($*)"; grep -h '^{"metric"' gpurun_out/lines/$name.log | tail -1 > gpurun_out/lines/$name.json; python tools/bench_digest.py < gpurun_out/lines/$name.json | head -1; }
run driver_like --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline
run beam1 --decoding beam1 --steps 3 --warmup 1 --no-cpu-baseline --no-extras
run configs4 --model base --kv-dtype fp8_e4m3 --dense-dtype fp8_e4m3 --steps 3 --warmup 1 --no-cpu-baseline --no-extras
run f32_headline --dtype float32 --steps 3 --warmup 1 --no-cpu-baseline --no-extras
for b in 128 512 1024; do run batch_$b --batch $b --steps 2 --warmup 1 --no-cpu-baseline --no-extras; done
