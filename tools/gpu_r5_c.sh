#!/bin/bash
# round 5, GPU call C: the pipelined poll -- tests, then poll-interval A/B (f32 256 / 1250 slots, bf16 256 slots)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_transcribe.py tests/test_gpu_end_to_end.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_c_tests.log
L=gpurun_out/r5_c_polls.jsonl; : > $L
timeout 300 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --polls 2,8,16 --groups 0 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype bfloat16 --mode refill --polls 2,8,16 --groups 0 2>&1 | grep '^{' >> $L
timeout 400 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --polls 2,8 --groups 0 2>&1 | grep '^{' >> $L
cat gpurun_out/r5_c_tests.log; wc -l $L
