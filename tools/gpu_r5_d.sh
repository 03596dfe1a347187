#!/bin/bash
# round 5, GPU call D: large f32 engines with the decode step on the three-plane tiles: tests, then corpus A/B against the
# f32 instruction (options 128) at 1250 and 512 slots, row-group counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_decode_x6.py -x -q -m gpu -s 2>&1 | tail -25 > gpurun_out/r5_d_tests.log
L=gpurun_out/r5_d_x6.jsonl; : > $L
timeout 400 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode both --decode-probe --polls 0 --groups 1,2 2>&1 | grep '^{' >> $L
timeout 400 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe --options 128 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 512 --segments 5120 --dtype float32 --mode refill --decode-probe 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 512 --segments 5120 --dtype float32 --mode refill --decode-probe --options 128 2>&1 | grep '^{' >> $L
cat gpurun_out/r5_d_tests.log; wc -l $L
