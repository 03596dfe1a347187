#!/bin/bash
# round 5, GPU call A: the new tests, then A/B of the refill schedule (poll interval, row groups) and of sleeping vs spinning waits
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_transcribe.py tests/test_gpu_end_to_end.py tests/test_gpu_retire.py tests/test_gpu_f32_encoder.py tests/test_gpu_engine.py -x -q -m gpu 2>&1 | tail -40 > gpurun_out/r5_a_tests.log
L=gpurun_out/r5_a_ab.jsonl; : > $L
timeout 300 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe --polls 16,8 --groups 0 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe --polls 0 --groups 2 2>&1 | grep '^{' >> $L
timeout 200 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe --spin-waits 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --decode-probe --polls 0,16,8 --groups 0,1,2 2>&1 | grep '^{' >> $L
timeout 200 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --decode-probe --spin-waits 2>&1 | grep '^{' >> $L
cat gpurun_out/r5_a_tests.log; wc -l $L
