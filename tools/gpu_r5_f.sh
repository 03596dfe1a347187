#!/bin/bash
# round 5, GPU call F: new tests; what the refill encoder passes cost a job (runs with the passes left out)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_end_to_end.py tests/test_gpu_transcribe.py -x -q -m gpu 2>&1 | tail -15 > gpurun_out/r5_f_tests.log
L=gpurun_out/r5_f_encoder_cost.jsonl; : > $L
timeout 400 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --polls 0 --groups 0,16 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --polls 0 --groups 0,16 2>&1 | grep '^{' >> $L
timeout 300 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype bfloat16 --mode refill --polls 0 --groups 0,16 2>&1 | grep '^{' >> $L
cat gpurun_out/r5_f_tests.log
python - <<'PY'
import json
for l in open("gpurun_out/r5_f_encoder_cost.jsonl"):
    d = json.loads(l); r = d["refill"]
    print(d["dtype"], d["slots"], "refill %.0f audio-s/s %.2f s" % (r["audio_s_per_s"], r["seconds"]), [(v["row_groups"], round(v["audio_s_per_s"]), round(v["seconds"], 2)) for v in d["variants"]])
PY
