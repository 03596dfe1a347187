#!/bin/bash
# round 5, GPU call I: soak of the streaming entry; data points: configs[4] shape refilled, bf16 corpus on the final sources, slot counts
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 400 python tools/soak_transcribe.py --seconds 150 --dtype float32 > gpurun_out/r5_soak_f32.log 2>&1; echo "exit $? soak f32"; tail -3 gpurun_out/r5_soak_f32.log | cut -c1-300
timeout 300 python tools/soak_transcribe.py --seconds 90 --dtype bfloat16 --slots 130 > gpurun_out/r5_soak_bf16.log 2>&1; echo "exit $? soak bf16"; tail -2 gpurun_out/r5_soak_bf16.log | cut -c1-300
L=gpurun_out/r5_i_points.jsonl; : > $L
timeout 400 python tools/eos_corpus.py --model base --dtype bfloat16 --kv-dtype fp8_e4m3 --dense-dtype fp8_e4m3 --slots 256 --segments 2560 --mode both --check 2>&1 | grep '^{' >> $L
timeout 400 python tools/eos_corpus.py --dtype bfloat16 --slots 1250 --segments 10000 --mode both --check 2>&1 | grep '^{' >> $L
timeout 400 python tools/eos_corpus.py --dtype float32 --slots 1024 --segments 8192 --mode refill 2>&1 | grep '^{' >> $L
timeout 400 python tools/eos_corpus.py --dtype float32 --slots 128 --segments 1280 --mode both 2>&1 | grep '^{' >> $L
python - <<'PY'
import json
for l in open("gpurun_out/r5_i_points.jsonl"):
    d = json.loads(l)
    print(d.get("model"), d["dtype"], d["slots"], {m: (round(d[m]["audio_s_per_s"]), round(d[m]["hbm_frac_on_live_bytes_whole_pass"], 3)) for m in ("batch", "refill") if m in d}, d.get("tokens_identical"))
PY
