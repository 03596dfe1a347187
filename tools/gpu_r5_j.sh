#!/bin/bash
# round 5, GPU call J: 64-row fold tiles for large row groups (experiment bits 256: K slices of 256, 512: of 128)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
L=gpurun_out/r5_j_tall.jsonl; : > $L
for o in 512 768; do
  timeout 300 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe --options $o 2>&1 | grep '^{' >> $L
done
timeout 300 python tools/eos_corpus.py --slots 512 --segments 5120 --dtype float32 --mode refill --decode-probe --options 0 2>&1 | grep "^{" >> $L
timeout 300 python tools/eos_corpus.py --slots 512 --segments 5120 --dtype float32 --mode refill --decode-probe --options 768 2>&1 | grep "^{" >> $L
python - <<'PY'
import json
for l in open("gpurun_out/r5_j_tall.jsonl"):
    d = json.loads(l); r = d["refill"]; f = d["full_length_decode"]
    print("slots", d["slots"], "options %4d" % d["options"], "refill %.0f audio-s/s (%.3f s)" % (r["audio_s_per_s"], r["seconds"]), "tokens", r["tokens_sha16"],
          "full-length decode: groups %.1f ms, one stream %.1f ms" % (f["row_groups"]["decode_ms"], f["single_stream"]["decode_ms"]))
PY
