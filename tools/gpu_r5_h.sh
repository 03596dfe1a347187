#!/bin/bash
# round 5, GPU call H: L2 weight prefetch A/B in the product (experiment bits 256 = GEGLU -> fold, 512 = attention ->
# out-projections, 1024 = out-projection -> GEGLU): refilled ragged pass at 256 slots + the canonical full-length decode
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 60 build/micro/l2_prefetch > gpurun_out/r5_l2_prefetch.txt 2>&1; cat gpurun_out/r5_l2_prefetch.txt
timeout 200 python tools/pf_dbg.py 2>&1 | tail -6
L=gpurun_out/r5_h_prefetch.jsonl; : > $L
for o in 0 256 512 1024 1792 0; do
  timeout 200 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype float32 --mode refill --decode-probe --options $o 2>&1 | grep '^{' >> $L
done
for o in 0 1792; do
  timeout 200 python tools/eos_corpus.py --slots 256 --segments 2560 --dtype bfloat16 --mode refill --decode-probe --options $o 2>&1 | grep '^{' >> $L
done
python - <<'PY'
import json
for l in open("gpurun_out/r5_h_prefetch.jsonl"):
    d = json.loads(l); r = d["refill"]; f = d["full_length_decode"]
    print(d["dtype"], "options %4d" % d["options"], "refill %.0f audio-s/s (%.3f s)" % (r["audio_s_per_s"], r["seconds"]),
          "full-length decode: groups %.1f ms, one stream %.1f ms" % (f["row_groups"]["decode_ms"], f["single_stream"]["decode_ms"]))
PY
