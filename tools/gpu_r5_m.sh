#!/bin/bash
# round 5, GPU call M: the 64-row tiles only for CONCURRENT row groups -- its test, the tile-family tests, one corpus run
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 300 python -m pytest tests/test_gpu_parity_r5.py "tests/test_gpu_retire.py::test_retired_rows_change_no_id" -q -m gpu 2>&1 | tail -3
timeout 200 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode refill --decode-probe 2>&1 | grep '^{' > gpurun_out/r5_m_corpus.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_m_corpus.jsonl").read())
print("refill", round(d["refill"]["audio_s_per_s"]), d["refill"]["tokens_sha16"], {k: round(v["decode_ms"], 1) for k, v in d["full_length_decode"].items()})
PY
