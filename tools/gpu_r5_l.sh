#!/bin/bash
# round 5, GPU call L: the 64-row f32 tiles for row blocks of >= 256 rows as the product rule -- the tests that hold the two
# tile families against each other, then the ragged corpus on the final sources
cd "${GRAFT_REPO_ROOT:-/root/repo}"
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_retire.py tests/test_gpu_parity_r5.py tests/test_gpu_kernels.py "tests/test_gpu_parity_r3.py::test_row_group_counts_follow_operand_type_and_batch" "tests/test_gpu_transcribe.py::test_refilled_slots_decode_every_segment_bit_identically" -q -m gpu 2>&1 | tail -4
timeout 300 python tools/eos_corpus.py --slots 1250 --segments 10000 --dtype float32 --mode both --check --decode-probe 2>&1 | grep '^{' > gpurun_out/r5_l_corpus.jsonl
python - <<'PY'
import json
d = json.loads(open("gpurun_out/r5_l_corpus.jsonl").read())
print({m: (round(d[m]["audio_s_per_s"]), round(d[m]["hbm_frac_on_live_bytes_whole_pass"], 3), d[m]["tokens_sha16"]) for m in ("batch", "refill")}, d["tokens_identical"],
      {k: round(v["decode_ms"], 1) for k, v in d["full_length_decode"].items()})
PY
