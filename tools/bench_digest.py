"""stdin: bench.py's JSON line -> the handful of figures worth reading in a gpurun tail."""
import json
import sys

line = sys.stdin.read().strip().splitlines()
try:
    d = json.loads(line[-1])
except Exception as e:
    print("no JSON line:", e, line[-3:] if line else "")
    sys.exit(0)
r = d.get("roofline") or {}
print("HEADLINE %s %.1f %s  ms/step %.1f  n_gpus %d steps %d | attn %.2f us frac %.3f traffic %s | cross %.2f us | small %.1f us/step "
      "| whole-step hbm %.3f (product schedule: %.1f ms, %.3f)" % (
          d["dtype"], d["value"], d["unit"], d["ms_per_step"], d["n_gpus"], d["steps"],
          r.get("avg_launch_us", 0), r.get("frac", 0), r.get("traffic"),
          (r.get("cross_attn") or {}).get("avg_launch_us", 0), r.get("small_kernel_us_per_step", 0),
          r.get("whole_step_hbm_frac", 0), r.get("decode_ms_product_schedule", 0),
          r.get("whole_step_hbm_frac_product_schedule", 0)))
print("  schedule: %s | graph replay %s: %.1f ms, host cpu %.2f s | direct launches: %.1f ms, host cpu %.2f s" % (
    (d.get("config") or {}).get("decode_schedule"), r.get("product_schedule_graph_replay"),
    r.get("decode_ms_product_schedule", 0), r.get("host_cpu_s_per_decode", 0), r.get("decode_ms_direct_launches", 0),
    r.get("host_cpu_s_per_decode_direct_launches", 0)))
for k, v in (d.get("extra") or {}).items():
    if not isinstance(v, dict):
        continue
    if k == "eos_schedule":
        for kk, vv in v.items():
            print("  EOS_SCHEDULE %s:" % kk, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in vv.items()
                                             if a != "workload"})
        continue
    if k == "divergence_vs_f32":
        for kk, vv in v.items():
            if isinstance(vv, dict):
                print("  DIVERGENCE %s:" % kk, {a: (round(b, 4) if isinstance(b, float) else b) for a, b in vv.items()
                                                 if not isinstance(b, dict)})
                for lab in ("trained", "boosted"):
                    w = vv.get(lab)
                    if isinstance(w, dict):
                        print("      %s: " % lab + ", ".join("%s=%s" % (a, round(b, 4) if isinstance(b, float) else b)
                                                             for a, b in w.items() if not isinstance(b, dict)) +
                              ("  | vs truth onset F1 %.4f" % w["vs_truth"]["onset_f1_note_number"] if "vs_truth" in w else ""))
        continue
    rr = v.get("roofline") or {}
    s = "  %s: " % k + ", ".join("%s=%s" % (a, round(b, 4) if isinstance(b, float) else b) for a, b in v.items()
                                 if a in ("value", "ms_per_step", "steps", "ms", "achieved", "frac", "encoder_ms",
                                          "segments_per_s", "traffic", "error"))
    if rr:
        s += " | attn %.2f us frac %.3f traffic %s small %.1f us/step whole-step %.3f (product %.3f; graph %.1f ms cpu %.2f s, direct %.1f ms cpu %.2f s)" % (
            rr["avg_launch_us"], rr["frac"], rr.get("traffic"), rr.get("small_kernel_us_per_step", 0),
            rr["whole_step_hbm_frac"], rr.get("whole_step_hbm_frac_product_schedule", 0),
            rr.get("decode_ms_product_schedule", 0), rr.get("host_cpu_s_per_decode", 0),
            rr.get("decode_ms_direct_launches", 0), rr.get("host_cpu_s_per_decode_direct_launches", 0))
    print(s[:600])
c = d.get("cpu_baseline")
if c:
    print("  CPU:", {k: v for k, v in c.items() if k in ("value", "cores", "nproc")}, c.get("sample", "")[-60:],
          {k: v.get("value") for k, v in c.items() if isinstance(v, dict)})
