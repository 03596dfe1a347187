#!/usr/bin/env python3
"""Round-6 A/B builds of libmt3hip.so: `python tools/ab_r6.py build` compiles gemm.hip once per MT3_EXP value (the other
objects come from the product build) into build/exp/libmt3hip_exp<N>.so; `python tools/ab_r6.py run [N ...]` / `phases` (on the GPU
box) puts each in the product's place in turn, runs the f32 headline (`bench.py --no-cpu-baseline --no-extras`) and prints
value / ms per step, then restores the product library.  Results: profiles/r6_ab_decode_gemm_variants.txt."""
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from mt3_amd import build as B  # noqa: E402

EXP = os.path.join(ROOT, "build", "exp")
# MT3_EXP value -> (what, sources compiled with -DMT3_EXP=value)
VARIANTS = {0: ("product (decode-sized GEMM tiles at s_setprio 3, kernel arguments behind one scalar round trip)", []),
            64: ("product without the kernel-argument pin (GEMM tiles and decode attention)", ["gemm.hip", "attention.hip"]),
            66: ("neither the priority nor the pin (the kernels of round 5)", ["gemm.hip", "attention.hip"]),
            2: ("decode-sized GEMM tiles at the default priority (the product of rounds 1-5)", ["gemm.hip"]),
            32: ("product + in-situ phase accounting of the decode-sized tiles (tools/gemm_phases_in_situ.py)", ["gemm.hip"])}


def build():
    B.build_library()
    os.makedirs(EXP, exist_ok=True)
    inc = ["-I", os.path.join(ROOT, "include"), "-I", B.CSRC]
    objs = [os.path.join(B.OBJ, f + ".o") for f in B.HIP_SOURCES + B.CPP_SOURCES]
    for n, (what, sources) in VARIANTS.items():
        if n == 0:
            continue
        link = list(objs)
        for src in sources:
            obj = os.path.join(EXP, "%s_exp%d.o" % (src, n))
            subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", *inc, "-DMT3_EXP=%d" % n, "-c",
                            os.path.join(B.CSRC, src), "-o", obj], check=True)
            link = [obj if o.endswith(src + ".o") else o for o in link]
        subprocess.run([B._hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", *link, "-o",
                        os.path.join(EXP, "libmt3hip_exp%d.so" % n)], check=True)
        print("built variant", n, what, flush=True)


def run(which, bench_args, phases=0, ids=False):
    keep = B.LIB + ".product"
    shutil.copy2(B.LIB, keep)
    try:
        if ids:
            for n in which:
                shutil.copy2(keep if n == 0 else os.path.join(EXP, "libmt3hip_exp%d.so" % n), B.LIB)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "ids_hash.py")], capture_output=True, text=True, cwd=ROOT)
                print("variant %d (%s): %s" % (n, VARIANTS.get(n, ("?",))[0], (r.stdout.strip().splitlines() or [r.stderr[-400:]])[-1]), flush=True)
            return
        if phases:
            shutil.copy2(os.path.join(EXP, "libmt3hip_exp%d.so" % phases), B.LIB)
            subprocess.run([sys.executable, os.path.join(ROOT, "tools", "gemm_phases_in_situ.py")], cwd=ROOT)
            return
        for n in which:
            shutil.copy2(keep if n == 0 else os.path.join(EXP, "libmt3hip_exp%d.so" % n), B.LIB)
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--no-cpu-baseline", "--no-extras", *bench_args],
                               capture_output=True, text=True, cwd=ROOT)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("variant %d (%s): FAILED rc %d\n%s" % (n, VARIANTS.get(n, ("?",))[0], r.returncode, (r.stdout + r.stderr)[-1500:]), flush=True)
                continue
            d = json.loads(line[-1])
            print("variant %d (%s): %.1f %s, %.1f ms per step, decode %.1f ms" % (
                n, VARIANTS.get(n, ("?",))[0], d["value"], d["unit"], d["ms_per_step"],
                (d.get("roofline") or {}).get("decode_ms_product_schedule", float("nan"))), flush=True)
    finally:
        shutil.copy2(keep, B.LIB)
        os.remove(keep)


if __name__ == "__main__":
    if sys.argv[1] == "build":
        build()
    elif sys.argv[1] == "ids":
        run([int(a) for a in sys.argv[2:]] or [0, 2, 64, 66], [], ids=True)
    elif sys.argv[1] == "eos":
        keep = B.LIB + ".product"
        shutil.copy2(B.LIB, keep)
        try:
            for n in [int(a) for a in sys.argv[2:]]:
                shutil.copy2(keep if n == 0 else os.path.join(EXP, "libmt3hip_exp%d.so" % n), B.LIB)
                r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "eos_profile.py"), "float32", "4"], capture_output=True, text=True, cwd=ROOT)
                print("variant %d (%s):\n%s" % (n, VARIANTS.get(n, ("?",))[0], "\n".join(r.stdout.strip().splitlines()[-3:]) or r.stderr[-400:]), flush=True)
        finally:
            shutil.copy2(keep, B.LIB)
            os.remove(keep)
    elif sys.argv[1] == "phases":
        run([], [], phases=int(sys.argv[2]) if len(sys.argv) > 2 else 32)
    else:
        args = sys.argv[2:]
        sep = args.index("--") if "--" in args else len(args)
        run([int(a) for a in args[:sep]] or sorted(VARIANTS), args[sep + 1:] or ["--steps", "6", "--warmup", "2"])
