#!/bin/bash
# Cross-compiles the micro-benchmarks in the CPU container; the binaries travel to the GPU box under build/micro/
# (git-ignored, not gpurun-ignored).  usage: build_glds_probe.sh ["<probe>:<ns>:<bk> ..."]
cd "$(dirname "$0")/../.."
mkdir -p build/micro
rm -f build/micro/glds_probe_*
for v in ${1:-0:3:32 0:4:32 1:3:32 2:3:32 4:3:32 6:3:32 0:2:64 4:2:64}; do
  IFS=: read -r p ns bk <<< "$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -DMT3_GLDS_PROBE=$p -DMT3_GLDS_NS=$ns -DMT3_GLDS_BK=$bk ${EXTRA_DEFS} \
    -I include -I mt3_amd/csrc tools/micro/glds_probe.hip mt3_amd/csrc/errors.cpp -o build/micro/glds_probe_p${p}_ns${ns}_bk${bk} &
done
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 tools/micro/xcd_barrier.hip -o build/micro/xcd_barrier &
for t in frontend_probe attn_probe; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -I include tools/micro/$t.cpp -L mt3_amd -lmt3hip \
    -Wl,-rpath,'$ORIGIN/../../mt3_amd' -o build/micro/$t &
done
wait
ls build/micro/
# the MXFP8 encoder GEMM: complete / parts left out, ring depth 2 and 3
rm -f build/micro/mx8_probe_*
for v in ${MX8:-0:2 0:3 1:2 2:2 4:2 6:2}; do
  IFS=: read -r p ns <<< "$v"
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -Wno-unused-value -DMT3_MX8_PROBE=$p -DMT3_MX8_NS=$ns -I include -I mt3_amd/csrc \
    tools/micro/mx8_probe.hip mt3_amd/csrc/errors.cpp mt3_amd/csrc/mx8_host.cpp -o build/micro/mx8_probe_p${p}_ns${ns} &
done
wait
ls build/micro/
