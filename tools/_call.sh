bash tools/gpurun.sh test smoke micro pmcdec
bash tools/gpurun.sh bench
