bash tools/gpurun.sh test smoke bench
