# the round's standard full call: `gpurun --timeout 3300 -- bash tools/_call.sh`
bash tools/gpurun.sh test smoke bench prof
