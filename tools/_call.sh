bash tools/gpurun.sh pmcattn
cp gpurun_out/pmc/r6_pmc_summary.json profiles/r6_pmc_summary.json 2>/dev/null
bash tools/gpurun.sh test smoke bench prof pmcenc
