mkdir -p gpurun_out/r4an
bash tools/pmc_attn.sh > gpurun_out/r4an/pmc_default.txt 2>&1
bash tools/pmc_attn.sh PM_ATTN_VARIANTS=0:47,1:4 > gpurun_out/r4an/pmc_variants.txt 2>&1
tail -60 gpurun_out/r4an/pmc_variants.txt | cut -c1-120
