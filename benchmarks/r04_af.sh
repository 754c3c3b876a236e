mkdir -p gpurun_out
export TMPDIR=/tmp
python -c "import __graft_entry__ as g; g.build(); g.smoke(); print('smoke ok')" 2>&1 | tail -3
bash benchmarks/pmc_kernels.sh > gpurun_out/r04af_pmc_gemm_stalls.txt 2>&1; tail -30 gpurun_out/r04af_pmc_gemm_stalls.txt
