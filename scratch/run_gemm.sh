set -e
echo "== default"; timeout -k 10 200 python3 scratch/gemm_bench.py 2>&1 | grep -v amdgpu
for t in 0 1 2 3; do echo "== tile $t"; CRFCONV_GEMM_TILE=$t timeout -k 10 200 python3 scratch/gemm_bench.py 2>&1 | grep -v amdgpu; done
