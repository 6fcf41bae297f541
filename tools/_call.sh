set -o pipefail
mkdir -p gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -s > gpurun_out/r02_gpu_tests_a.log 2>&1
echo "pytest rc=$?"
grep -a "net-error\|transcript\|passed\|failed\|Error\|error" gpurun_out/r02_gpu_tests_a.log | tail -60
