set -e
mkdir -p gpurun_out
./benchmarks/micro/glds_ring.bin
python -m pytest tests/test_gpu_parity.py -x -q -k ssim > gpurun_out/r3_ssim_t.log 2>&1 || { tail -30 gpurun_out/r3_ssim_t.log; exit 1; }
tail -2 gpurun_out/r3_ssim_t.log
python benchmarks/bench_kernels.py --cases ssim > gpurun_out/r3_ssim_k.jsonl 2>&1; cat gpurun_out/r3_ssim_k.jsonl
python bench.py --ssim-threshold 0.95 --no-cpu-baseline --no-h2d --no-decode > gpurun_out/r3_bench_ssim.json 2>gpurun_out/r3_bench_ssim.err
python -c "
import json; d=json.load(open('gpurun_out/r3_bench_ssim.json')); print(d['kernel_ms'], d['ssim']['roofline'], d['ssim']['kept'], d['clusters'], d['ms_per_step'])"
