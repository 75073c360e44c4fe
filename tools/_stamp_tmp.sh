timeout 600 python -m pytest tests/test_encoder_hip.py tests/test_encoder_sizes_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python tools/bench_encoder.py --steps 360 --tag x6; MTFJSP_GEMM_DBG=16 python tools/bench_encoder.py --steps 360 --tag x6-prio; done
