timeout 600 python -m pytest tests/test_encoder_hip.py tests/test_encoder_sizes_gpu.py tests/test_eval_per_instance_bn.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python tools/bench_encoder.py --steps 360 --tag headsx; done
