timeout 600 python -m pytest tests/test_encoder_hip.py tests/test_encoder_sizes_gpu.py tests/test_eval_per_instance_bn.py -m gpu -x -q 2>&1 | tail -12
for i in 1 2; do python tools/bench_encoder.py --steps 360 --tag gin0x; MTFJSP_GIN0_VALU=1 python tools/bench_encoder.py --steps 360 --tag gin0-valu; done
python tools/bench_encoder.py --size 10x10x2 --batch 8192 --steps 200 --tag gin0x-10; MTFJSP_GIN0_VALU=1 python tools/bench_encoder.py --size 10x10x2 --batch 8192 --steps 200 --tag gin0-valu-10
