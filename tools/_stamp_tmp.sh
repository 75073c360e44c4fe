timeout 600 python -m pytest tests/test_encoder_hip.py tests/test_encoder_sizes_gpu.py -m gpu -x -q 2>&1 | tail -4
for i in 1 2; do python tools/bench_encoder.py --steps 360 --tag x6; done
export MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/libmtfjsp_stamp.so
MTFJSP_STAMP_PRINT=1 python tools/bench_encoder.py --steps 36 --tag stamp 2>&1 | grep -E "STAMP gin_gemm|stamp" | awk 'NR>=22 && NR<=27 || /^stamp/'
