export MTFJSP_LIB=$PWD/e2e-mappo-for-mt-fjsp_amd/libmtfjsp_stamp.so
MTFJSP_STAMP_PRINT=1 python tools/bench_encoder.py --steps 36 --tag stamp 2>&1 | grep -E "STAMP k_heads|STAMP k_gat3|stamp" | head -12
