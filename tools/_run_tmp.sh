timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
python bench.py --no-cpu-baseline --no-env-sweep > gpurun_out/b_6x6.json 2>/dev/null
python bench.py --no-cpu-baseline --no-env-sweep --size 10x10x2 --batch 8192 --steps 400 --warmup 200 > gpurun_out/b_10x10.json 2>/dev/null
python bench.py --no-cpu-baseline --no-env-sweep --size 20x20x4 --batch 2048 --steps 800 --warmup 400 > gpurun_out/b_20x20.json 2>/dev/null
python bench.py --no-cpu-baseline --no-env-sweep --trajectory full > gpurun_out/b_6x6_full.json 2>/dev/null
