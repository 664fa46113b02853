python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "glu" 2>&1 | tail -2
L2S_GLU_CT=64 python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "glu" 2>&1 | tail -1
for i in 1 2; do echo ct64; L2S_GLU_CT=64 python tools/glu_bench.py 640 | grep glu; echo ct128; python tools/glu_bench.py 640 | grep glu; done
python -m pytest tests/test_models_gpu.py -m gpu -x -q -k "conformer" 2>&1 | tail -1
