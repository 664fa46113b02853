python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "glu" 2>&1 | tail -1
for i in 1 2; do echo prev; L2S_LIB_PATH=build_ab/glu_prev/liblip2speech_hip.so python tools/glu_bench.py 640 | grep glu; echo upfront; python tools/glu_bench.py 640 | grep glu; done
echo prev160; L2S_LIB_PATH=build_ab/glu_prev/liblip2speech_hip.so python tools/glu_bench.py 160 | grep glu; echo upfront160; python tools/glu_bench.py 160 | grep glu
python -m pytest tests/test_models_gpu.py -m gpu -x -q -k "conformer" 2>&1 | tail -1
