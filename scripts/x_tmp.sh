mkdir -p gpurun_out/r04
python -m pytest tests/test_gpu_cli.py -x -q -k "text_on_device" > gpurun_out/r04/cli_default.txt 2>&1; tail -n 12 gpurun_out/r04/cli_default.txt
GTX_TEXT_ON_DEVICE=1 python -m pytest tests/test_gpu_cli.py tests/test_gpu_group.py -x -q > gpurun_out/r04/cli_text_forced.txt 2>&1; tail -n 12 gpurun_out/r04/cli_text_forced.txt
