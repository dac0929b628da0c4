cd /root/repo
timeout 1500 python -m pytest tests -m gpu -x -q > gpurun_out/pt.log 2>&1; echo rc=$? >> gpurun_out/pt.log
