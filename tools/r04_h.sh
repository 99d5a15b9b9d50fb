#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/r4h
python - > gpurun_out/r4h/clock_probe.txt 2>&1 <<'PY'
import torch, time, subprocess
try:
    print('torch.cuda.clock_rate', torch.cuda.clock_rate())
except Exception as e:
    print('clock_rate failed', repr(e))
print(subprocess.run(['rocm-smi', '--showclocks'], capture_output=True, text=True).stdout[-1500:])
import glob
for f in sorted(glob.glob('/sys/class/drm/card*/device/pp_dpm_sclk')): print(f, open(f).read())
for f in sorted(glob.glob('/sys/class/drm/card*/device/hwmon/hwmon*/freq1_input')): print(f, open(f).read())
PY
cat gpurun_out/r4h/clock_probe.txt
timeout 900 python -m pytest tests/test_distributed_gpu.py -m gpu -x -q -k "unattended or one_rank" 2>&1 | tail -5
