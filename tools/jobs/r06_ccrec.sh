#!/bin/bash
# GPU job (round 6): StackTwoCubes' cube<->cube records at 12 floats in the Newton kernels (66 KiB of LDS per wave instead of 74): parity, step times, config 5
mkdir -p gpurun_out
{
python -m pytest tests/test_gpu_parity.py tests/test_gpu_property.py -q -n 4 -p no:cacheprovider -k "stack or property or frames" 2>&1 | grep -E "^(FAILED|ERROR)|passed|failed" | cut -c1-200 | tail -12
for n in 65536 32768; do echo "$(python tools/quick_times.py stack --n $n --steps 100 2>&1 | grep -v amdgpu.ids | tail -1 | cut -c1-20,100-170)"; done
for ov in 0 1; do echo "LCR_RENDER_OVERLAP=$ov: $(LCR_RENDER_OVERLAP=$ov python bench.py --config 5 --no-cpu-baseline 2>/dev/null | tail -1 | head -c 230)"; done
} 2>&1 | tee gpurun_out/r06_ccrec.txt
