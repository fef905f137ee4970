#!/bin/bash
cd /root/repo
NSR_VARIANT_DATA=build/step_inputs.pt python tools/bin_variants.py build/variants/libnsr_hip_b256.so build/variants/libnsr_hip_b512.so build/variants/libnsr_hip_b1024.so build/variants/libnsr_hip_b512s.so build/variants/libnsr_hip_b1024s.so > gpurun_out/bin_variants.jsonl 2>&1
cat gpurun_out/bin_variants.jsonl
