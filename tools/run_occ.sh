#!/bin/bash
timeout -k 10 300 python tools/gpu_perf.py dragon random cornell scene perlin checker mesh:160x320 --spp 1024 --batch 1024 --chunk 0 --counters 0 2>&1 | grep Msamples | sed -E "s/chunk=.*kernel_s/kernel_s/; s/mean=\S+//"
timeout -k 10 100 python tools/gpu_perf.py dragon --threads 768 --spp 1024 --batch 1024 --chunk 0 --counters 0 2>&1 | grep Msamples | sed -E "s/chunk=.*kernel_s/kernel_s/; s/mean=\S+//"
