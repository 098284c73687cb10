#!/bin/bash
# A/B on one box of two builds of the library on the N 4096 attention kernels: bash tools/attn_lib_ab.sh /root/repo/adaprompt_amd/build/lib_prev.so
for rep in 1 2 3; do
    timeout -k 10 120 python3 tools/attn_quick_probe.py 2>/dev/null | grep "scale=0.0" | sed "s/^/this /"
    ADAP_LIB_PATH=$1 timeout -k 10 120 python3 tools/attn_quick_probe.py 2>/dev/null | grep "scale=0.0" | sed "s/^/that /"
done
