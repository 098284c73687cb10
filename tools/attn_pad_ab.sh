#!/bin/bash
# A/B on one box of attention switches (N 4096, d 40, pre-scaled queries) through ADAP_ATTN_XCD: bit 16 / 32 switch the spare-chunk
# addends of the forward / the dK/dV kernel off, bit 8 forces the XCD-aware map on the N 4096 backward kernels.
# Interleaved repeats; each line is tools/attn_quick_probe.py's.   bash tools/attn_pad_ab.sh "7 15 23"
for rep in 1 2 3; do
    for x in ${1:-7 23 39 55}; do
        ADAP_ATTN_XCD=$x timeout -k 10 120 python3 tools/attn_quick_probe.py 2>/dev/null | grep "scale=0.0" | sed "s/^/xcd=$x /"
    done
done
