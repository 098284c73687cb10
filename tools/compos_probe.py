#!/usr/bin/env python3
"""The config-4 leg of bench.py alone (Stage-2 compositional micro-batches), for `rocprofv3 --kernel-trace --stats`."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import bench
from adaprompt_amd import _lib

dev = torch.device("cuda:0")
torch.cuda.set_device(dev)
_lib.load()
gen = torch.Generator(device=dev).manual_seed(99)
print(json.dumps(bench.compos_leg(dev, gen, micro_batches=int(os.environ.get("MB", "8")))))
