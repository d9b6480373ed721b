#!/usr/bin/env python3
"""Body of tests/test_gpu_rccl.py: a fresh process that runs rise_sdf_amd.dist.rccl_selftest (a one-rank ``nccl`` = RCCL group on
cuda:0 driving GradBuckets over the real gradient tensors of the yaml's model).  Prints ``RESULT {json}``."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

if __name__ == "__main__":
    from rise_sdf_amd.dist import rccl_selftest
    print("RESULT " + json.dumps(rccl_selftest()), flush=True)
