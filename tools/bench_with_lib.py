#!/usr/bin/env python3
"""Run bench.py against another build of the library: python tools/bench_with_lib.py <libsequitr_hip.so> [bench args]"""
import os
import runpy
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: F401  (torch's HIP runtime must load first)

from sequitr_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
sys.argv = [os.path.join(ROOT, "bench.py")] + sys.argv[2:]
runpy.run_path(sys.argv[0], run_name="__main__")
