#!/usr/bin/env python3
"""Dev tool: one case of tests/test_gpu_dis_bisect.py under the resident launch and the launch chain (eps, ESS to 17 digits)."""
import os
import sys
sys.path.insert(0, '.')
sys.path.insert(0, 'tests')
import viabel_amd as vb
from viabel_amd import _lib
_lib.default_engine()
import test_gpu_dis_bisect as T
case = T.CASES[int(sys.argv[1])]
for env in ({}, {'VB_DIS_TRACE': '1'}, {'VB_DIS_RESIDENT': '0'}, {'VB_DIS_RESIDENT': '0', 'VB_DIS_TRACE': '1'}):
    r = T.run(vb, *case, env)
    print(env, repr(r[0]), repr(r[1]))
