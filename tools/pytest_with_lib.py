"""Runs pytest with another build of the library in place of the in-tree one (mutation checks: a build
whose shadow-ray proof is deliberately too permissive must fail tests/test_shadow_proof_gpu.py).

usage: python tools/pytest_with_lib.py path/to/librm_other.so <pytest arguments>
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from raymarching_amd import _lib

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
import pytest

sys.exit(pytest.main(sys.argv[2:]))
