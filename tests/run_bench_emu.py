"""test helper: bench.py's main() with the host emulator injected as the device library
(tests/test_bench_line.py runs this in a subprocess; the product loader refuses `host-emu`)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "emu"))

import build_emu                      # noqa: E402
from pyro2_amd import _lib            # noqa: E402

_lib.use_library(build_emu.build(), allow_backends=("host-emu",))
import bench                          # noqa: E402

sys.argv = ["bench.py"] + sys.argv[1:]
bench.main()
