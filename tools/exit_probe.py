#!/usr/bin/env python3
"""Diagnostic for a crash inside the process's exit handlers (seen under `rocprofv3 --pmc` after cooperative
launches, round 1): runs bench.main() with the given arguments, then saves /proc/self/maps so that the addresses
of the fault handler's stack trace can be attributed to a library.  Usage (on the GPU box):
    rocprofv3 --pmc SQ_WAVES -d gpurun_out/x -- python3 tools/exit_probe.py gpurun_out/maps.txt --steps 1 ..."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = sys.argv[1]
sys.argv = ["bench.py"] + sys.argv[2:]
import bench  # noqa: E402

bench.main()
with open("/proc/self/maps") as f, open(out, "w") as g:
    g.write(f.read())
print("exit_probe: maps saved, leaving main", flush=True)
