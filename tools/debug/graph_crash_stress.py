"""Stress for the box-dependent crash inside hipGraphLaunch: build a small model, run the golden case's few engine steps
(eager, capture, replays), drop it, repeat.  Prints the cycle count reached; a segmentation fault ends the process.
usage: python tools/debug/graph_crash_stress.py [cycles] [case]"""
import faulthandler
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
faulthandler.enable()
import mirror_utils as MU  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100
case = sys.argv[2] if len(sys.argv) > 2 else "two_mod_odd"
for i in range(n):
    MU.replay_training(case, "cuda", use_engine=True)
    if i % 20 == 19:
        print("cycles", i + 1, flush=True)
print("done", n, flush=True)
