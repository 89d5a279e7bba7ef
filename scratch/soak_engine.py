"""soak of the engine-level and lexical fuzzers (tests/test_gpu_fuzz.py) with other seeds: python scratch/soak_engine.py <cases> <seed>"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import test_gpu_fuzz as F
cases, seed = int(sys.argv[1]), int(sys.argv[2])
print("engine fuzz ok: %d engines" % F.fuzz_engine(cases, seed), flush=True)
print("lexical fuzz ok: %d indexes" % F.fuzz_lexical(cases, seed + 1), flush=True)
