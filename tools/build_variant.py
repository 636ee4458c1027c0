#!/usr/bin/env python3
"""Build a macro variant of libtt.so for tools/ab_run.py / the tests' comparison library:
python tools/build_variant.py NAME [-DX=Y ...]  ->  ab/libtt_NAME.so   (twotowermlretrieval_amd.build.build_variant;
`python tools/build_variant.py ab -DTT_AB` is the comparison build the tests load)."""
import sys
from pathlib import Path
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from twotowermlretrieval_amd import build as B

print(B.build_variant(sys.argv[1], sys.argv[2:], force=True))
