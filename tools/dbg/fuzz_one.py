"""Replay scenarios of tools/fuzz_parity.py by seed, optionally with extra debug knobs forced on every map (a bisect aid).
usage: python tools/dbg/fuzz_one.py [key=value,...|none] seed [seed ...]"""
import sys
sys.path.insert(0, 'tools'); sys.path.insert(0, '.')
import fuzz_parity as fp
from gem_amd import ElevationMap

args = sys.argv[1:]
extra = {}
if args and ('=' in args[0] or args[0] == 'none'):
    if args[0] != 'none':
        extra = {k: int(v) for k, v in (kv.split('=') for kv in args[0].split(','))}
    args = args[1:]
ElevationMap.base_debug = dict(getattr(ElevationMap, "base_debug", {}) or {}, **extra)
for seed in [int(a) for a in args]:
    try:
        print('seed', seed, extra, 'ok', fp.scenario(seed)[1:], flush=True)
    except AssertionError as e:
        print('seed', seed, extra, 'MISMATCH', str(e)[:400], flush=True)
