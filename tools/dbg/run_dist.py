import sys, runpy, json
sys.path.insert(0, '.')
from gem_amd import ElevationMap
knobs = {k: int(v) for k, v in (kv.split('=') for kv in sys.argv[1].split(','))} if sys.argv[1] != 'none' else {}
ElevationMap.default_debug = knobs
sys.argv = ['bench.py', '--gpus', '1', '--steps', '20', '--warmup', '5', '--no-cpu-baseline']
runpy.run_path('bench.py', run_name='__main__')
