import sys
sys.path.insert(0,'/root/repo/tools'); sys.path.insert(0,'/root/repo')
import fuzz_parity as fp
for seed in [int(a) for a in sys.argv[1:]]:
    print('seed', seed, fp.scenario(seed), flush=True)
