# C4 kernels standalone (--debug overlap=0).  NOTE (round 1): the two PMC passes below over the multi-hundred-microsecond batched
# kernels did not finish within 10 minutes of box time -- every rocprofv3 call is now bounded by `timeout`; only the kernel
# trace has been used.
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-/root/repo}"
O=gpurun_out/prof_c4; mkdir -p $O
timeout 120 rocprofv3 --kernel-trace --stats -d $O/trace -o t --output-format csv -- python tools/bench_configs.py --configs c4 --debug overlap=0 > $O/trace.log 2>&1
python tools/rocprof_summary.py --trace $O/trace | head -12
timeout 120 rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_BUSY_CYCLES --kernel-trace -d $O/sq -o s --output-format csv -- python tools/bench_configs.py --configs c4 --debug overlap=0 > $O/sq.log 2>&1
timeout 120 rocprofv3 --pmc FETCH_SIZE WRITE_SIZE --kernel-trace -d $O/fw -o s --output-format csv -- python tools/bench_configs.py --configs c4 --debug overlap=0 > $O/fw.log 2>&1
python tools/rocprof_summary.py --pmc $O/sq --pmc $O/fw | grep "k_bin_wave<0, 4, true>\|k_fuse_list<4, 256, 1024, 0, true>"
