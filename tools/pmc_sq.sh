#!/bin/bash
# SQ counters of the enhancement kernels (issue / wait breakdown); two passes of 8 SQ slots
out=$PWD/gpurun_out/pmc_sq
mkdir -p $out
export TMPDIR=/tmp
A="--frames 64 --steps 1 --warmup 0 --lanes 1 --cpu-frames 0 --no-profile-pass"
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_VALU --output-format csv -d $out/p1 -o r -- python3 bench.py $A > /dev/null 2> $out/p1.err || exit 1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_SALU SQ_BUSY_CYCLES SQ_WAVES SQ_ACTIVE_INST_SCA SQ_INSTS_VMEM_RD --output-format csv -d $out/p2 -o r -- python3 bench.py $A > /dev/null 2> $out/p2.err || exit 2
find $out/p1 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/p1.csv
find $out/p2 -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/p2.csv
rm -rf $out/p1 $out/p2
ls -la $out
