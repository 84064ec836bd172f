#!/bin/bash
# Round profile: kernel durations (rocprofv3 --kernel-trace --stats) and HBM traffic (separate --pmc passes).
# Writes under gpurun_out/prof_<tag>; copy what should be judged into profiles/.
tag=${1:-r02}
out=$PWD/gpurun_out/prof_$tag
mkdir -p $out
export TMPDIR=/tmp
python bench.py > $out/bench.json 2> $out/bench.err || exit 1
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o run -- python3 bench.py --steps 3 --warmup 1 --cpu-frames 0 > $out/bench_under_rocprof.json 2> $out/rocprof_stats.err || exit 2
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_timed -o run -- python3 bench.py --steps 3 --warmup 1 --no-profile-pass --cpu-frames 0 > $out/bench_timed_only_under_rocprof.json 2> $out/rocprof_stats_timed.err || exit 2
find $out/stats_timed -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats_timed_only.csv
rm -rf $out/stats_timed
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/pmc_fetch -o runc -- python3 bench.py --frames 128 --steps 1 --warmup 1 --lanes 1 --cpu-frames 0 --no-region-leg --no-4k --no-noise-leg --no-class-api > /dev/null 2> $out/rocprof_fetch.err || exit 3
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/pmc_write -o runc -- python3 bench.py --frames 128 --steps 1 --warmup 1 --lanes 1 --cpu-frames 0 --no-region-leg --no-4k --no-noise-leg --no-class-api > /dev/null 2> $out/rocprof_write.err || exit 4
find $out/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
find $out/pmc_fetch -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/pmc_fetch.csv
find $out/pmc_write -name "*counter_collection.csv" | head -1 | xargs -I{} cp {} $out/pmc_write.csv
rm -rf $out/stats $out/pmc_fetch $out/pmc_write
ls -la $out
head -c 600 $out/bench.json
