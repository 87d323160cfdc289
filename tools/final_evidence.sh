#!/bin/bash
# one GPU-box call that refreshes the measured evidence of the current build -> gpurun_out/evidence/ (copy into profiles/):
#   bench line, the same command under rocprofv3 --kernel-trace --stats (kernel stats CSV + its bench line), the PMC traffic
#   record of the metric kernel (tools/pmc_record.sh), the per-config timings (tools/cfgbench.py), per-pass long transforms
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
E=gpurun_out/evidence
rm -rf $E; mkdir -p $E
bash tools/pmc_record.sh 2.0 > $E/pmc_record.log 2>&1          # first: bench.py reports roofline.traffic from this record
cp gpurun_out/pmc_record/pmc_metric_kernel_current.json $E/ 2>/dev/null
python3 bench.py --steps 200 --warmup 20 > $E/bench.json 2> $E/bench.err || echo "bench failed"
rocprofv3 --kernel-trace --stats --output-format csv -d $E/rocprof -- python3 bench.py --steps 100 --warmup 10 --cpu-log2n 24 > $E/bench_under_rocprof.json 2> $E/rocprof.err || echo "rocprof bench failed"
cp $(ls $E/rocprof/*/*kernel_stats.csv | head -1) $E/bench_kernel_stats.csv
python3 tools/cfgbench.py > $E/cfgbench.txt 2>&1 || echo "cfgbench failed"
bash tools/longpass_prof.sh > $E/long_passes.txt 2>&1
python3 tools/longbench.py > $E/longbench.txt 2>&1
head -c 600 $E/bench.json; echo; head -5 $E/bench_kernel_stats.csv | cut -c1-160; tail -n 12 $E/pmc_record.log; tail -n 30 $E/cfgbench.txt
