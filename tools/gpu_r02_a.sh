#!/bin/bash
# GPU session A: test suite, bench, error tables, shard-size sweep, cache-counter passes over the bench
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r2a; mkdir -p $O
export TMPDIR=/tmp
timeout -k 10 900 python -u -m pytest tests -m gpu -x -q --timeout 300 --timeout-method thread > $O/tests.log 2>&1; rc=$?
tail -5 $O/tests.log
[ $rc -ne 0 ] && exit $rc
timeout -k 10 300 python bench.py > $O/bench.log 2>&1 || exit 2
tail -1 $O/bench.log | cut -c1-400
timeout -k 10 300 python tools/err_report.py > $O/err_report.log 2>&1 || exit 3
timeout -k 10 300 python tools/stage_err.py > $O/stage_err.log 2>&1 || exit 4
for b in 32 16 8 4; do
  timeout -k 10 200 python bench.py --no-cpu-baseline --batch $b --steps 30 > $O/bench_b$b.log 2>&1 || exit 5
  python - <<PY
import json
l=json.loads(open("$O/bench_b$b.log").read().strip().splitlines()[-1]); print("B=$b", l["ms_per_step"], l["value"])
PY
done
cd /tmp
R=$GRAFT_REPO_ROOT
timeout -s KILL 120 rocprofv3 --pmc TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum --kernel-trace --output-format csv -d $R/$O/pmc_tcp -o tcp -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/$O/pmc_tcp.log 2>&1 || exit 6
timeout -s KILL 120 rocprofv3 --pmc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum --kernel-trace --output-format csv -d $R/$O/pmc_tcc -o tcc -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/$O/pmc_tcc.log 2>&1 || exit 7
timeout -s KILL 120 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_VALU SQ_INST_CYCLES_VMEM GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/$O/pmc_sq -o sq -- python3 $R/bench.py --no-cpu-baseline --steps 5 --warmup 2 > $R/$O/pmc_sq.log 2>&1 || exit 8
cd $R
python tools/pmc_summary.py $O/pmc_caches.json $O/pmc_tcp $O/pmc_tcc $O/pmc_sq > $O/pmc_summary.log 2>&1
tail -12 $O/pmc_summary.log
exit 0
