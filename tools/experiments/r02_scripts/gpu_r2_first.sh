#!/bin/bash
# round 2, first GPU call: full GPU test suite, FETCH_SIZE calibration, default bench, 2-rank rehearsal
export TMPDIR=/tmp; R=$PWD; O=$R/gpurun_out/r2a; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q --timeout 600 > $O/tests.log 2>&1; echo "tests rc=$?"; tail -3 $O/tests.log
cd /tmp
timeout -k 10 200 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/calib -- $R/tools/fetch_calib_probe > $O/calib.log 2>&1; echo "calib rc=$?"
cd $R
cat $O/calib.log | grep -v "^W2\|^E2" | tail -12
python3 - <<'PY'
import csv, glob, collections
for f in glob.glob("gpurun_out/r2a/calib/*/*counter_collection.csv"):
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        acc[r["Kernel_Name"][:60]].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        print("FETCH_SIZE_KB %-62s %s" % (k, ["%.0f" % x for x in v]))
PY
timeout -k 10 500 python bench.py > $O/bench_c3.log 2>&1; echo "bench rc=$?"; grep -o '{"metric.*' $O/bench_c3.log > $O/bench_c3.json; python3 -c "
import json; d=json.load(open('$O/bench_c3.json')); print('value %.4g incl %.4g ms/step %.3f frac %.3f' % (d['value'], d['value_incl_transfers'], d['ms_per_step'], d['roofline']['frac'])); print(d['stage_ms_per_step']); print(d['incl_transfers_note']); print(d['cpu_baseline']['value'], d['cpu_baseline']['sample']); print('flat', d['flat_noise_structured_solve']['roofline'])"
HPX_BENCH_DEVICE=0 HPX_BENCH_BACKEND=gloo timeout -k 10 400 python bench.py --gpus 2 --steps 10 --warmup 2 --nbl 512 > $O/bench_2rank.log 2>&1; echo "2-rank rc=$?"; grep -o '{"metric.*' $O/bench_2rank.log | python3 -c "
import json,sys; d=json.loads(sys.stdin.read()); print('n_gpus', d['n_gpus'], 'total', d['config']['baselines_total'], 'value %.4g' % d['value'])"
