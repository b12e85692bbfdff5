set -u
cd /tmp && export TMPDIR=/tmp
ROOT="${GRAFT_REPO_ROOT:-/root/repo}"
OUT=$ROOT/gpurun_out/narrow_exp
mkdir -p $OUT
for pad in 0 1; do
  export TRL_RING_PAD_EXPERIMENT=$pad   # (the switch existed for this experiment only: padding is the layout now)
  timeout -k 10 120 python3 $ROOT/bench.py --only gather_narrow_65536 > $OUT/line_pad$pad.json 2> $OUT/err_pad$pad.txt || exit 1
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $OUT/raw
    timeout -s KILL 120 rocprofv3 --pmc $c -d $OUT/raw -o x --output-format csv -- python3 $ROOT/bench.py --only gather_narrow_65536 --kernel-iters 6 > /dev/null 2> $OUT/pmc_err.txt || { tail -3 $OUT/pmc_err.txt; exit 1; }
    f=$(find $OUT/raw -name "*counter_collection.csv" | head -1)
    cp $f $OUT/pad${pad}_$c.csv
  done
done
rm -rf $OUT/raw
python3 - <<'PY'
import csv, json, os
out = os.path.join(os.environ.get('GRAFT_REPO_ROOT', '/root/repo'), 'gpurun_out/narrow_exp')
for pad in (0, 1):
  line = json.load(open('%s/line_pad%d.json' % (out, pad)))
  r = {}
  for c in ('FETCH_SIZE', 'WRITE_SIZE'):
    vals = [float(x['Counter_Value']) * 1024 for x in csv.DictReader(open('%s/pad%d_%s.csv' % (out, pad, c))) if x['Counter_Name'] == c and 'k_gather_narrow' in x['Kernel_Name']]
    r[c] = sum(vals) / max(1, len(vals))
  print('pad', pad, 'us', line['us_per_launch'], 'frac', line['frac'], 'fetch raw MB', r['FETCH_SIZE'] / 1e6, 'write MB', r['WRITE_SIZE'] / 1e6)
PY
