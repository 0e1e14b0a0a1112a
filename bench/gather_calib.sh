cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/calib
mkdir -p $O
cd $R
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O -o calib -- ./bench/gather_calib > $O/stdout.txt 2> $O/err.txt
cat $O/stdout.txt
python3 - <<'PY'
import csv, glob, collections
f = glob.glob('/root/repo/gpurun_out/calib/**/calib_counter_collection.csv', recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if r['Counter_Name'] == 'FETCH_SIZE']
for i, r in enumerate(rows):
    print(i, r['Kernel_Name'][:40], 'FETCH_SIZE x 1024 = %.1f MB' % (float(r['Counter_Value']) * 1024 / 1e6))
PY
