# rocprofv3 kernel trace of the OBJ scenes at the default table step: per-kernel average durations.
#   gpurun -- 'bash tools/prof_table.sh r03_obj_a'      (summary: gpurun_out/<tag>/summary.txt)
set -e
TAG=${1:-r03_obj}
R=${GRAFT_REPO_ROOT:-/root/repo}; OUT=$R/gpurun_out/$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
for cfg in "teapot 2048 0.05" "monkey 4096 0.05"; do
  tag=$(echo $cfg | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o p -- python3 $R/tools/prof_object.py $cfg > $OUT/$tag.log 2>&1
  f=$(find $OUT/stats_$tag -name '*kernel_stats.csv' | head -1)
  echo "== $cfg" >> $OUT/summary.txt
  python3 - "$f" >> $OUT/summary.txt <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    n = r['Name'].split('(')[0].replace('void vrtk::', '')
    print(f"{n[:70]:70s} calls {r['Calls']:>4s} avg {float(r['AverageNs'])/1e6:9.4f} ms  total {float(r['TotalDurationNs'])/1e6:9.3f} ms")
PY
done
cat $OUT/summary.txt
