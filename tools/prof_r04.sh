# Round-4 profiles on the MI355X box:  gpurun -- 'bash tools/prof_r04.sh'   (results under gpurun_out/r04p, copied to profiles/ by hand)
set -e
R=${GRAFT_REPO_ROOT:-/root/repo}
OUT=$R/gpurun_out/r04p
PART=${1:-all}
[ "$PART" = all -o "$PART" = bench ] && rm -rf $OUT
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
if [ "$PART" = all -o "$PART" = bench ]; then
# --- PMC passes on the bench configuration (separate passes, --pmc with --kernel-trace only) ---
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc_$c -o pmc -- python3 $R/tools/pmc_workload.py > $OUT/pmc_$c.log 2>&1
done
rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $OUT/pmc_SQ -o pmc -- python3 $R/tools/pmc_workload.py > $OUT/pmc_SQ.log 2>&1
echo "pmc done"
cd $R && python3 tools/pmc_traffic.py $OUT > $OUT/pmc_traffic.json && cp $OUT/pmc_traffic.json profiles/r04_pmc_traffic.json
# --- bench under rocprofv3 --stats, then plain ---
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats -o bench -- python3 $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.json 2> $OUT/stats.log
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_serial -o bench -- python3 $R/bench.py --frames-in-flight 1 --no-cpu-baseline > $OUT/bench_serial_under_rocprof.json 2>> $OUT/stats.log
cd $R
find $OUT/stats -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_kernel_stats.csv
find $OUT/stats_serial -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $OUT/bench_serial_kernel_stats.csv
python3 bench.py > $OUT/bench.json 2> $OUT/bench.log
python3 bench.py --steps 20 --warmup 5 > $OUT/bench_20steps.json 2>> $OUT/bench.log
python3 bench.py --frames-in-flight 1 --no-cpu-baseline > $OUT/bench_serial.json 2>> $OUT/bench.log
python3 tools/timeline.py 64 2048 2> $OUT/timeline.txt > /dev/null
echo "bench done"
python3 -c "
import json
for n in ('bench','bench_20steps','bench_serial','bench_under_rocprof'):
    d=json.loads([l for l in open('$OUT/'+n+'.json') if l.startswith('{')][-1])
    print(n, round(d['value']), 'Mrays/s', round(d['ms_per_step']*1e3,2), 'us/step', 'serial', round(d['serial']['ms_per_step']*1e3,2), 'kernel', round(d['roofline']['kernel_ms']*1e3,2), d['roofline'].get('traffic'), d['roofline']['frac'], d['valu'].get('executed',{}).get('frac'), d['valu']['algorithmic']['frac'], (d.get('moving_camera') or {}).get('ms_per_step'))
"
head -8 $OUT/bench_serial_kernel_stats.csv | cut -c1-200
fi
if [ "$PART" = all -o "$PART" = objects ]; then
# --- OBJ scenes: stats + instruction counters, default table mode and exact ---
cd /tmp
for cfg in "teapot 2048 0.05" "monkey 4096 0.05" "teapot 2048 0" "monkey 4096 0"; do
  tag=$(echo $cfg | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/obj_stats_$tag -o p -- python3 $R/tools/prof_object.py $cfg > $OUT/obj_$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/obj_pmc_$tag -o p -- python3 $R/tools/prof_object.py $cfg >> $OUT/obj_$tag.log 2>&1
done
cd $R
OUT=$OUT python3 - > $OUT/objects.md <<'PY'
import csv, glob, collections, os
OUT = os.environ['OUT']
PEAK = 1024 * 2.4e9 / 2
print("# rocprofv3 on the OBJ scenes, round 4 (`tools/prof_r04.sh`, `tools/prof_object.py`: 5 frames each, MI355X)\n")
print("`--kernel-trace --stats` per-launch averages and a separate `--pmc SQ_INSTS_VALU GRBM_GUI_ACTIVE` pass: VALU wave-instructions per launch,")
print("their rate against the spec issue peak (1024 SIMDs x 2.4 GHz / 2 cycles = 1.2288e12 /s), the clock during the launch (GRBM_GUI_ACTIVE summed")
print("over the 8 XCDs / 8 / duration) and cycles per executed instruction per SIMD.  Table step 0.05 = the default; 0 = the exact kernels.\n")
print("| scene | kernel | launches | avg ms | SQ_INSTS_VALU / launch | of spec issue peak | clock | cycles per instruction per SIMD |")
print("|---|---|---|---|---|---|---|---|")
for tag, name in (("teapot_2048_0_05", "`-f teapot.obj -w 2048` (default)"), ("teapot_2048_0", "`-f teapot.obj -w 2048 --table-step 0`"),
                  ("monkey_4096_0_05", "`-f monkey.obj -w 4096` (default)"), ("monkey_4096_0", "`-f monkey.obj -w 4096 --table-step 0`")):
    st = glob.glob(f"{OUT}/obj_stats_{tag}/**/*kernel_stats.csv", recursive=True)[0]
    pm = glob.glob(f"{OUT}/obj_pmc_{tag}/**/*counter_collection.csv", recursive=True)[0]
    cnt = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pm)):
        k = r['Kernel_Name'].split('(')[0].replace('void vrtk::', '').replace('vrtk::', '')
        cnt[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(st)):
        k = r['Name'].split('(')[0].replace('void vrtk::', '').replace('vrtk::', '')
        if not any(x in k for x in ('render_', 'build_tile', 'order_dense', 'tile_cones')): continue
        avg = float(r['AverageNs']) / 1e6
        iv = cnt[k].get('SQ_INSTS_VALU'); gv = cnt[k].get('GRBM_GUI_ACTIVE')
        n = sum(iv) / len(iv) if iv else None
        g = (sum(gv) / len(gv)) if gv else None
        if n and avg > 0.05 and g:
            clock = g / 8 / (avg * 1e-3)
            print(f"| {name} | `{k}` | {r['Calls']} | {avg:.3f} | {n:.3e} | {n/(avg*1e-3)/PEAK*100:.0f} % | {clock/1e9:.2f} GHz | {avg*1e-3*clock/(n/1024):.2f} |")
        else:
            print(f"| {name} | `{k}` | {r['Calls']} | {avg:.3f} | {n:.3e} | | | |" if n else f"| {name} | `{k}` | {r['Calls']} | {avg:.3f} | | | | |")
PY
cp $OUT/objects.md profiles/r04_objects.md
cat $OUT/objects.md
fi
