set -e
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/r02_obj; rm -rf $OUT; mkdir -p $OUT
for cfg in "teapot 2048" "monkey 4096" "monkey 4096 0.12"; do
  tag=$(echo $cfg | tr ' .' '__')
  rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/stats_$tag -o p -- python3 $R/tools/prof_object.py $cfg > $OUT/$tag.log 2>&1
  rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $OUT/pmc_$tag -o p -- python3 $R/tools/prof_object.py $cfg >> $OUT/$tag.log 2>&1
  echo "$tag done"
done
cd $R
OUT=$OUT python3 - <<'PY'
import csv, glob, collections, os
OUT=os.environ['OUT']
PEAK=1024*2.4e9/2
print("# rocprofv3 on the OBJ scenes (`tools/prof_object.py`, 5 frames each, MI355X, final kernels of round 2)\n")
print("`--kernel-trace --stats` per-launch averages, and a separate `--pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU` pass: VALU wave-instructions")
print("per launch, their rate against the spec issue peak (1024 SIMDs x 2.4 GHz / 2 cycles = 1.2288e12 wave-instructions/s) and the")
print("share of the launch the VALU pipes are busy (VALUBusy = SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x GRBM_GUI_ACTIVE); the clock is\nGRBM_GUI_ACTIVE / launch duration).\n")
print("| scene | kernel | launches | avg ms | SQ_INSTS_VALU / launch | of spec issue peak | VALU pipes busy |")
print("|---|---|---|---|---|---|---|")
for tag,name in (("teapot_2048","`-f teapot.obj -w 2048`"),("monkey_4096","`-f monkey.obj -w 4096`"),("monkey_4096_0_12","`-f monkey.obj -w 4096 --table-step 0.12`")):
    st=glob.glob(f"{OUT}/stats_{tag}/**/*kernel_stats.csv",recursive=True)[0]
    pm=glob.glob(f"{OUT}/pmc_{tag}/**/*counter_collection.csv",recursive=True)[0]
    cnt=collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(pm)):
        k=r['Kernel_Name'].split('(')[0].replace('void vrtk::','').replace('vrtk::','')
        cnt[k][r['Counter_Name']].append(float(r['Counter_Value']))
    for r in csv.DictReader(open(st)):
        k=r['Name'].split('(')[0].replace('void vrtk::','').replace('vrtk::','')
        if not any(x in k for x in ('render_','build_tile','order_dense','tile_cones')): continue
        avg=float(r['AverageNs'])/1e6
        iv=cnt[k].get('SQ_INSTS_VALU'); av=cnt[k].get('SQ_ACTIVE_INST_VALU')
        n=sum(iv)/len(iv) if iv else None; a=sum(av)/len(av) if av else None
        frac=f"{n/(avg*1e-3)/PEAK*100:.0f} %" if n and avg>0.05 else ""
        gv=cnt[k].get('GRBM_GUI_ACTIVE'); gcyc=(sum(gv)/len(gv)) if gv else None
        busy=(f"{a*4/(1024*gcyc)*100:.0f} % (clock {gcyc/(avg*1e-3)/1e9:.2f} GHz)" if (a and gcyc and avg>0.05) else (f"{a*4/(1024*2.4e9*avg*1e-3)*100:.0f} % at 2.4 GHz" if a and avg>0.05 else ""))
        print(f"| {name} | `{k}` | {r['Calls']} | {avg:.3f} | {n:.3e} | {frac} | {busy} |" if n else f"| {name} | `{k}` | {r['Calls']} | {avg:.3f} | | | |")
PY
