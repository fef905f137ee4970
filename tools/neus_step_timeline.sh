# usage (on the GPU box): bash tools/neus_step_timeline.sh neus-blender|neus-dtu|neuralangelo  ->  gpurun_out/<config>_refresh_timeline.csv:
# rocprofv3 --kernel-trace of tools/neus_operating_point.py, cut to the last occupancy-refresh step and the two steps after it
# (start_us, dur_us, queue, kernel)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
c=${1:-neus-dtu}
rm -rf /tmp/pt && timeout 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/pt -o t -- python tools/neus_operating_point.py $c 100 > /dev/null 2>&1
f=$(find /tmp/pt -name "*kernel_trace.csv" | head -1)
python3 - "$f" "$c" <<'P'
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
idx=[i for i,r in enumerate(rows) if "k_occ_make_samples" in r["Kernel_Name"]]
# last refresh: from the k_prepare_train_rays / first occ kernel to the next k_adamw_multi
lo=idx[-2] if len(idx)>1 else idx[-1]
lo=max(0,lo-3)
hi=lo
n_adam=0
while hi<len(rows)-1 and n_adam<2:
    hi+=1
    if "k_adamw_multi" in rows[hi]["Kernel_Name"]: n_adam+=1
t0=int(rows[lo]["Start_Timestamp"])
out=open(f"/root/repo/gpurun_out/{sys.argv[2]}_refresh_timeline.csv","w")
out.write("start_us,dur_us,queue,kernel\n")
for r in rows[lo:hi+1]:
    n=r["Kernel_Name"].replace("(anonymous namespace)::","").replace("void ","")
    n=n[:n.find("(")] if "(" in n else n
    out.write(f"{(int(r['Start_Timestamp'])-t0)/1e3:.1f},{(int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3:.1f},{r.get('Queue_Id','')},{n[:70]}\n")
print("rows", hi-lo)
P
