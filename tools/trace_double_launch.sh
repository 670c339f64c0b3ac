#!/bin/bash
R=${GRAFT_REPO_ROOT:-/root/repo}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/r3_dbl -- python3 $R/tools/exp_sector.py 12 5 --sector-only --opt=sector_sweep_dbg=16 > $R/gpurun_out/r3_dbl.log 2>&1
python3 - <<'PY'
import csv, glob, os
R=os.environ.get("GRAFT_REPO_ROOT","/root/repo")
f=glob.glob(R+"/gpurun_out/r3_dbl/**/*kernel_trace.csv", recursive=True)[0]
rows=[r for r in csv.DictReader(open(f)) if "sector_sweep2" in r["Kernel_Name"]]
rows.sort(key=lambda r:int(r["Start_Timestamp"]))
d=[int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in rows]
tail=d[-92:]
cold=tail[0::2]; warm=tail[1::2]
print("last evaluation: cold avg %.2f us, warm avg %.2f us"%(sum(cold)/len(cold)/1e3, sum(warm)/len(warm)/1e3))
print("cold", [round(x/1e3,1) for x in cold[:16]])
print("warm", [round(x/1e3,1) for x in warm[:16]])
gaps=[int(rows[i+1]["Start_Timestamp"])-int(rows[i]["End_Timestamp"]) for i in range(len(rows)-92,len(rows)-1)]
print("gaps avg %.2f us"%(sum(gaps)/len(gaps)/1e3))
PY
rm -rf $R/gpurun_out/r3_dbl
