#!/bin/bash
# r06: rocprofv3 kernel statistics of the bench command (the driver's command, shorter) -> gpurun_out/r06/prof_bench/
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/r06/prof_bench
mkdir -p $O
cd $R
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 bench.py --steps 20 --warmup 3 --no-cpu --no-config-legs > $O/bench_under_rocprof.json 2> $O/bench_under_rocprof.err
cp $O/stats/*/*kernel_stats.csv $O/bench_kernel_stats.csv 2>/dev/null
python3 - <<PY
import csv,glob
rows=[]
for f in glob.glob("$O/stats/*/*kernel_trace.csv"):
    for r in csv.DictReader(open(f)):
        if "scan_piece_kernel" in r["Kernel_Name"] or "sp_refine_kernel" in r["Kernel_Name"]:
            rows.append((r["Kernel_Name"][:60], (int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6))
big=[t for n,t in rows if t>5]
print("main scan launches under the profiler:", len(big), "mean ms %.3f" % (sum(big)/max(1,len(big))), "min %.3f max %.3f" % (min(big), max(big)) if big else "")
open("$O/scan_launches.txt","w").write("\n".join(f"{n}\t{t:.3f}" for n,t in rows))
PY
find $O -name "*.csv" -size +3M -delete
head -16 $O/bench_kernel_stats.csv
