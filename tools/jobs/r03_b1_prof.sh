cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r03/b1; mkdir -p $O; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -- python tools/b1_profile.py 20 2>/dev/null | tail -2
S=$(ls -t $O/prof/*/*_kernel_stats.csv | head -1); cp $S $O/b1_kernel_stats.csv
python - <<PY
import csv
rows = list(csv.DictReader(open("$O/b1_kernel_stats.csv")))
tot = sum(float(r["TotalDurationNs"]) for r in rows)
print("total kernel time per utterance: %.3f ms over %d launches" % (tot / 23 / 1e6, sum(int(r["Calls"]) for r in rows) / 23))
for r in rows[:26]:
    print("%-72s %6d calls/utt %8.1f us avg %6.1f %%  min %6.1f max %7.1f" % (r["Name"][:72], int(r["Calls"]) / 23, float(r["AverageNs"]) / 1e3, float(r["Percentage"]), float(r["MinNs"]) / 1e3, float(r["MaxNs"]) / 1e3))
PY
rm -f $O/prof/*/*kernel_trace.csv
