cd $GRAFT_REPO_ROOT
tools/prof_stats.sh c1diag 200 python tools/diag_c1_trace.py | head -30
f=$(find gpurun_out/prof_c1diag -name "*kernel_trace.csv" | head -1)
python - "$f" <<'PY'
import csv, sys, collections
rows = list(csv.DictReader(open(sys.argv[1])))
by = collections.defaultdict(list)
for r in rows:
    by[r["Kernel_Name"].split("(")[0][:60]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]) - int(r["Start_Timestamp"])))
for k, v in by.items():
    if len(v) < 100: continue
    v.sort()
    d = [x[1] / 1e3 for x in v]
    print(k, len(d), "us by 20-launch bucket:", " ".join("%.0f" % (sum(d[i:i + 20]) / len(d[i:i + 20])) for i in range(0, len(d), 20)))
PY
cat gpurun_out/prof_c1diag/run.log | grep frame | head -30
