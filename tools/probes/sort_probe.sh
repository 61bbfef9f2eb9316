# (GPU box) durations of the one-workgroup std::sort replay for a few list shapes
O=$GRAFT_REPO_ROOT/gpurun_out/sort_probe; mkdir -p $O; cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace -d $O/trace -o sp --output-format csv -- python $GRAFT_REPO_ROOT/tools/probes/sort_probe.py > $O/cases.txt 2> $O/err.txt
python - <<'P'
import csv, glob, os
O = os.environ.get("GRAFT_REPO_ROOT", "/root/repo") + "/gpurun_out/sort_probe"
f = glob.glob(O + "/trace/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f)) if "k_debug_pileup_order" in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
cases = [l.strip() for l in open(O + "/cases.txt") if l.strip() and l[0].isalpha() or l[:1].isdigit()]
cases = [l for l in cases if l.split()[-1].isdigit()]
for k, c in enumerate(cases):
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in rows[5 * k:5 * k + 5]]
    if d: print("%-22s %6.1f us (min of 5: %.1f)" % (c, sorted(d)[len(d) // 2] / 1e3, min(d) / 1e3))
P
