# Kernel times of the reference's unmodified train.py in its first training phase (tools/loop_rate.py, mode auto, the child under
# rocprofv3 --kernel-trace --stats):   bash tools/gpu_loop_trace.sh [epochs]   -> gpurun_out/loop_trace.txt
mkdir -p gpurun_out
R=$PWD
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/lp
LOOP_TRACE_DIR=/tmp/lp timeout 600 python $R/tools/loop_rate.py --modes auto --epochs ${1:-14} --out /tmp/lp_out.json > /tmp/lp.log 2>&1
tail -2 /tmp/lp.log > $R/gpurun_out/loop_trace.txt
for f in $(find /tmp/lp -name "*kernel_stats.csv"); do python - $f >> $R/gpurun_out/loop_trace.txt <<PY
import csv,sys
rows=list(csv.DictReader(open(sys.argv[1])))
fw=[r for r in rows if "mlp_fwd_kernel<256, true" in r["Name"]]
n=int(fw[0]["Calls"]) if fw else 1
print("steps", n, "launches/step %.1f" % (sum(int(r["Calls"]) for r in rows)/n), "GPU busy ms/step %.3f (under the tracer)" % (sum(float(r["TotalDurationNs"]) for r in rows)/n/1e6))
for r in rows[:24]:
    print("%8.2f calls/step %9.1f us avg  %s" % (int(r["Calls"])/n, float(r["AverageNs"])/1e3, r["Name"][:90]))
PY
done
cat $R/gpurun_out/loop_trace.txt
