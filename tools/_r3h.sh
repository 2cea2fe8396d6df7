R=$PWD
cd /tmp && export TMPDIR=/tmp
for lv in 6; do
O=$R/gpurun_out/sortprobe$lv; rm -rf $O; mkdir -p $O
timeout -k 5 300 rocprofv3 --kernel-trace --output-format csv -d $O -o r1 -- python $R/tools/sort_probe.py $lv 8 > $O/log.txt 2>&1
tail -5 $O/log.txt
python - $O <<'PY'
import sys, glob, pandas as pd
f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
t = pd.read_csv(f).sort_values("Start_Timestamp")
t = t[t["Kernel_Name"].str.contains("k_closest")]
d = ((t["End_Timestamp"] - t["Start_Timestamp"]) / 1e3).round(0).tolist()
print("k_closest hook launches (us):", d)
PY
done
