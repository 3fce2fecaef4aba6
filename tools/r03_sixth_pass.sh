set -o pipefail
O=gpurun_out/r03f; mkdir -p $O
timeout -k 10 200 ./crypto12381_amd/bin/valu_rates $O/valu_rates.txt > $O/valu_rates.log 2>&1; echo "valu_rates rc=$?"; grep "v_mad_i64_i32 " $O/valu_rates.txt
for v in default msm28 default msm28; do
  if [ $v = default ]; then unset C12381_LIB; else export C12381_LIB=$PWD/crypto12381_amd/lib/exp/lib$v.so; fi
  echo "== $v" >> $O/ab.txt
  timeout -k 10 300 python tools/g2_mul_bench.py 2>&1 | grep -v amdgpu.ids >> $O/ab.txt || exit 1
done
unset C12381_LIB
cat $O/ab.txt
ROOT=$PWD
cd /tmp && export TMPDIR=/tmp
for set in "GRBM_GUI_ACTIVE" "WRITE_SIZE" "FETCH_SIZE"; do
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$O/pmc/$set" -o p -- python3 "$ROOT/tools/prof_driver3.py" g2 > "$ROOT/$O/pmc_$set.log" 2>&1
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $set --output-format csv -d "$ROOT/$O/pmc_msm/$set" -o p -- python3 "$ROOT/tools/prof_driver3.py" msm > "$ROOT/$O/pmc_msm_$set.log" 2>&1
done
cd $ROOT
python - <<'PY'
import csv, glob
for sub in ("pmc", "pmc_msm"):
    for tag in ("GRBM_GUI_ACTIVE", "WRITE_SIZE", "FETCH_SIZE"):
        for path in glob.glob("gpurun_out/r03f/%s/%s/*counter_collection.csv" % (sub, tag)):
            best = {}
            for r in csv.DictReader(open(path)):
                name = r["Kernel_Name"].split("(")[0]
                if "g2_mul2" not in name and "msm_bucket" not in name: continue
                dur = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6
                v = float(r["Counter_Value"])
                k = (name, r["Dispatch_Id"])
                best[k] = (best.get(k, (0, 0))[0] + v, dur)
            for k, (v, dur) in best.items():
                print(sub, tag, k[0], "%.3f ms" % dur, "value %.4g" % v, ("clock %.3f GHz" % (v / 8 / dur / 1e6)) if tag.startswith("GRBM") else "%.3f GB" % (v * 1024 / 1e9))
PY
