set -e
R=${GRAFT_REPO_ROOT:-$PWD}; OUT=$R/gpurun_out/pmc2; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_MISC SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $OUT -o a -- python3 $R/tools/profile_run.py > $OUT/a.log 2>&1
timeout -k 10 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INST_CYCLES_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_ANY --output-format csv -d $OUT -o b -- python3 $R/tools/profile_run.py > $OUT/b.log 2>&1
cd $R
python3 - <<'P'
import csv, collections, glob, os
out=os.path.join(os.environ.get("GRAFT_REPO_ROOT","."),"gpurun_out","pmc2")
for tag in ("a","b"):
    for fn in glob.glob(out+"/%s_counter_collection.csv"%tag):
        acc=collections.defaultdict(lambda: collections.defaultdict(list))
        for r in csv.DictReader(open(fn)):
            k=r["Kernel_Name"]
            if "k_gamma_scan<double, 8>" in k or "k_solve_gcf_g" in k:
                acc[k.split("(")[0][-40:]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        for k,v in acc.items():
            print(tag, k, {c: round(sum(x)/len(x)) for c,x in v.items()})
P
