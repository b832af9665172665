#!/bin/bash
# round 4, session o: SQ counters of the broadphase sweep (k_sweep_rows) in a running cfg 2 world
cd /tmp && export TMPDIR=/tmp; R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r4o; rm -rf $O; mkdir -p $O
timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_SALU --output-format csv -d $O -o sq1 -- python $R/tools/steady.py 12 --no-phase-timing > $O/s1.txt 2> $O/sq1.err
timeout 600 rocprofv3 --pmc SQ_WAVES SQ_INSTS_LDS SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_VMEM_RD SQ_INSTS_SMEM --output-format csv -d $O -o sq2 -- python $R/tools/steady.py 12 --no-phase-timing > $O/s2.txt 2> $O/sq2.err
timeout 600 rocprofv3 --pmc SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT --output-format csv -d $O -o sq3 -- python $R/tools/steady.py 12 --no-phase-timing > $O/s3.txt 2> $O/sq3.err
python - <<'PY'
import csv, glob, collections, os
O=os.environ.get("GRAFT_REPO_ROOT","/root/repo")+"/gpurun_out/r4o"
for f in sorted(glob.glob(O+"/*counter_collection.csv")):
    acc=collections.defaultdict(lambda: collections.defaultdict(float)); n=collections.Counter()
    for r in csv.DictReader(open(f)):
        k=r["Kernel_Name"].split("(")[0].replace("void phx::","").replace("phx::","")
        if "k_sweep_rows" not in k and "k_build_bin" not in k and "k_joint_components" not in k: continue
        acc[k][r["Counter_Name"]]+=float(r["Counter_Value"]); 
        if r["Counter_Name"]==list(acc[k].keys())[0]: n[k]+=1
    for k,c in acc.items():
        print(os.path.basename(f)[:4], k, "launches", n[k], {a: round(b/max(n[k],1)) for a,b in c.items()})
PY
