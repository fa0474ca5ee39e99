#!/bin/bash
# tools/pcsample.sh <workload> [method] [interval] — rocprofv3 PC sampling of one workload's kernel (run on the GPU box):
# per-instruction sample histogram with issue / stall reasons, aggregated into gpurun_out/pcs_<workload>_<method>.txt
export TMPDIR=/tmp
ROOT=$(pwd); WL=${1:-pipeline}; METHOD=${2:-stochastic}; INTERVAL=${3:-65536}
UNIT=cycles; [ "$METHOD" = host_trap ] && UNIT=time
OUT=$ROOT/gpurun_out/pcs_${WL}_$METHOD; rm -rf $OUT; mkdir -p $OUT
cd /tmp
timeout 240 rocprofv3 --pc-sampling-beta-enabled --pc-sampling-unit $UNIT --pc-sampling-method $METHOD --pc-sampling-interval $INTERVAL \
  --kernel-trace --output-format csv -d $OUT -o pcs -- python $ROOT/bench.py --workload $WL --also none --no-cpu-baseline --no-verify --steps 3 --warmup 1 --launches-per-step 2 > $OUT/log.txt 2>&1
cd $ROOT
tail -5 $OUT/log.txt
find $OUT -name "*.csv" | head; 
python - "$OUT" "$WL" "$METHOD" <<'PY'
import csv, sys, glob, collections, os
out, wl, method = sys.argv[1:4]
files = [f for f in glob.glob(out + "/**/*pc_sampling*.csv", recursive=True)]
res = open(f"gpurun_out/pcs_{wl}_{method}.txt", "w")
for f in files:
    rows = list(csv.DictReader(open(f)))
    print(f, len(rows), "samples", file=res)
    if not rows: continue
    print("columns:", list(rows[0].keys()), file=res)
    def col(*names):
        for n in names:
            if n in rows[0]: return n
        return None
    ins, typ, stall, issued = col("Instruction"), col("Instruction_Type"), col("Stall_Reason"), col("Wave_Issued_Instruction")
    for key, title in ((typ, "instruction type"), (stall, "stall reason"), (issued, "issued")):
        if key:
            c = collections.Counter(r[key] for r in rows)
            print(f"-- by {title}", file=res)
            for k, v in c.most_common(20): print(f"{v:9d} {100.0*v/len(rows):6.2f}%  {k}", file=res)
    if ins:
        c = collections.Counter((r[ins].split()[0] if r[ins] else "?") for r in rows)
        print("-- by opcode", file=res)
        for k, v in c.most_common(40): print(f"{v:9d} {100.0*v/len(rows):6.2f}%  {k}", file=res)
        if stall:
            c = collections.Counter(((r[ins].split()[0] if r[ins] else "?"), r[stall], r.get(issued, "")) for r in rows)
            print("-- by opcode x stall x issued", file=res)
            for k, v in c.most_common(60): print(f"{v:9d} {100.0*v/len(rows):6.2f}%  {k}", file=res)
res.close()
print(open(f"gpurun_out/pcs_{wl}_{method}.txt").read()[:6000])
PY
rm -rf $OUT/*/*.db 2>/dev/null
du -sh $OUT
