#!/bin/bash
# float64 modal STFT (stft5_kernel, ira_stft_logbin): its own instruction stream with and without memory, split into
# transform / conversion / aggregation by the tuning build's timing-only switches (IRA_STFT5_ABLATE: 1 no window loads,
# 2 no sample loads, 4 no dB -> float32 -> linear conversion), 256 x 10 s, + SQ counters of the product kernel.
#   bash tools/experiments/r5_stft5_rate.sh <outdir>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r5_stft5}; mkdir -p $O
export IRA_TUNING=1 IRA_LIBRARY=$R/audio_analysis_amd/csrc/libira_tuning.so
for rep in 1 2; do
  for a in 0 3 4 7; do echo -n "rep $rep ablate $a: "; IRA_STFT5_ABLATE=$a timeout -k 10 100 python3 $R/tools/block_probe.py --block modal --batch 256 --iters 4 2>&1 | grep "block=" | cut -c1-140; done
done
cd /tmp && export TMPDIR=/tmp
for a in 0 3; do
i=0
for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_VALU_TRANS_F64 SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64" "SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE" "SQ_INSTS_LDS SQ_INSTS_SALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32"; do
  i=$((i+1))
  IRA_STFT5_ABLATE=$a timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/a${a}_p$i -- python3 $R/tools/block_probe.py --block modal --batch 256 --iters 2 > $O/a${a}_p$i.log 2>&1 || echo "pmc $i failed"
done
python3 - $O a$a <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(f"{out}/{tag}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft5_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
d = {k: v / cnt[k] for k, v in acc.items()}
waves = d.get("SQ_WAVES", 1); frames = waves / 4.0
simd = d.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0
print(f"{tag}: frames {frames:.0f}  SIMD-cycles per frame {simd/frames:.0f}  VALU insts per wave {d.get('SQ_INSTS_VALU',0)/waves:.0f} "
      f"(f64 fma {d.get('SQ_INSTS_VALU_FMA_F64',0)/waves:.0f} add {d.get('SQ_INSTS_VALU_ADD_F64',0)/waves:.0f} mul {d.get('SQ_INSTS_VALU_MUL_F64',0)/waves:.0f} "
      f"trans64 {d.get('SQ_INSTS_VALU_TRANS_F64',0)/waves:.0f} cvt {d.get('SQ_INSTS_VALU_CVT',0)/waves:.0f} int32 {d.get('SQ_INSTS_VALU_INT32',0)/waves:.0f})  "
      f"VALU-active cycles per wave {4*d.get('SQ_ACTIVE_INST_VALU',0)/waves:.0f}  LDS insts per wave {d.get('SQ_INSTS_LDS',0)/waves:.0f} LDS-active cycles per wave {4*d.get('SQ_ACTIVE_INST_LDS',0)/waves:.0f}  "
      f"SALU per wave {d.get('SQ_INSTS_SALU',0)/waves:.0f}  waiting {d.get('SQ_WAIT_INST_ANY',0)/max(d.get('SQ_WAVE_CYCLES',1),1):.1%} of wave-cycles  resident waves/SIMD {4*d.get('SQ_WAVE_CYCLES',0)/simd:.2f}  bank conflicts {d.get('SQ_LDS_BANK_CONFLICT',0)/max(d.get('SQ_LDS_IDX_ACTIVE',1),1):.1%}")
PY
done
