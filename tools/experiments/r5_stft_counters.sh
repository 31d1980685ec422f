#!/bin/bash
# SQ counters of the float32 STFT kernel (tools/stft_probe.py --tf --batch 256), builds x ablations on one box.
#   bash tools/experiments/r5_stft_counters.sh <outdir> <other tuning .so>
R=$GRAFT_REPO_ROOT; O=$R/${1:-gpurun_out/r5_stft_cnt}; other=$2; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
for arm in "new:audio_analysis_amd/csrc/libira_tuning.so:0" "new:audio_analysis_amd/csrc/libira_tuning.so:7" "poly:audio_analysis_amd/csrc/libira_tuning.so:16" "poly:audio_analysis_amd/csrc/libira_tuning.so:23" "prev:$other:0" "prev:$other:7"; do
  IFS=: read name lib ab <<< "$arm"
  i=0
  for grp in "SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VALU2 SQ_THREAD_CYCLES_VALU SQ_INSTS_VALU_TRANS_F32 SQ_BUSY_CU_CYCLES" "SQ_IFETCH SQ_IFETCH_LEVEL SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_SALU" "SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE SQ_INSTS_SALU"; do
    i=$((i+1)); d=$O/${name}_ab${ab}_p$i
    IRA_TUNING=1 IRA_LIBRARY=$R/$lib IRA_STFT6_ABLATE=$ab timeout -k 10 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $d -- python3 $R/tools/stft_probe.py --tf --batch 256 --iters 3 > $d.log 2>&1 || echo "failed $arm $i"
  done
  python3 - $O ${name}_ab${ab} <<'PY'
import csv, glob, sys, collections
out, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(float); cnt = collections.Counter()
for f in glob.glob(f"{out}/{tag}_p*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "stft6_kernel" in r["Kernel_Name"]:
            acc[r["Counter_Name"]] += float(r["Counter_Value"]); cnt[r["Counter_Name"]] += 1
d = {k: v / cnt[k] for k, v in acc.items()}
frames = 256 * 929.0
simd = d.get("GRBM_GUI_ACTIVE", 0) / 8.0 * 1024.0
print(f"{tag:12s} SIMD-cycles/frame {simd/frames:7.0f} | VALU insts/frame {d.get('SQ_INSTS_VALU',0)/frames:6.0f} trans {d.get('SQ_INSTS_VALU_TRANS_F32',0)/frames:4.0f} | "
      f"VALU busy cyc/frame {4*d.get('SQ_ACTIVE_INST_VALU',0)/frames:6.0f} (VALU2 {4*d.get('SQ_ACTIVE_INST_VALU2',0)/frames:6.0f}) thread-cycles/64/frame {d.get('SQ_THREAD_CYCLES_VALU',0)/64/frames:6.0f} | "
      f"SCA busy {4*d.get('SQ_ACTIVE_INST_SCA',0)/frames:6.0f} SALU cyc {4*d.get('SQ_INST_CYCLES_SALU',0)/frames:6.0f} LDS busy {4*d.get('SQ_ACTIVE_INST_LDS',0)/frames:6.0f} MISC {4*d.get('SQ_ACTIVE_INST_MISC',0)/frames:5.0f} | "
      f"ifetch {d.get('SQ_IFETCH',0)/frames:6.1f} ifetch-level {d.get('SQ_IFETCH_LEVEL',0)/frames:8.0f} | wave-cycles/frame {d.get('SQ_WAVE_CYCLES',0)/frames*4:7.0f} wait-inst {d.get('SQ_WAIT_INST_ANY',0)/frames*4:7.0f}")
PY
done
