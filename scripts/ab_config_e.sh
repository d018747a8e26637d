#!/bin/bash
# Round 6: config E (SRIF 12/6 fp32, 262 144 filters) on the two-lane kernel against the FOUR-lane split kernel (diagnostic instantiation
# srif_split_kernel<float, 12, 6, 4>, variants e4l2 / e4l4 of scripts/build_variant_multi.sh), inside one gpurun call: timings (three
# alternations), parity of the four-lane instantiation against the oracle, SQ counters and FETCH / WRITE traffic of both.
export KB_SRIF_SPLIT_ALL=1 TMPDIR=/tmp
OUT=gpurun_out/ab_config_e; mkdir -p $OUT
scripts/ab_libs.sh "python scripts/bench_srif_sizes.py 65536 262144 524288 | cut -c1-200" 3 base e4l2 > $OUT/timings.txt 2>&1
cp gokalman_amd/libgokalman_amd.so /tmp/base_keep.so
cp gokalman_amd/_variants/libe4l2.so gokalman_amd/libgokalman_amd.so
python - > $OUT/parity_four_lane.txt 2>&1 <<'PY'
import sys; sys.path.insert(0, '.')
import bench, gokalman_amd as ga
from gokalman_amd import _capi as k, synth
print(bench._leg_parity(ga, k, synth, "srif_fp32"))
PY
SQ="SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
for v in base e4l2; do
  if [ $v = base ]; then cp /tmp/base_keep.so gokalman_amd/libgokalman_amd.so; else cp gokalman_amd/_variants/lib$v.so gokalman_amd/libgokalman_amd.so; fi
  rocprofv3 --pmc $SQ --output-format csv -d $OUT/sq_$v -- python3 scripts/bench_srif_sizes.py 262144 > $OUT/sq_$v.out 2>&1
  rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/fetch_$v -- python3 scripts/bench_srif_sizes.py 262144 > $OUT/fetch_$v.out 2>&1
  rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/write_$v -- python3 scripts/bench_srif_sizes.py 262144 > $OUT/write_$v.out 2>&1
done
cp /tmp/base_keep.so gokalman_amd/libgokalman_amd.so
python3 - > $OUT/counters.txt 2>&1 <<'PY'
import csv, glob, collections
for v in ("base", "e4l2"):
    for kind in ("sq", "fetch", "write"):
        acc = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob("gpurun_out/ab_config_e/%s_%s/**/*counter_collection.csv" % (kind, v), recursive=True):
            for r in csv.DictReader(open(f)):
                if "srif_" in r["Kernel_Name"]:
                    acc[r["Kernel_Name"]][r["Counter_Name"]].append(float(r["Counter_Value"]))
                    acc[r["Kernel_Name"]]["_vgpr"] = [float(r.get("VGPR_Count") or 0)]; acc[r["Kernel_Name"]]["_lds"] = [float(r.get("LDS_Block_Size") or 0)]; acc[r["Kernel_Name"]]["_scr"] = [float(r.get("Scratch_Size") or 0)]
        for kn, cs in acc.items():
            print(v, kind, kn[:70], {c: round(sum(x) / len(x), 1) for c, x in cs.items()})
PY
find $OUT -name "*.csv" -delete; find $OUT -type d -empty -delete
cat $OUT/timings.txt $OUT/parity_four_lane.txt $OUT/counters.txt
