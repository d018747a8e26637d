#!/bin/bash
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; O=gpurun_out/ldsprobe; mkdir -p $O
rocprofv3 --pmc SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAVES --output-format csv -d $O/a -- python3 scripts/bench_kinds.py ${KINDS:-vsplit vpad} > $O/a.out 2> $O/a.log
rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_ADDR_CONFLICT SQ_LDS_UNALIGNED_STALL SQ_BUSY_CYCLES --output-format csv -d $O/b -- python3 scripts/bench_kinds.py ${KINDS:-vsplit vpad} > $O/b.out 2> $O/b.log
python3 - <<'PY'
import csv,glob,collections
for tag in ('a','b'):
    agg=collections.defaultdict(lambda: collections.defaultdict(float)); cnt=collections.Counter()
    for f in glob.glob('gpurun_out/ldsprobe/%s/**/*counter_collection.csv'%tag, recursive=True):
        for r in csv.DictReader(open(f)):
            k=r['Kernel_Name'][:90]
            agg[k][r['Counter_Name']]+=float(r['Counter_Value']); 
            if r['Counter_Name']=='SQ_BUSY_CYCLES': cnt[k]+=1
    with open('gpurun_out/ldsprobe/%s_summary.txt'%tag,'w') as out:
        for k,v in agg.items():
            if not any(t in k for t in ("split", "srif", "vanilla_reg", "mc_", "chisq", "shared")): continue
            n=max(cnt[k],1)
            out.write(k+' | launches %d | '%n+' '.join('%s=%.4g'%(c,x/n) for c,x in sorted(v.items()))+'\n')
PY
tail -3 $O/a.log $O/b.log; cat $O/a_summary.txt $O/b_summary.txt; find $O -name "*.csv" -delete; find $O -name "*.db" -delete
