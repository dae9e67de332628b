#!/bin/bash
mkdir -p gpurun_out; export PYTHONUNBUFFERED=1 TMPDIR=/tmp
R=$PWD; OUT=$R/gpurun_out/pmc3; rm -rf $OUT; mkdir -p $OUT
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_INSTS_VALU -d $OUT -o pmca --output-format csv -- python $R/tools/pmc_attn.py ) > gpurun_out/pmc3_run.log 2>&1
( cd /tmp && timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_WAIT_INST_LDS SQ_ACTIVE_INST_VMEM SQ_INSTS_VMEM_RD SQ_WAVES -d $OUT -o pmcb --output-format csv -- python $R/tools/pmc_attn.py ) >> gpurun_out/pmc3_run.log 2>&1
python tools/pmc_summary.py $OUT/pmca_counter_collection.csv $OUT/pmcb_counter_collection.csv > gpurun_out/pmc3_summary.txt 2>&1
python - <<'PY'
L=open('gpurun_out/pmc3_summary.txt').read().splitlines()
hdr=L[0].split()
for l in L[1:]:
    p=l.split()
    if not (p[0].startswith('s3_') or p[0].startswith('xattn')): continue
    d=dict(zip(hdr[3:], [float(x) for x in p[2:]]))
    wc=d['SQ_WAVE_CYCLES']
    print(f"{p[0]:22s} active {d['SQ_ACTIVE_INST_ANY']/wc:.2f} valu {d['SQ_ACTIVE_INST_VALU']/wc:.2f} lds {d['SQ_ACTIVE_INST_LDS']/wc:.2f} wait_any {d['SQ_WAIT_ANY']/wc:.2f} wait_inst {d['SQ_WAIT_INST_ANY']/wc:.2f} | insts valu {d['SQ_INSTS_VALU']:.3g} salu {d['SQ_INSTS_SALU']:.3g} lds {d['SQ_INSTS_LDS']:.3g} | lds_active {d['SQ_LDS_IDX_ACTIVE']:.3g} bank_conf {d['SQ_LDS_BANK_CONFLICT']:.3g} ({d['SQ_LDS_BANK_CONFLICT']/max(d['SQ_LDS_IDX_ACTIVE'],1):.2f})")
PY
