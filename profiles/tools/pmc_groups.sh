#!/bin/bash
# One rocprofv3 --pmc pass per counter group (never combined with anything but --kernel-trace) over profiles/tools/pmc_apply.py.
# usage: bash profiles/tools/pmc_groups.sh <tag> <pmc_apply.py arguments ...>      -> gpurun_out/pmc_<tag>/<group>/..., then
#        python3 profiles/tools/pmc_summarize.py gpurun_out/pmc_<tag> profiles/<name>.json
TAG=$1; shift
REPO=$(pwd); OUT=$REPO/gpurun_out/pmc_$TAG; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
i=0
while read -r GROUP; do
    [ -z "$GROUP" ] && continue
    i=$((i + 1))
    timeout -k 10 400 rocprofv3 --pmc $GROUP --kernel-trace --output-format csv -d $OUT/g$i -o g$i -- python3 $REPO/profiles/tools/pmc_apply.py "$@" > $OUT/g$i.log 2>&1
    rc=$?
    echo "[$TAG] group $i rc=$rc: $GROUP | $(tail -1 $OUT/g$i.log | cut -c1-200)"
    if [ $rc -eq 124 ] || [ $rc -eq 137 ]; then echo "timeout: stopping"; exit 1; fi
done <<'GROUPS'
SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU GRBM_GUI_ACTIVE
SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM SQ_LEVEL_WAVES SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS
TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_TCC_READ_REQ_sum
TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum
TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum
TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum TCC_TAG_STALL_sum
TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_32B_sum
TCC_EA0_WRREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_64B_sum
FETCH_SIZE
WRITE_SIZE
GROUPS
cd $REPO
