#!/bin/bash
# round 6, second GPU call: the offset sweep inside one slab, and the level / hit counters on alternating slow (slab) and fast (torch) placements
O=gpurun_out/r06b; mkdir -p $O
python scratch/exp_r06/placement_sweep.py --slabs 3 --tag sweep > $O/sweep.log 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
for pass in "lvlr TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_TAG_STALL TCC_CYCLE" \
            "lvlw TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_BUSY" \
            "hit TCC_HIT TCC_MISS TCC_WRITEBACK TCC_NORMAL_EVICT" \
            "req TCC_REQ TCC_STREAMING_REQ TCC_IB_STALL TCC_LATENCY_FIFO_FULL" \
            "sq SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD"; do
  set -- $pass; tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$tag --output-format csv -- python $R/scratch/exp_r06/placement_cause.py --trials 8 --mode alt --tag pmc2_$tag > $R/$O/place_pmc_$tag.log 2>&1
  python $R/scratch/exp_r06/summarize_pmc.py $R/$O/pmc_$tag $R/$O/pmc_$tag.json.gz >> $R/$O/place_pmc_$tag.log 2>&1
  rm -rf $R/$O/pmc_$tag
done
cd $R; du -sh $O
