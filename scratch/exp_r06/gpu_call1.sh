#!/bin/bash
# round 6, first GPU call: does the tree work on the box (symlinked package dir, host-visible single aviaries), the new parity tests with
# their measured figures, the placement experiment (plain + three counter passes), then the whole GPU suite
O=gpurun_out/r06a; mkdir -p $O
ls -la | grep gym > $O/tree.log; ls gym-pybullet-drones_amd/csrc | head -3 >> $O/tree.log
python scratch/exp_r06/dropin_latency.py > $O/dropin.log 2>&1
timeout 1200 python -m pytest tests -m gpu -q -s -k "host_visible or dropin_aviary or envelope or population or fixture_multihover or auto_reset or kinematic_planes or dslpid_calls or pid_circle or velocity_aviary" > $O/new_tests.log 2>&1
grep -h "MEASURED\|POPULATION\|ENVELOPE\|passed\|failed" $O/new_tests.log > $O/new_tests_summary.log
python scratch/exp_r06/placement_cause.py --trials 10 --tag plain > $O/place_plain.log 2>&1
python scratch/exp_r06/placement_cause.py --trials 6 --mode slab --tag slab > $O/place_slab.log 2>&1
R=$PWD; cd /tmp; export TMPDIR=/tmp
for pass in "tlb TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum" \
            "tccw TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL TCC_EA0_WRREQ_DRAM_CREDIT_STALL" \
            "tccr TCC_EA0_RDREQ TCC_EA0_RDREQ_DRAM_CREDIT_STALL TCC_TAG_STALL TCC_BUBBLE"; do
  set -- $pass; tag=$1; shift
  timeout 600 rocprofv3 --kernel-trace --pmc "$@" -d $R/$O/pmc_$tag --output-format csv json -- python $R/scratch/exp_r06/placement_cause.py --trials 8 --tag pmc_$tag > $R/$O/place_pmc_$tag.log 2>&1
  python $R/scratch/exp_r06/summarize_pmc.py $R/$O/pmc_$tag $R/$O/pmc_$tag.json.gz >> $R/$O/place_pmc_$tag.log 2>&1
  find $R/$O/pmc_$tag -name "*.csv" -size +20M -delete; find $R/$O/pmc_$tag -name "*.json" -delete
done
cd $R
timeout 1500 python -m pytest tests -m gpu -x -q > $O/full_gpu.log 2>&1; tail -3 $O/full_gpu.log
du -sh $O
