for spec in "" "SSN_PHASED_SOLO=1" "SSN_PHASED_PES_FOLD=1" "SSN_PHASED_SOLO=1 SSN_PHASED_PES_FOLD=1" "SSN_CYCLE_STEPS=32" "SSN_CYCLE_STEPS=64"; do
  echo "== $spec"
  env $spec SSN_DEBUG_PLAN=1 timeout -k 10 250 python bench.py --workload slam --rehearse-dist --steps 3 --warmup 1 > gpurun_out/cyc_x.json 2> gpurun_out/cyc_x.err || exit 1
  tail -c 80 gpurun_out/cyc_x.json; echo; grep "pipelined over the exchange" gpurun_out/cyc_x.err | cut -c1-200
done
