#!/bin/bash
for v in default lvstop; do
  if [ $v = default ]; then unset PS_LIB_PATH; else export PS_LIB_PATH=$PWD/build_exp/libps_$v.so; fi
  echo "== $v"; tools/pmc_py.sh lp_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" tools/orb_real_bench.py 2>&1 | grep "orb_level"
done
