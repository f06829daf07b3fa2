#!/bin/bash
# developer tool: instructions per cell-wave of orb_fast_cells by phase (build_exp/libps_stop<n>.so: tools/build_variant.sh stop<n> orb_kernels.hip -DFAST_STOP=<n>)
for v in ${@:-default stop1 stop2 stop3 stop4 stop5}; do
  if [ $v = default ]; then unset PS_LIB_PATH; else export PS_LIB_PATH=$PWD/build_exp/libps_$v.so; fi
  echo "== $v"; tools/pmc_py.sh fp_$v "SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_INSTS_LDS" tools/orb_real_bench.py 2>&1 | grep "orb_fast"
done
