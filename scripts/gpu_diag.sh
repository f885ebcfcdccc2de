#!/bin/bash
# Diagnostic GPU session with tight timeouts: watchdog build of the library + native smoke.
mkdir -p gpurun_out /tmp/wd
rm -f gpurun_out/diag.log
HIPCC=/opt/rocm/bin/hipcc
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=off -Wno-unused-function"
cd string_grouper_amd/csrc
for f in sg_api sg_postings sg_vectorize sg_matchlist; do cp $f.o /tmp/wd/$f.o; done
$HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_topn.hip -o /tmp/wd/sg_spgemm_topn.o || exit 1
$HIPCC $FLAGS -DSG_WATCHDOG -c sg_spgemm_pruned.hip -o /tmp/wd/sg_spgemm_pruned.o || exit 1
$HIPCC --offload-arch=gfx950 -shared -fPIC -o /tmp/wd/libsg_hip.so /tmp/wd/*.o || exit 1
cd ../..
$HIPCC -O2 tests/native/k4_smoke.cpp -Iinclude -L/tmp/wd -lsg_hip -ldl -Wl,-rpath,/tmp/wd -o /tmp/wd/k4_smoke || exit 1
for args in "200 300" "2000 3000" "20000 30000"; do
  for prune in 1 0; do
    echo "== args $args prune $prune" >> gpurun_out/diag.log
    SG_PRUNE=$prune timeout 40 /tmp/wd/k4_smoke $args >> gpurun_out/diag.log 2>&1
    echo "exit $?" >> gpurun_out/diag.log
  done
done
cat gpurun_out/diag.log
