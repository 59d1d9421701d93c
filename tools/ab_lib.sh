#!/bin/bash
# diagnostic: A/B of two builds of the library on one box (box-to-box clock differences are ~10 %):
#   tools/ab_lib.sh tools/_lib_a.so tools/_lib_b.so   (alternates them; restores the first at the end)
A=$1; B=$2
for rep in 1 2; do
  for L in $A $B; do
    cp $L cerberus_amd/libcerberus_mfp.so
    echo "== $L"
    tools/seg_sweep.sh 4096 CMFP_X=1 || exit 1
    tools/seg_sweep3d.sh 256 CMFP_X=1 || exit 1
  done
done
cp $A cerberus_amd/libcerberus_mfp.so
