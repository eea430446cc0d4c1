#!/bin/bash
# runs the bench once per experimental build of the library (imageprocessing_amd/libipamd_exp*.so)
cp imageprocessing_amd/libipamd_hip.so /tmp/lib_keep.so
for f in imageprocessing_amd/libipamd_exp*.so; do
  cp $f imageprocessing_amd/libipamd_hip.so
  echo "== $f"; bash scripts/ab.sh "A=1"
done
cp /tmp/lib_keep.so imageprocessing_amd/libipamd_hip.so
