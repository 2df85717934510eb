#!/bin/bash
# ab/lib_base.so = the library at HEAD, ab/lib_new.so = the library of the working tree.
set -e
cd "$(dirname "$0")/.."
mkdir -p ab
csrc=graph-unet-traffic-prediction_amd/csrc
make -C $csrc >/dev/null && cp graph-unet-traffic-prediction_amd/libgunet_hip.so ab/lib_new.so
git stash -q
make -C $csrc >/dev/null && cp graph-unet-traffic-prediction_amd/libgunet_hip.so ab/lib_base.so
git stash pop -q
make -C $csrc >/dev/null
echo built ab/lib_base.so ab/lib_new.so
