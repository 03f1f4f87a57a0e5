#!/bin/bash
# experiment build of the device code only: tools/vbuild.sh NAME [-DFOO ...]  ->  treeqp_amd/lib_var/NAME/libtreeqp_amd.so (host objects must exist)
set -e
cd "$(dirname "$0")/.."
name=$1; shift
out=treeqp_amd/lib_var/$name; mkdir -p $out
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 ${TQ_SCHED--mllvm -amdgpu-sched-strategy=max-ilp} -fPIC -std=c++17 -Wall -Wno-unused-function -DTQ_SMALL_TABLE "$@" -Iinclude -Itreeqp_amd/csrc/device -c treeqp_amd/csrc/device/tdunes_device.hip -o $out/tdunes_device.hip.o
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 build/obj/blasfeo_compat.c.o build/obj/tree_topology.c.o build/obj/host_utils.c.o build/obj/qp_container.c.o build/obj/tdunes_host.c.o $out/tdunes_device.hip.o -o $out/libtreeqp_amd.so -lm
