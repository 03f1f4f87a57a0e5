#!/bin/bash
# CPU sanitizer pass (AddressSanitizer + UBSan) over the host C layer and the oracle; GPU sanitizers are not available on the
# pool.  Builds sanitized copies next to the real artefacts, runs the CPU test files that exercise them, restores the originals.
set -e
cd "$(dirname "$0")/.."
mkdir -p build/asan
H=treeqp_amd/csrc/host
SAN="-g -O1 -std=gnu99 -fPIC -fsanitize=address,undefined -fno-omit-frame-pointer"
for f in blasfeo_compat tree_topology host_utils qp_container tdunes_host; do gcc $SAN -Iinclude -c $H/$f.c -o build/asan/$f.o; done
hipcc -shared -fPIC --offload-arch=gfx950 build/asan/*.o build/obj/device_*.o -o build/asan/libtreeqp_amd.so -lm -fsanitize=address,undefined      # (the device parts of the current product build: python treeqp_amd/build.py first)
gcc $SAN -fopenmp -shared -o build/asan/liboracle.so oracle/tdunes_oracle.c -lm
cp treeqp_amd/lib/libtreeqp_amd.so build/asan/lib_orig.so; cp oracle/liboracle.so build/asan/liboracle_orig.so
restore() { cp build/asan/lib_orig.so treeqp_amd/lib/libtreeqp_amd.so; cp build/asan/liboracle_orig.so oracle/liboracle.so; }
trap restore EXIT
cp build/asan/libtreeqp_amd.so treeqp_amd/lib/libtreeqp_amd.so; cp build/asan/liboracle.so oracle/liboracle.so
export LD_PRELOAD=$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1
python -m pytest tests/test_container.py tests/test_tree.py tests/test_oracle.py -x -q -m "not gpu" --deselect tests/test_abi.py
