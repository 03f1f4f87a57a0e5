# A/B of device-code builds (treeqp_amd/lib_var/<name>, see tools/vbuild.sh) on C4: rocprofv3 kernel averages of 40 solves per build, two rounds.
# usage (through gpurun): bash tools/ab_kernels.sh dev other ...
# A/B of device-code variants on C4: rocprofv3 kernel averages of 30 solves each, two rounds
mkdir -p gpurun_out/r3e && cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for round in 1 2; do
for v in "$@"; do
  TREEQP_AMD_LIB=$GRAFT_REPO_ROOT/treeqp_amd/lib_var/$v/libtreeqp_amd.so rocprofv3 --kernel-trace --stats -d gpurun_out/r3e/prof_${v}_$round -o c4 -- python3 tools/prof_flat.py C4 40 > gpurun_out/r3e/prof_$v.log 2>&1
  echo "$v: $(python tools/kstats.py gpurun_out/r3e/prof_${v}_$round/c4_results.db | grep -E 'k_hf_w|k_sg|k_fwd3' | awk -F, '{printf "%s %s/%s  ", substr($1,22,8), $4, $6}')"
done
done
