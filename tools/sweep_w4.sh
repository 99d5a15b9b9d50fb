#!/bin/bash
# Schedule sweep of the two generated four-wave kernels (tools/gen/gen_w4.py knobs): every variant is generated and compiled in a
# private copy of tools/gen + m3p_amd/csrc under /tmp/sweep_w4 and linked to m3p_amd/libm3p_hip_sw_<name>.so (git-ignored), for
# tools/ab_gemm.py / tools/ab_wgrad.py on the GPU box.   usage: tools/sweep_w4.sh name "ENV=VAL ENV2=VAL" [name2 "..."] ...
set -u
R=$(cd "$(dirname "$0")/.." && pwd)
W=/tmp/sweep_w4; mkdir -p $W
FLAGS="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics -Wno-unused-result -ffp-contract=fast"
build_one() {
  name=$1; envs=$2
  d=$W/$name; rm -rf $d; mkdir -p $d/tools $d/m3p_amd $d/include
  cp -r $R/tools/gen $d/tools/; cp -r $R/m3p_amd/csrc $d/m3p_amd/; cp $R/include/*.h $d/include/
  (cd $d && env $envs python tools/gen/gen_w4.py > gen.log 2>&1) || { echo "$name: generator refused ($envs)"; tail -2 $d/gen.log; return; }
  (cd $d/m3p_amd/csrc && /opt/rocm/bin/hipcc $FLAGS -c gemm.hip -o gemm.o > cc.log 2>&1) || { echo "$name: compile failed"; tail -3 $d/m3p_amd/csrc/cc.log; return; }
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $R/m3p_amd/libm3p_hip_sw_$name.so $d/m3p_amd/csrc/gemm.o $(ls $R/m3p_amd/csrc/*.o | grep -v "/gemm.o")
  echo "$name: ok ($envs)"
}
n=0
while [ $# -ge 2 ]; do
  build_one "$1" "$2" &
  shift 2
  n=$((n + 1)); [ $((n % 7)) -eq 0 ] && wait
done
wait
