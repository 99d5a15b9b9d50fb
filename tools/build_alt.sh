#!/bin/bash
# Builds an alternate library m3p_amd/libm3p_hip_alt.so with extra -D flags for A/B runs
# (select it with M3P_HIP_LIB=m3p_amd/libm3p_hip_alt.so).  usage: tools/build_alt.sh -DFOO=0 ...
set -e
cd "$(dirname "$0")/../m3p_amd/csrc"
mkdir -p /tmp/m3p_alt
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fvisibility=hidden -munsafe-fp-atomics \
    -Wno-unused-result -ffp-contract=fast "$@" -c $f -o /tmp/m3p_alt/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libm3p_hip_alt.so /tmp/m3p_alt/*.o
echo built alt
